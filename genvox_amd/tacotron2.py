"""Host-side mirror of the reference's ``models.tts.tacotron2.Tacotron2`` for the MI355X forward path.

Same constructor, attributes, ``state_dict`` key names, ``forward(batch)`` / ``inference(inputs)``
contracts and checkpoint helpers as the reference (models/tts/tacotron2.py:416-596,
models/tts/__init__.py:64-95), but no torch.nn compute: parameters are plain tensors in a module
tree, and ``forward`` / ``inference`` hand device pointers to the HIP library through the C ABI
(include/genvox_amd.h).  PyTorch is used for allocation, streams and the user-visible tensors only.

Differences from the reference, on purpose:
  * no autograd: ``forward`` builds no graph in either mode; the training step (``train_step``)
    runs an explicit backward through the whole model (genvox_amd/training.py);
  * the Prenet's always-on dropout (tacotron2.py:143) takes explicit keep masks when the caller
    supplies ``batch["prenet_keep_masks"]`` (parity tests); otherwise masks are drawn on the device
    from a seed taken from torch's RNG, so ``torch.manual_seed`` still makes runs repeatable;
  * ``inference`` accepts a batch (the reference is batch-1 only, tacotron2.py:405) and then also
    returns ``mel_lengths``;
  * ``forward(batch, strict=True)`` synchronises once per call to make sure the call's in-launch hand-offs all happened, and
    runs the call again on the launch-per-step kernels if they did not (``strict=False``: fully asynchronous; a timed-out
    call then returns NaN in every output and ``check_status()`` raises).
"""
from __future__ import annotations

import ctypes as C
import os
import math
import warnings
from typing import Dict, Optional

import numpy as np
import torch
from torch import nn

from . import _lib
from .configs import AudioConfig, BaseConfig, Tacotron2Config, TextConfig
from .weights import state_dict_spec

MAX_CALL_BATCH = 64   # rows per C-ABI call of the recurrent entry points
STREAM_ROWS = 32      # teacher-forced batches above this are cut into chunks of at most this many rows that run on two HIP
                      # streams at once: a decoder step is one chip-wide weight-streaming launch followed by a small
                      # latency-bound attention launch, so a second independent batch fills the attention gaps of the first
                      # (measured: 64 rows as 2 x 32 concurrently 40.6 ms, as one 64-row call 43.5 ms)

_GAIN = {"linear": "linear", "sigmoid": "sigmoid", "tanh": "tanh", "relu": "relu"}


class _Node(nn.Module):
    """Parameter container; children are created on demand so dotted names nest like the reference's modules."""

    def child(self, name: str) -> "_Node":
        if name not in self._modules:
            self.add_module(name, _Node())
        return self._modules[name]


def dims_from_configs(mc: Tacotron2Config, ac: AudioConfig, tc: TextConfig) -> _lib.gvx_dims:
    assert tc.n_tokens is not None and tc.n_tokens >= 1, "text_config.n_tokens must be set (size of the token map)"
    return _lib.gvx_dims(
        n_tokens=tc.n_tokens, embed_dim=mc.encoder_embedding_dim, enc_kernel=mc.encoder_kernel_size,
        enc_n_conv=mc.encoder_n_convolutions, prenet_dim=mc.prenet_dim, att_rnn_dim=mc.attention_rnn_dim,
        dec_rnn_dim=mc.decoder_rnn_dim, att_dim=mc.attention_dim, att_loc_filters=mc.attention_location_n_filters,
        att_loc_kernel=mc.attention_location_kernel_size, postnet_dim=mc.postnet_embedding_dim,
        postnet_kernel=mc.postnet_kernel_size, postnet_n_conv=mc.postnet_n_convolutions, n_mels=ac.n_mels)


class Tacotron2(nn.Module):
    def __init__(self, model_config: Tacotron2Config, audio_config: AudioConfig, text_config: TextConfig) -> None:
        super().__init__()
        self.model_name = self.__class__.__name__
        self.model_config = model_config
        self.audio_config = audio_config
        self.text_config = text_config
        self._spec = state_dict_spec(model_config, audio_config, text_config)
        for key, (shape, kind, arg) in self._spec.items():
            *path, leaf = key.split(".")
            node = self
            for name in path:
                node = node.child(name)
            if kind == "count":
                node.register_buffer(leaf, torch.zeros((), dtype=torch.long))
            elif kind in ("bn_mean", "bn_var"):
                node.register_buffer(leaf, torch.ones(shape) if kind == "bn_var" else torch.zeros(shape))
            else:
                t = torch.empty(shape)
                if kind == "xavier":
                    nn.init.xavier_uniform_(t, gain=nn.init.calculate_gain(_GAIN[arg]))
                elif kind == "uniform":
                    t.uniform_(-arg, arg)
                elif kind == "bn_w":
                    t.fill_(1.0)
                elif kind == "bn_b":
                    t.zero_()
                node.register_parameter(leaf, nn.Parameter(t, requires_grad=False))
        self._handle: Optional[int] = None
        self._key_tensors = None
        self._blob: Optional[torch.Tensor] = None
        self._host_repack = os.environ.get("GVX_HOST_REPACK") == "1"   # A/B knob: every re-pack through the host packer
        self._packed_key = None
        self._workspace: Optional[torch.Tensor] = None
        self._lane_ws = [None, None]     # workspaces of the two concurrent chunk lanes
        self._lane_streams = None
        self._lane_handles = [None, None]   # autoregressive lanes run on host threads: one C-ABI handle each (same blob)
        self._timing = False
        self._sticky_bad = False        # host-side copies of the device status words (a replaced workspace takes its words with it)
        self._sticky_stalled = False
        self._resident_off = False      # a hand-off timed out once: this model runs on the launch-per-step kernels from then on
        self.eval()

    # child() for the root, which is not a _Node
    def child(self, name: str) -> _Node:
        return _Node.child(self, name)

    # ------------------------------------------------------------------ C-ABI plumbing
    def __del__(self):
        try:
            for h in [self._handle] + list(getattr(self, "_lane_handles", [])):
                if h is not None:
                    _lib.load().gvx_model_destroy(h)
        except Exception:
            pass

    def _device(self) -> torch.device:
        return self.embedding.weight.device

    def _require_gpu(self) -> torch.device:
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("genvox_amd.Tacotron2 runs on an MI355X only: move the model with .to('cuda:0'). "
                               "There is no CPU fallback (the reference's CPU path is the oracle under oracle/).")
        if self._lane_streams is None:
            # The two side streams of the large-batch paths are created before the C library creates streams of its own
            # (capture stream, the persistent attention kernel's stream): HIP deals streams to a few hardware queues in
            # order of first use, and in the other order the two lanes ended up on one queue and ran one after the other
            # (measured: 54 instead of 43 ms for 64 x 800 frames).
            self._lane_streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
            for lane in self._lane_streams:
                with torch.cuda.stream(lane):
                    torch.zeros(1, device=dev)   # first launch on the stream: this is what binds it to a queue
        return dev

    def _weights_key(self):
        """Identity + version of every weight tensor.  Walks a cached list of the parameter / buffer objects (forward()
        calls this: building a state_dict per call cost ~0.3 ms of host time inside the timed region)."""
        ts = self._key_tensors
        if ts is None:
            ts = self._key_tensors = [v for v in self.state_dict(keep_vars=True).values()]
        return (str(self._device()),) + tuple((v.data_ptr(), v._version) for v in ts)

    def _apply(self, fn, *args, **kwargs):   # .to() / .cuda(): parameter objects may be replaced
        out = super()._apply(fn, *args, **kwargs)
        self._key_tensors = None
        return out

    def _ensure_handle(self) -> int:
        if self._handle is None:
            h = C.c_void_p()
            dims = dims_from_configs(self.model_config, self.audio_config, self.text_config)
            _lib.check(_lib.load().gvx_model_create(C.byref(dims), C.byref(h)))
            self._handle = h.value
        return self._handle

    def blob_numel(self) -> int:
        """Number of fp32 elements of this model's packed weight blob (fixed by the configs)."""
        return _lib.load().gvx_model_blob_bytes(self._ensure_handle()) // 4

    def pack_weights_host(self) -> torch.Tensor:
        """Fold / repack the current state_dict into the blob layout the kernels read (host memory, host-only work:
        BatchNorm folding, LSTM gate-row permutation, MFMA-fragment order).  Needs no GPU."""
        lib = _lib.load()
        h = self._ensure_handle()
        host = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in self.state_dict().items()
                if v.is_floating_point()}
        table = (_lib.gvx_weight_desc * len(host))()
        for i, (k, v) in enumerate(host.items()):
            table[i] = _lib.gvx_weight_desc(k.encode(), v.data_ptr(), v.numel())
        blob_host = torch.empty(self.blob_numel(), dtype=torch.float32)
        _lib.check(lib.gvx_model_pack_weights(h, table, len(host), blob_host.data_ptr()))
        return blob_host

    def pack_weights_device(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The same blob built on the device from the parameters where they are (gvx_model_pack_weights_device): what the
        training loop calls after every optimizer step - bit-identical to ``pack_weights_host`` (tests/test_training_gpu.py)."""
        dev = self._require_gpu()
        lib = _lib.load()
        h = self._ensure_handle()
        srcs = {k: v.detach() for k, v in self.state_dict().items() if v.is_floating_point()}
        for k, v in srcs.items():
            if v.device != dev or v.dtype != torch.float32 or not v.is_contiguous():
                srcs[k] = v.to(device=dev, dtype=torch.float32).contiguous()
        table = (_lib.gvx_weight_desc * len(srcs))()
        for i, (k, v) in enumerate(srcs.items()):
            table[i] = _lib.gvx_weight_desc(k.encode(), v.data_ptr(), v.numel())
        blob = torch.empty(self.blob_numel(), dtype=torch.float32, device=dev) if out is None else out
        _lib.check(lib.gvx_model_pack_weights_device(h, table, len(srcs), blob.data_ptr(), self._stream()))
        return blob

    def _ensure_packed(self) -> None:
        """Make the device blob match the current state_dict (once per weight version).  The first blob of a model comes from
        the host packer (which needs no GPU state); later versions - a training loop re-packs after every optimizer step -
        are re-built in place on the device."""
        dev = self._require_gpu()
        key = self._weights_key()
        if self._packed_key == key:
            return
        if self._blob is not None and self._blob.device == dev and self._blob.numel() == self.blob_numel() and not self._host_repack:
            self.pack_weights_device(out=self._blob)   # same storage: the handles' binding and cached graphs stay valid
        else:
            self._bind(self.pack_weights_host().to(dev))
        self._packed_key = key

    def _bind(self, blob: torch.Tensor) -> None:
        lib = _lib.load()
        _lib.check(lib.gvx_model_bind_blob(self._ensure_handle(), blob.data_ptr()))
        for h in self._lane_handles:
            if h is not None:
                _lib.check(lib.gvx_model_bind_blob(h, blob.data_ptr()))
        self._blob = blob   # replaces (and frees) the previous blob only after the handle points at the new one
        if self._timing:
            _lib.check(lib.gvx_stage_timing_enable(self._handle, 1))

    def packed_blob(self) -> torch.Tensor:
        """The packed weight blob on the model's device (what a multi-GPU start-up broadcasts, see genvox_amd.dist).
        On a CPU-resident model this is the host blob (rehearsal of the broadcast protocol; nothing can run on it)."""
        if self._device().type != "cuda":
            return self.pack_weights_host()
        self._ensure_packed()
        return self._blob

    def bind_packed_blob(self, blob: torch.Tensor) -> None:
        """Adopt a packed blob produced by another rank's ``packed_blob()`` (same configs).  On a CUDA model the kernels
        read it from now on; on a CPU model it is only kept (``_blob``) so that the protocol can be checked without a GPU."""
        assert blob.dtype == torch.float32 and blob.is_contiguous() and blob.device == self._device()
        assert blob.numel() == self.blob_numel(), "blob size does not match this model's dims"
        if blob.device.type == "cuda":
            self._bind(blob)
        else:
            self._blob = blob
        self._packed_key = self._weights_key()

    def _get_workspace(self, B: int, L: int, T: int) -> torch.Tensor:
        return self._workspace_of(_lib.load().gvx_workspace_bytes(self._handle, B, L, T))

    def _workspace_of(self, need: int) -> torch.Tensor:
        """The main workspace, grown to `need` bytes.  The status words live at its front: before a workspace is replaced they
        are read into the host-side sticky flags, so that an error raised by an earlier call survives until somebody looks."""
        dev = self._device()
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != dev:
            if self._workspace is not None:
                self._poll_status()
            self._workspace = self._new_workspace(need, dev)
        return self._workspace

    def _stream(self) -> int:
        return torch.cuda.current_stream(self._device()).cuda_stream

    @staticmethod
    def _new_workspace(nbytes: int, dev) -> torch.Tensor:
        """Scratch for the C-ABI calls.  The status words at its front (gvx_workspace_status) are only ever written by the
        kernels that raise them, so they start out cleared; the rest needs no initialisation."""
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ws[:65536].zero_()
        return ws

    def _poll_status(self) -> None:
        """Read and clear the device status words of every workspace into the host-side sticky flags (synchronises the stream)."""
        out = (C.c_int32 * 2)()
        for ws in [self._workspace] + self._lane_ws:
            if ws is None:
                continue
            _lib.check(_lib.load().gvx_workspace_status(self._handle, ws.data_ptr(), ws.numel(), self._stream(), out))
            self._sticky_bad |= bool(out[0])
            self._sticky_stalled |= bool(out[1])

    def check_status(self) -> None:
        """Report and clear the sticky status of every call since the last check (synchronises the stream): raises IndexError
        for a token id outside the embedding table, which is what nn.Embedding does in the reference
        (models/tts/tacotron2.py:459), RuntimeError when an in-launch hand-off of a resident loop timed out (that call's
        outputs are all NaN, so they cannot be mistaken for results even without this check).
        ``forward(strict=False)`` does not call it - it costs a synchronisation; one check after a batch of calls is enough."""
        self._poll_status()
        bad, stalled = self._sticky_bad, self._sticky_stalled
        self._sticky_bad = self._sticky_stalled = False
        if stalled:
            raise RuntimeError("genvox_amd: an in-launch hand-off of the decoder loop timed out; the last outputs are invalid")
        if bad:
            raise IndexError("genvox_amd: token id outside [0, n_tokens)")

    def _stalled_once(self) -> bool:
        """After a call that has been synchronised: did one of its in-launch hand-offs time out?  If so - the resident kernels
        could not run at the same time here (CU masking, a serialising profiler, a co-tenant on the GPU) - this model switches
        to the launch-per-step kernels for good (gvx_model_set_resident_kernels; same results, no kernel waits for another
        one) and the caller runs the call again.  A second time-out in a row cannot happen on that path."""
        self._poll_status()
        if not self._sticky_stalled:
            return False
        self._sticky_stalled = False
        if self._resident_off:
            raise RuntimeError("genvox_amd: a hand-off timed out although the resident kernels are switched off")
        import warnings

        warnings.warn("genvox_amd: an in-launch hand-off of a resident kernel timed out (the kernels could not run at the same "
                      "time on this GPU); switching this model to the launch-per-step kernels and running the call again")
        lib = _lib.load()
        for h in [self._handle] + self._lane_handles:
            if h is not None:
                _lib.check(lib.gvx_model_set_resident_kernels(h, 0))
        self._resident_off = True
        return True

    def _ensure_lane_handles(self) -> None:
        """One C-ABI handle per chunk lane, bound to the same weight blob (a handle is not re-entrant: the lanes run
        concurrently, the autoregressive ones on host threads).  The resident attention kernel stays off on them - two
        of them would hold 64 CUs beside launches that need one CU per tile - and is never touched on the main handle."""
        lib = _lib.load()
        for i in range(2):
            if self._lane_handles[i] is None:
                h = C.c_void_p()
                dims = dims_from_configs(self.model_config, self.audio_config, self.text_config)
                _lib.check(lib.gvx_model_create(C.byref(dims), C.byref(h)))
                _lib.check(lib.gvx_model_set_persistent_attention(h.value, 0))
                if self._resident_off:
                    _lib.check(lib.gvx_model_set_resident_kernels(h.value, 0))
                _lib.check(lib.gvx_model_bind_blob(h.value, self._blob.data_ptr()))
                if self._timing:
                    _lib.check(lib.gvx_stage_timing_enable(h.value, 1))
                self._lane_handles[i] = h.value

    def _lanes(self, dev, need_bytes: int):
        """Two side streams with a workspace each (chunks of a large batch run on them concurrently)."""
        if self._lane_streams is None:
            self._lane_streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        for i in range(2):
            if self._lane_ws[i] is None or self._lane_ws[i].numel() < need_bytes or self._lane_ws[i].device != dev:
                if self._lane_ws[i] is not None:
                    self._poll_status()   # (the replaced workspace takes its status words with it)
                self._lane_ws[i] = self._new_workspace(need_bytes, dev)
        return self._lane_streams, self._lane_ws

    def enable_stage_timing(self, enable: bool = True) -> None:
        self._timing = enable
        for h in [self._handle] + self._lane_handles:
            if h is not None:
                _lib.check(_lib.load().gvx_stage_timing_enable(h, int(enable)))

    def stage_times_ms(self, lane: Optional[int] = None):
        """Stage times of the last whole-forward call on the main handle, or on chunk lane `lane` (batches above 32 rows)."""
        t = (C.c_float * 5)()
        n = C.c_int()
        _lib.check(_lib.load().gvx_stage_times_ms(self._handle if lane is None else self._lane_handles[lane], t, C.byref(n)))
        names = ("encoder", "prenet", "decoder_loop", "projection", "postnet")
        return dict(zip(names, list(t))), n.value

    def enable_kernel_timing(self, enable: bool = True) -> None:
        self._ensure_packed()
        _lib.check(_lib.load().gvx_kernel_timing_enable(self._handle, int(enable)))

    def loop_kind(self, B: int, L: int) -> int:
        """How a teacher-forced inference call of this shape runs its decoder loop (gvx_teacher_forced_loop_kind): 2 = one resident
        weight-stationary kernel, 1 = a weight-streaming launch per step beside the resident attention kernel, 0 = a launch pair
        per step."""
        self._ensure_packed()
        return int(_lib.load().gvx_teacher_forced_loop_kind(self._handle, B, L))

    def ar_loop_kind(self, B: int, L: int) -> int:
        """How an autoregressive call of this shape runs its decode (gvx_autoregressive_loop_kind): 2 = two resident kernels for the
        whole decode, 1 / 0 = launches per step (beside the resident attention kernel / not)."""
        self._ensure_packed()
        return int(_lib.load().gvx_autoregressive_loop_kind(self._handle, B, L))

    def kernel_times_ms(self):
        a, b, n = C.c_float(), C.c_float(), C.c_int()
        _lib.check(_lib.load().gvx_kernel_times_ms(self._handle, C.byref(a), C.byref(b), C.byref(n)))
        return {"decoder_lstm_step": a.value, "attention_step": b.value, "steps": n.value}

    def _keep_masks(self, given: Optional[torch.Tensor], n: int, dev) -> torch.Tensor:
        if given is not None:
            m = given.to(device=dev, dtype=torch.uint8).contiguous()
            assert m.numel() == n, f"prenet_keep_masks has {m.numel()} elements, expected {n}"
            return m
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())  # CPU generator: torch.manual_seed makes this repeatable
        m = torch.empty(n, dtype=torch.uint8, device=dev)
        _lib.check(_lib.load().gvx_prenet_masks_generate(m.data_ptr(), n, seed, self._stream()))
        return m

    # ------------------------------------------------------------------ reference surface
    def forward(self, batch: Dict[str, torch.Tensor], strict: bool = True) -> Dict[str, torch.Tensor]:
        """Teacher-forced text->mel (reference: models/tts/tacotron2.py:450-481).  No autograd graph is built in either
        mode.  Under ``.train()`` the forward is the reference's training-mode forward - BatchNorm batch statistics (and the
        running-statistics update), dropout after the encoder / Postnet convolutions and on both LSTM cells' outputs - see
        ``_forward_train`` (the backward through the whole model is genvox_amd.training.train_backward; ``train_step`` runs both).

        strict (default): like the reference, the call always returns numbers - it synchronises once, and if an in-launch
        hand-off of its resident kernels timed out (they could not run at the same time on this GPU) it is run again on the
        launch-per-step kernels, which this model keeps from then on.  strict=False: nothing synchronises; a timed-out call
        returns NaN in every output and raises at the next ``check_status()``."""
        out = self._forward_once(batch)
        if strict and not self.training and self._stalled_once():
            out = self._forward_once(batch)
            self._stalled_once()   # (raises if the second run reports a time-out as well)
        return out

    def _forward_once(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        dev = self._require_gpu()
        if self.training:
            return self._forward_train(batch)[0]
        self._ensure_packed()
        tokens = batch["token_padded"].to(device=dev, dtype=torch.int64).contiguous()
        tok_len = batch["token_lengths"].to(device=dev, dtype=torch.int32).contiguous()
        mel_in = batch["mel_padded"].to(device=dev, dtype=torch.float32).contiguous()
        mel_len = batch["mel_lengths"].to(device=dev, dtype=torch.int32).contiguous()
        B, L = tokens.shape
        _, M, T = mel_in.shape
        assert M == self.audio_config.n_mels
        P = self.model_config.prenet_dim
        given = batch.get("prenet_keep_masks")
        if given is not None:
            given = given.to(device=dev, dtype=torch.uint8).reshape(2, T + 1, B, P)
        out = {"mel_outputs": torch.empty(B, M, T, device=dev), "mel_outputs_postnet": torch.empty(B, M, T, device=dev),
               "gate_outputs": torch.empty(B, T, device=dev), "alignments": torch.empty(B, T, L, device=dev)}
        lib = _lib.load()

        def run(lo: int, hi: int, ws: torch.Tensor, handle: int, postnet: bool = True) -> None:
            n = hi - lo
            masks = self._keep_masks(given[:, :, lo:hi].contiguous() if given is not None else None, 2 * (T + 1) * n * P, dev)
            _lib.check(lib.gvx_tacotron2_forward(
                handle, tokens[lo:hi].data_ptr(), tok_len[lo:hi].data_ptr(), n, L, mel_in[lo:hi].data_ptr(),
                mel_len[lo:hi].data_ptr() if self.model_config.mask_padding else None, T, masks.data_ptr(),
                out["mel_outputs"][lo:hi].data_ptr(), out["mel_outputs_postnet"][lo:hi].data_ptr() if postnet else None,
                out["gate_outputs"][lo:hi].data_ptr(), out["alignments"][lo:hi].data_ptr(),
                ws.data_ptr(), ws.numel(), self._stream()))

        # rows one C-ABI call takes: 64 where the 64-row loop beside the resident attention kernel serves the shape (default
        # layer sizes, L <= 128: one pass over the weights for all 64 rows), else 32
        rows = STREAM_ROWS
        if B > STREAM_ROWS and lib.gvx_teacher_forced_rows_per_call(self._handle, L) == MAX_CALL_BATCH:
            rows = MAX_CALL_BATCH
        self.last_forward_lanes = B > rows and rows == STREAM_ROWS   # (bench.py: whose stage timers hold the last call)
        self.last_forward_sequential32 = False
        if B <= rows:
            run(0, B, self._get_workspace(B, L, T), self._handle)
            return out
        # 32-row chunks that take the resident-attention loop run one after the other on the caller's stream: that loop fills
        # the chip by itself (2 x 18.7 ms for 64 x 800 frames), two concurrent lanes would have to give it up (39.3 ms).
        # GVX_TF_LANES=1 forces the lanes (tests, A/B runs).
        sequential = rows == MAX_CALL_BATCH or (os.environ.get("GVX_TF_LANES", "0") != "1"
                                                and lib.gvx_teacher_forced_resident(self._handle, STREAM_ROWS, L) == 1)
        self.last_forward_lanes = self.last_forward_lanes and not sequential
        self.last_forward_sequential32 = sequential and rows == STREAM_ROWS
        if sequential:   # chunks of at most `rows` rows, one after the other on the caller's stream
            n_chunks = -(-B // rows)
            bounds = [(B * i) // n_chunks for i in range(n_chunks + 1)]
            # (one workspace for the chunks' calls and the Postnet over all rows behind them)
            ws = self._workspace_of(max(lib.gvx_workspace_bytes(self._handle, max(hi - lo for lo, hi in zip(bounds, bounds[1:])), L, T),
                                        lib.gvx_postnet_workspace_bytes(self._handle, B, T)))
            # the Postnet and the padding mask once over all rows, behind the chunks' decoder loops: its GEMMs fill the chip better on
            # B x T frames than on a chunk's (models/tts/tacotron2.py:464-473 in the same order: Postnet on the unmasked mel, then the mask)
            for lo, hi in zip(bounds, bounds[1:]):
                run(lo, hi, ws, self._handle, postnet=False)
            _lib.check(lib.gvx_postnet_forward(self._handle, out["mel_outputs"].data_ptr(), None, B, T, out["mel_outputs_postnet"].data_ptr(),
                                               ws.data_ptr(), ws.numel(), self._stream()))
            if self.model_config.mask_padding:
                _lib.check(lib.gvx_mask_padding(out["mel_outputs"].data_ptr(), out["mel_outputs_postnet"].data_ptr(), out["gate_outputs"].data_ptr(),
                                                mel_len.data_ptr(), B, M, T, self._stream()))
            return out
        # chunks of near-equal size, alternating over two streams (rows never interact, so the split is invisible in the
        # results); the caller's stream waits for both lanes before anything downstream may touch the outputs
        n_chunks = -(-B // STREAM_ROWS)
        bounds = [(B * i) // n_chunks for i in range(n_chunks + 1)]
        self._ensure_lane_handles()   # a handle per lane, attention as a launch per step (see _ensure_lane_handles)
        need = lib.gvx_workspace_bytes(self._lane_handles[0], max(hi - lo for lo, hi in zip(bounds, bounds[1:])), L, T)
        streams, wss = self._lanes(dev, need)
        cur = torch.cuda.current_stream(dev)
        for lane in streams:
            lane.wait_stream(cur)
        for ci, (lo, hi) in enumerate(zip(bounds, bounds[1:])):
            with torch.cuda.stream(streams[ci % 2]):
                run(lo, hi, wss[ci % 2], self._lane_handles[ci % 2])
        for lane in streams:
            cur.wait_stream(lane)
        return out

    def _forward_train(self, batch: Dict[str, torch.Tensor]):
        """Training-mode forward (models/tts/tacotron2.py:450-481 under .train()).  Keep masks of every dropout may be given
        in ``batch["train_keep_masks"]`` (parity tests): dict with ``encoder`` [n_enc, B, E, L], ``attention_rnn`` [T, B, A],
        ``decoder_rnn`` [T, B, D], ``postnet`` list of [B, C_i, T]; otherwise they are drawn on the device from torch's
        generator.  Returns (outputs, tape) - the tape holds what genvox_amd.training's backward pieces need."""
        from . import training

        dev = self._require_gpu()
        mc = self.model_config
        self._ensure_packed()   # LSTM / attention / Prenet / projection weights come from the packed blob
        lib = _lib.load()
        tokens = batch["token_padded"].to(device=dev, dtype=torch.int64).contiguous()
        tok_len = batch["token_lengths"].to(device=dev, dtype=torch.int32).contiguous()
        mel_in = batch["mel_padded"].to(device=dev, dtype=torch.float32).contiguous()
        mel_len = batch["mel_lengths"].to(device=dev, dtype=torch.int32).contiguous()
        B, L = tokens.shape
        _, M, T = mel_in.shape
        E, A, D, P = mc.encoder_embedding_dim, mc.attention_rnn_dim, mc.decoder_rnn_dim, mc.prenet_dim
        given = batch.get("train_keep_masks") or {}
        draw = lambda shape, p: (torch.rand(shape, device=dev) >= p).to(torch.uint8)
        enc_keep = given.get("encoder")
        enc_keep = [draw((B, E, L), 0.5) for _ in range(mc.encoder_n_convolutions)] if enc_keep is None else list(enc_keep)
        att_keep = given.get("attention_rnn")
        att_keep = draw((T, B, A), mc.p_attention_dropout) if att_keep is None else att_keep
        dec_keep = given.get("decoder_rnn")
        dec_keep = draw((T, B, D), mc.p_decoder_dropout) if dec_keep is None else dec_keep
        post_keep = given.get("postnet")
        if post_keep is None:
            n = mc.postnet_n_convolutions
            post_keep = [draw((B, mc.postnet_embedding_dim if i < n - 1 else M, T), 0.5) for i in range(n)]
        att_keep = att_keep.to(device=dev, dtype=torch.uint8).contiguous()
        dec_keep = dec_keep.to(device=dev, dtype=torch.uint8).contiguous()
        pk = batch.get("prenet_keep_masks")
        masks = self._keep_masks(pk.to(device=dev, dtype=torch.uint8).reshape(2, T + 1, B, P).contiguous() if pk is not None else None,
                                 2 * (T + 1) * B * P, dev).reshape(2, T + 1, B, P)
        # encoder: embedding, training-mode convolution stack on the WHOLE batch (BatchNorm batch statistics), parameters in place
        # (nn.Embedding raises IndexError for an id outside the table, :459; checked here, before anything mutates model state -
        # an out-of-range torch index would trip a device-side assert and leave the HIP context unusable)
        if bool(((tokens < 0) | (tokens >= self.text_config.n_tokens)).any()):
            raise IndexError("genvox_amd: token id outside [0, n_tokens)")
        emb = self.embedding.weight.data[tokens].transpose(1, 2).contiguous()          # [B, E, L]  (:459)
        conv_out, enc_tape = training.convstack_train_forward(self, "encoder.convolutions", training.encoder_acts(self), emb, enc_keep)
        st = self._stream()
        dec_mel = torch.empty(B, M, T, device=dev)
        gate = torch.empty(B, T, device=dev)
        align = torch.empty(B, T, L, device=dev)
        ops = training._Ops(dev)
        # The recurrent part - BiLSTM and decoder loop - never mixes rows: batches above 32 rows (the reference's trainer
        # defaults to 64, its run.py uses 256) go through it in chunks of at most 32 rows, each with its own tape
        def run_chunks():
            chunks = []
            for lo in range(0, B, STREAM_ROWS):
                hi = min(B, lo + STREAM_ROWS)
                Bc = hi - lo
                ws = self._get_workspace(Bc, L, T)
                tl_c = tok_len[lo:hi].contiguous()
                memory = torch.empty(Bc, L, E, device=dev)
                enc_c = torch.empty(Bc, L, E, device=dev)
                _lib.check(lib.gvx_encoder_lstm_forward(self._handle, conv_out[lo:hi].data_ptr(), tl_c.data_ptr(), Bc, L, memory.data_ptr(),
                                                        enc_c.data_ptr(), None, ws.data_ptr(), ws.numel(), st))
                ha_blk = torch.empty((T + 1) * Bc * A, device=dev)
                c_a_all, c_d_all = torch.empty(T + 1, Bc, A, device=dev), torch.empty(T + 1, Bc, D, device=dev)
                hc_blk = torch.empty((T + 1) * Bc * (D + E), device=dev)
                pre_a, pre_d = torch.empty(T, Bc, A, 4, device=dev), torch.empty(T, Bc, D, 4, device=dev)   # gate pre-activations of both cells
                whole = lo == 0 and hi == B
                mk = masks if whole else masks[:, :, lo:hi].contiguous()
                ak = att_keep if whole else att_keep[:, lo:hi].contiguous()
                dk = dec_keep if whole else dec_keep[:, lo:hi].contiguous()
                _lib.check(lib.gvx_decoder_teacher_forced_train(
                    self._handle, memory.data_ptr(), tl_c.data_ptr(), Bc, L, mel_in[lo:hi].data_ptr(), T, mk.data_ptr(), ak.data_ptr(),
                    dk.data_ptr(), float(mc.p_attention_dropout), float(mc.p_decoder_dropout), dec_mel[lo:hi].data_ptr(), gate[lo:hi].data_ptr(),
                    align[lo:hi].data_ptr(), ha_blk.data_ptr(), c_a_all.data_ptr(), c_d_all.data_ptr(), hc_blk.data_ptr(), pre_a.data_ptr(), pre_d.data_ptr(),
                    ws.data_ptr(), ws.numel(), st))
                # the rest of the chunk's tape: per-step vectors as plain rows, the Prenet activations and the processed memory
                export = lambda what, shape: (lambda t: (_lib.check(lib.gvx_train_export(self._handle, ws.data_ptr(), ws.numel(), Bc, L, T, what, t.data_ptr(), st)), t)[1])(torch.empty(*shape, device=dev))
                chunks.append({"rows": (lo, hi), "token_lengths": tl_c, "memory": memory, "enc_cell_states": enc_c,
                               "h_a_all": training._unblock(ops, ha_blk, T + 1, Bc, A), "hc_all": training._unblock(ops, hc_blk, T + 1, Bc, D + E),
                               "c_a_all": c_a_all, "c_d_all": c_d_all, "pre_a": pre_a, "pre_d": pre_d, "frames": export(0, (T + 1, Bc, M)),
                               "p1": export(1, (T + 1, Bc, P)), "p2": export(2, (T + 1, Bc, P)), "pm": export(3, (Bc, L, mc.attention_dim)),
                               "att_keep": ak, "dec_keep": dk, "prenet_keep": mk})
            return chunks

        chunks = run_chunks()
        # A resident loop whose hand-off timed out left NaN in dec_mel: it must not reach the Postnet's BatchNorm statistics
        # (updated in place below).  One synchronisation; on a time-out the recurrent part runs again on the launch-per-step kernels
        if self._stalled_once():
            chunks = run_chunks()
            self._stalled_once()
        post_y, post_tape = training.convstack_train_forward(self, "postnet.convolutions", training.postnet_acts(self), dec_mel, post_keep)
        mel_out, mel_post = dec_mel.clone(), dec_mel + post_y
        if mc.mask_padding:
            pad = torch.arange(T, device=dev)[None, :] >= mel_len[:, None]
            mel_out.masked_fill_(pad[:, None, :], 0.0)
            mel_post.masked_fill_(pad[:, None, :], 0.0)
            gate.masked_fill_(pad, 1e3)
        outputs = {"mel_outputs": mel_out, "mel_outputs_postnet": mel_post, "gate_outputs": gate, "alignments": align}
        tape = {"encoder": enc_tape, "postnet": post_tape, "decoder_mel_unmasked": dec_mel, "enc_conv_out": conv_out, "chunks": chunks}
        # the HIP kernels updated the BatchNorm running statistics through raw pointers (no tensor version bump): the packed
        # blob - whose eval-mode convolutions have those statistics folded in - is stale now
        self._packed_key = None
        return outputs, tape

    def inference(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Autoregressive text->mel (reference: models/tts/tacotron2.py:483-499; Decoder.inference :390-414).

        inputs: ``tokens`` [B, L] (the reference: B = 1); optional ``token_lengths`` [B] for padded batches and
        ``prenet_keep_masks`` uint8 [2, max_decoder_steps, B, prenet_dim].  Outputs are trimmed to the longest row; for B > 1
        ``mel_lengths`` holds every row's frame count and frames past it carry the padding values of the reference's
        mask_padding (mel / mel_postnet / alignment 0, gate 1e3) - row b up to its length is what a batch-1 run gives.

        Batches above STREAM_ROWS rows are decoded as chunks of at most that many rows: one after the other when a chunk decodes as
        two resident kernels that fill the chip (``ar_loop_kind`` 2: default layer sizes, rows of <= 128 tokens), otherwise on two
        streams at once (each chunk a C-ABI call of its own, driven by its own host thread because the call polls the stop flags):
        a second independent batch then fills the latency gaps of the first."""
        dev = self._require_gpu()
        self._ensure_packed()
        tokens = inputs["tokens"].to(device=dev, dtype=torch.int64).contiguous()
        B, L = tokens.shape
        lens = inputs.get("token_lengths")
        lens_t = lens.to(device=dev, dtype=torch.int32).contiguous() if lens is not None else None
        mc = self.model_config
        M, E, P, S = self.audio_config.n_mels, mc.encoder_embedding_dim, mc.prenet_dim, mc.max_decoder_steps
        given = inputs.get("prenet_keep_masks")
        if given is not None:
            given = given.to(device=dev, dtype=torch.uint8).reshape(2, S, B, P)
        lib = _lib.load()
        mel_out = torch.zeros(B, M, S, device=dev)
        gate_out = torch.zeros(B, S, device=dev)
        align = torch.zeros(B, S, L, device=dev)
        n_frames = torch.zeros(B, dtype=torch.int32, device=dev)
        steps_run = []

        def run(lo: int, hi: int, handle: int, ws: torch.Tensor, seed) -> None:
            n = hi - lo
            if given is not None:
                masks = given[:, :, lo:hi].contiguous()
            else:
                masks = torch.empty(2 * S * n * P, dtype=torch.uint8, device=dev)
                _lib.check(lib.gvx_prenet_masks_generate(masks.data_ptr(), masks.numel(), seed, self._stream()))
            memory = torch.empty(n, L, E, device=dev)
            ln = lens_t[lo:hi].data_ptr() if lens_t is not None else None
            steps = C.c_int(0)
            st = self._stream()
            _lib.check(lib.gvx_encoder_forward(handle, tokens[lo:hi].data_ptr(), ln, n, L, memory.data_ptr(), ws.data_ptr(), ws.numel(), st))
            _lib.check(lib.gvx_decoder_autoregressive(
                handle, memory.data_ptr(), ln, n, L, S, float(mc.gate_threshold), masks.data_ptr(), mel_out[lo:hi].data_ptr(),
                gate_out[lo:hi].data_ptr(), align[lo:hi].data_ptr(), n_frames[lo:hi].data_ptr(), C.byref(steps), ws.data_ptr(),
                ws.numel(), st))
            steps_run.append(steps.value)

        n_chunks = -(-B // STREAM_ROWS)
        seeds = [int(torch.randint(0, 2 ** 62, (1,)).item()) for _ in range(n_chunks)]  # CPU generator: manual_seed repeats runs
        bounds = [(B * i) // n_chunks for i in range(n_chunks + 1)]
        chunks = list(zip(bounds, bounds[1:]))
        if n_chunks == 1:
            run(0, B, self._handle, self._workspace_of(lib.gvx_workspace_bytes_autoregressive(self._handle, B, L, S)), seeds[0])
        elif all(lib.gvx_autoregressive_loop_kind(self._handle, hi - lo, L) == 2 for lo, hi in chunks):
            # every chunk decodes as two resident kernels that fill the chip: one after the other on this stream (two such loops
            # cannot share the device; 2 x 25 us per step pair against 61 us for two concurrent lanes of launch-per-step loops)
            ws = self._workspace_of(lib.gvx_workspace_bytes_autoregressive(self._handle, max(hi - lo for lo, hi in chunks), L, S))
            for ci, (lo, hi) in enumerate(chunks):
                run(lo, hi, self._handle, ws, seeds[ci])
        else:
            import threading

            need = lib.gvx_workspace_bytes_autoregressive(self._handle, max(hi - lo for lo, hi in chunks), L, S)
            streams, wss = self._lanes(dev, need)
            self._ensure_lane_handles()
            cur = torch.cuda.current_stream(dev)
            errors = []

            def lane_main(lane: int) -> None:
                try:
                    with torch.cuda.device(dev), torch.cuda.stream(streams[lane]):
                        for ci in range(lane, n_chunks, 2):
                            run(chunks[ci][0], chunks[ci][1], self._lane_handles[lane], wss[lane], seeds[ci])
                except BaseException as e:  # re-raised on the calling thread
                    errors.append(e)

            for st in streams:
                st.wait_stream(cur)
            threads = [threading.Thread(target=lane_main, args=(i,)) for i in range(2)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            for st in streams:
                cur.wait_stream(st)
            if errors:
                raise errors[0]
        n_host = n_frames.cpu()
        if not inputs.get("_retried") and self._stalled_once():   # (a resident encoder / attention kernel could not overlap its partners)
            return self.inference({**inputs, "_retried": True})
        self.check_status()   # the decoder calls have synchronised already: a bad token id raises here like nn.Embedding does
        Tn = int(n_host.max())
        if max(steps_run) >= S and int((n_host >= S).sum()) > 0:
            print("Warning! Reached max decoder steps")
        mel_out = mel_out[:, :, :Tn].contiguous()
        mel_post = torch.empty_like(mel_out)
        # every row is post-processed as a sequence of its own length (what a batch-1 run of the reference sees)
        ws = self._workspace_of(lib.gvx_postnet_workspace_bytes(self._handle, B, Tn))
        _lib.check(lib.gvx_postnet_forward(self._handle, mel_out.data_ptr(), n_frames.data_ptr() if B > 1 else None, B, Tn,
                                           mel_post.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()))
        out = {"mel_outputs": mel_out, "mel_outputs_postnet": mel_post,
               "gate_outputs": gate_out[:, :Tn].contiguous(), "alignments": align[:, :Tn].contiguous()}
        if B > 1:
            out["mel_lengths"] = n_frames
        return out

    # ---- stage-level entry points (used by the parity tests to localise failures)
    def encode(self, tokens: torch.Tensor, token_lengths: Optional[torch.Tensor]) -> torch.Tensor:
        dev = self._require_gpu()
        self._ensure_packed()
        tokens = tokens.to(device=dev, dtype=torch.int64).contiguous()
        B, L = tokens.shape
        lens = token_lengths.to(device=dev, dtype=torch.int32).contiguous() if token_lengths is not None else None
        ws = self._get_workspace(B, L, 1)
        memory = torch.empty(B, L, self.model_config.encoder_embedding_dim, device=dev)
        _lib.check(_lib.load().gvx_encoder_forward(self._handle, tokens.data_ptr(), lens.data_ptr() if lens is not None else None,
                                                   B, L, memory.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()))
        return memory

    def postnet_residual(self, mel: torch.Tensor, mel_lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        """mel + Postnet(mel) for any batch size; with ``mel_lengths`` every row is treated as a sequence of its own length."""
        dev = self._require_gpu()
        self._ensure_packed()
        mel = mel.to(device=dev, dtype=torch.float32).contiguous()
        B, M, T = mel.shape
        lens = mel_lengths.to(device=dev, dtype=torch.int32).contiguous() if mel_lengths is not None else None
        lib = _lib.load()
        ws = self._workspace_of(lib.gvx_postnet_workspace_bytes(self._handle, B, T))
        out = torch.empty_like(mel)
        _lib.check(lib.gvx_postnet_forward(self._handle, mel.data_ptr(), lens.data_ptr() if lens is not None else None, B, T,
                                           out.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()))
        return out

    # ---- reference contract: batches, checkpoints, configs (models/tts/__init__.py:91-95, tacotron2.py:574-596)
    def prepare_batch(self, batch: Dict[str, torch.Tensor], device: str) -> Dict[str, torch.Tensor]:
        for key, val in batch.items():
            batch[key] = val.to(device=device)
        return batch

    def get_checkpoint_statedicts(self, optimizer: Optional[Dict] = None) -> Dict:
        statedicts = {"model_statedict": self.state_dict()}
        if optimizer is not None:
            statedicts["optim_statedict"] = optimizer["optimizer"].state_dict()
        return statedicts

    def load_checkpoint_statedicts(self, statedicts: Dict, save_optimizer_dict: bool = False, optimizer: Optional[Dict] = None) -> None:
        self.load_state_dict(statedicts["model_statedict"])
        if save_optimizer_dict:
            optimizer["optimizer"].load_state_dict(statedicts["optim_statedict"])

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        out = super().load_state_dict(state_dict, strict=strict, assign=assign)
        self._packed_key = None
        self._key_tensors = None
        return out

    @staticmethod
    def load_from_config(config_path: str) -> "Tacotron2":
        configs = BaseConfig.load_configs_from_file(
            path=config_path,
            config_map={"text_config": TextConfig, "audio_config": AudioConfig, "model_config": Tacotron2Config})
        return Tacotron2(**configs)

    def get_optimizer(self) -> Dict:
        """Reference: tacotron2.py:506-513 (torch.optim.Adam over all parameters with the config's learning rate and weight
        decay); here genvox_amd.training.Adam, the same update as one HIP kernel per tensor."""
        from . import training

        return {"optimizer": training.Adam(self, lr=self.model_config.learning_rate, weight_decay=self.model_config.weight_decay)}

    def train_step(self, batch: Dict, criterion: Optional[Dict] = None, optimizer: Optional[Dict] = None) -> None:
        """One training step (reference: tacotron2.py:515-522): training-mode forward, Tacotron2Loss, backward through the whole
        model, clip_grad_norm_, Adam.  No autograd: the backward is explicit (genvox_amd.training.train_backward, HIP kernels
        behind the C ABI; the formulas are pinned to the reference's loss.backward() through oracle/train_ref.py).  Fills
        ``loss_items`` and ``grad_norm_val`` like the reference.  Any batch size (the recurrent part runs in chunks of at most
        32 rows); both recurrences are walked back inside single C-ABI calls and the packed blob is re-built on the device:
        27 ms per step at 32 x 200 frames, reproducible bit for bit (profiles/r04_train_step_timing.txt)."""
        from . import training

        if optimizer is None:
            raise ValueError("train_step needs the optimizer dict of get_optimizer()")
        was_training = self.training   # (the reference's train_step does not switch modes: its trainer calls .train() once)
        self.train()
        try:
            outputs, tape = self._forward_train(batch)
            loss = (criterion or self.get_criterion())["loss"](batch, outputs)
            self.loss_items = {key: val.item() for key, val in loss.items()}
            # (a token id outside the table raised before the forward touched anything; a hand-off time-out was caught - and the
            # recurrent part re-run - before the Postnet's BatchNorm update; this is the last look before gradients reach the weights)
            self.check_status()
            self._enc_bptt_workspaces = []
            grads = training.train_backward(self, batch, outputs, tape)
            self.grad_norm_val, scale = training.clip_grad_norm(grads, self.model_config.grad_clip_thresh)
            if not math.isfinite(self.grad_norm_val) and training.encoder_bptt_timed_out(self):
                # the resident walk of the encoder BiLSTM gave up on a hand-off (NaN gradients): from now on a launch per time
                # step for this model, and the backward once more - nothing has touched the weights yet
                warnings.warn("genvox_amd: the resident encoder-BiLSTM backward timed out; re-running the backward with a launch per time step")
                self._enc_bptt_resident = False
                grads = training.train_backward(self, batch, outputs, tape)
                self.grad_norm_val, scale = training.clip_grad_norm(grads, self.model_config.grad_clip_thresh)
            self._enc_bptt_workspaces = []
            optimizer["optimizer"].step(grads, scale)
            self._packed_key = None
            self.last_grads = grads   # (kept for inspection / tests; the reference keeps them in .grad)
        finally:
            if not was_training:
                self.eval()

    def get_train_step_logs(self) -> Dict:
        """Reference: tacotron2.py:553-560."""
        return {**self.loss_items, "grad_norm": self.grad_norm_val}

    def get_criterion(self) -> Dict:
        """Reference: tacotron2.py:501-504."""
        return {"loss": Tacotron2Loss}

    def eval_step(self, batch: Dict, criterion: Optional[Dict] = None, eval_outdir: Optional[str] = None) -> Dict[str, torch.Tensor]:
        """Forward + criterion under no_grad (reference: tacotron2.py:524-529): fills `loss_items_eval` with
        loss_eval / mel_loss_eval / gate_loss_eval like the reference and returns the forward outputs.  The plots the
        reference writes to `eval_outdir` are out of scope."""
        with torch.no_grad():
            outputs = self.forward(batch)
            loss = (criterion or self.get_criterion())["loss"](batch, outputs)
        self.loss_items_eval = {(key + "_eval"): val.item() for key, val in loss.items()}
        self.check_status()   # the .item() calls above have synchronised: a bad token id raises here like nn.Embedding does
        return outputs

    def get_eval_priority(self) -> float:
        """Reference: tacotron2.py:550-551."""
        return self.loss_items_eval["loss_eval"]


def Tacotron2Loss(batch: Dict[str, torch.Tensor], outputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Criterion of the evaluation step, same name / arguments / result keys as the reference's Tacotron2Loss
    (models/tts/tacotron2.py:598-615): {"loss", "mel_loss", "gate_loss"} as 0-dim float32 tensors on the outputs'
    device.  One HIP reduction (gvx_tacotron2_loss); forward only - there is no backward on this path."""
    lib = _lib.load()
    mel = outputs["mel_outputs"]
    dev = mel.device
    if dev.type != "cuda":
        raise RuntimeError("genvox_amd.Tacotron2Loss needs the outputs of the GPU forward path (cuda tensors)")
    B, M, T = mel.shape
    f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
    mel, post, gate = f32(mel), f32(outputs["mel_outputs_postnet"]), f32(outputs["gate_outputs"])
    mel_t, gate_t = f32(batch["mel_padded"]), f32(batch["gate_padded"])
    if mel_t.shape != mel.shape or gate_t.shape != gate.shape:
        raise ValueError(f"target shapes {tuple(mel_t.shape)}, {tuple(gate_t.shape)} do not match the outputs {tuple(mel.shape)}, {tuple(gate.shape)}")
    out = torch.empty(3, dtype=torch.float32, device=dev)
    scratch = torch.empty(6144, dtype=torch.uint8, device=dev)   # float64 partial sums of the reduction
    _lib.check(lib.gvx_tacotron2_loss(mel.data_ptr(), post.data_ptr(), gate.data_ptr(), mel_t.data_ptr(), gate_t.data_ptr(),
                                      B, M, T, out.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                      torch.cuda.current_stream(dev).cuda_stream))
    return {"loss": out[0], "mel_loss": out[1], "gate_loss": out[2]}
