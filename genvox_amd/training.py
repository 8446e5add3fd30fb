"""The training step of the MI355X path - SURVEY.md section 8f rank 4 (the reference's ``Tacotron2.train_step``,
models/tts/tacotron2.py:515-522: forward under ``.train()``, ``Tacotron2Loss``, ``loss.backward()``, ``clip_grad_norm_``,
``Adam.step``).

There is no autograd here.  The forward (``Tacotron2._forward_train``) records a tape; ``train_backward`` walks the model
backwards in explicit formulas - criterion, Postnet, projections, BPTT over the decoder loop (both LSTM cells with their
output dropout, the location-sensitive attention with its previous / cumulative-weights path), Prenet, encoder BiLSTM with
packed-sequence semantics, encoder convolutions, embedding - exactly as ``oracle/train_ref.py`` states them, and every
formula runs in a HIP kernel behind the C ABI (csrc/train.hip: ``gvx_conv_bn_act_train_*``, ``gvx_train_*``; the dense
products on the exact-fp32 MFMA GEMM of the forward path).  The two recurrences are one C-ABI call each
(``gvx_train_decoder_bptt``: three launches per decoder step, ``gvx_train_encoder_lstm_bptt``: one per time step); torch
allocates, slices and reshapes around them.  Parameters are updated in
place in the reference's own layouts (``Adam``); the packed blob of the forward kernels is re-built before the next forward.

Pinned by ``tests/golden/train_small.npz`` - the reference's own train step (all 48 parameter gradients, the gradient norm,
every parameter after the Adam update) - and against the oracle at the default layer sizes (tests/test_training_gpu.py).
Dropout keep masks are explicit inputs where parity is tested (uint8, the shape of the dropout's input) and drawn on the
device otherwise.  32 x 200 frames x 128 tokens: 0.54 s per step with the first version (host-strung primitives, host
re-packing), 0.031 s now (profiles/r03_train_step_timing.txt).
"""
from __future__ import annotations

import contextlib
import os
import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib

ACT = {"none": 0, "relu": 1, "tanh": 2}


class ConvStackTape:
    """What the backward of a stack needs: per layer the C library's saved buffer, plus shapes and masks."""

    def __init__(self):
        self.layers: List[dict] = []


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stack_layers(model, prefix: str, n: int):
    sd = dict(model.named_parameters())
    buf = dict(model.named_buffers())
    out = []
    for i in range(n):
        out.append(dict(w=sd[f"{prefix}.{i}.0.conv.weight"], b=sd[f"{prefix}.{i}.0.conv.bias"], gamma=sd[f"{prefix}.{i}.1.weight"],
                        beta=sd[f"{prefix}.{i}.1.bias"], rmean=buf[f"{prefix}.{i}.1.running_mean"], rvar=buf[f"{prefix}.{i}.1.running_var"],
                        name=f"{prefix}.{i}"))
    return out


def convstack_train_forward(model, prefix: str, acts: Sequence[str], x: torch.Tensor, keeps: Sequence[Optional[torch.Tensor]],
                            p_drop: float = 0.5, update_running_stats: bool = True) -> Tuple[torch.Tensor, ConvStackTape]:
    """x [B, Cin, T] (cuda, fp32) through ``len(acts)`` layers ``{prefix}.{i}`` of ``model`` in training mode."""
    lib = _lib.load()
    dev = x.device
    if dev.type != "cuda":
        raise RuntimeError("genvox_amd training ops run on an MI355X only (no CPU fallback)")
    st = torch.cuda.current_stream(dev).cuda_stream
    tape = ConvStackTape()
    x = x.to(torch.float32).contiguous()
    for lay, act, keep in zip(_stack_layers(model, prefix, len(acts)), acts, keeps):
        B, Cin, T = x.shape
        Cout, _, k = lay["w"].shape
        keep_t = None if keep is None else keep.to(device=dev, dtype=torch.uint8).contiguous()
        saved = torch.empty(lib.gvx_conv_train_saved_bytes(B, Cin, Cout, T, k), dtype=torch.uint8, device=dev)
        ws = torch.empty(lib.gvx_conv_train_workspace_bytes(B, Cin, Cout, T, k), dtype=torch.uint8, device=dev)
        y = torch.empty(B, Cout, T, device=dev)
        _lib.check(lib.gvx_conv_bn_act_train_forward(
            _p(x), _p(lay["w"].data), _p(lay["b"].data), _p(lay["gamma"].data), _p(lay["beta"].data),
            _p(lay["rmean"]) if update_running_stats else None, _p(lay["rvar"]) if update_running_stats else None,
            B, Cin, Cout, T, k, ACT[act], _p(keep_t), float(p_drop), _p(y), _p(saved), saved.numel(), _p(ws), ws.numel(), st))
        tape.layers.append(dict(saved=saved, shape=(B, Cin, Cout, T, k), act=act, keep=keep_t, p=p_drop, lay=lay))
        x = y
    return x, tape


def convstack_train_backward(tape: ConvStackTape, dy: torch.Tensor, first_layer_wgrad_input: Optional[torch.Tensor] = None
                             ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """Gradients of every parameter of the stack (keys = the reference's state_dict names) and of its input."""
    lib = _lib.load()
    dev = dy.device
    st = torch.cuda.current_stream(dev).cuda_stream
    grads: Dict[str, torch.Tensor] = {}
    dy = dy.to(torch.float32).contiguous()
    for li in reversed(range(len(tape.layers))):
        L = tape.layers[li]
        B, Cin, Cout, T, k = L["shape"]
        lay = L["lay"]
        ws = torch.empty(lib.gvx_conv_train_workspace_bytes(B, Cin, Cout, T, k), dtype=torch.uint8, device=dev)
        dx = torch.empty(B, Cin, T, device=dev)
        dw, db = torch.empty_like(lay["w"].data), torch.empty_like(lay["b"].data)
        dg, dbe = torch.empty_like(lay["gamma"].data), torch.empty_like(lay["beta"].data)
        xw = first_layer_wgrad_input.to(device=dev, dtype=torch.float32).contiguous() if (li == 0 and first_layer_wgrad_input is not None) else None
        _lib.check(lib.gvx_conv_bn_act_train_backward(
            _p(dy), _p(L["saved"]), L["saved"].numel(), _p(lay["w"].data), _p(lay["gamma"].data), _p(xw), B, Cin, Cout, T, k,
            ACT[L["act"]], _p(L["keep"]), float(L["p"]), _p(dx), _p(dw), _p(db), _p(dg), _p(dbe), _p(ws), ws.numel(), st))
        n = lay["name"]
        grads[f"{n}.0.conv.weight"], grads[f"{n}.0.conv.bias"] = dw, db
        grads[f"{n}.1.weight"], grads[f"{n}.1.bias"] = dg, dbe
        dy = dx
    return dy, grads


def encoder_acts(model) -> List[str]:
    return ["relu"] * model.model_config.encoder_n_convolutions          # models/tts/tacotron2.py:234-235


def postnet_acts(model) -> List[str]:
    n = model.model_config.postnet_n_convolutions
    return ["tanh"] * (n - 1) + ["none"]                                  # models/tts/tacotron2.py:194-199


def loss_backward(batch: Dict[str, torch.Tensor], outputs: Dict[str, torch.Tensor]):
    """d loss / d (mel_outputs [its own MSE term], mel_outputs_postnet, gate_outputs) of ``Tacotron2Loss``
    (models/tts/tacotron2.py:598-615) on the outputs of the forward (which carry the reference's padding values)."""
    lib = _lib.load()
    mel = outputs["mel_outputs"]
    dev = mel.device
    f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
    mel, post, gate = f32(mel), f32(outputs["mel_outputs_postnet"]), f32(outputs["gate_outputs"])
    mel_t, gate_t = f32(batch["mel_padded"]), f32(batch["gate_padded"])
    B, M, T = mel.shape
    dmel, dpost, dgate = torch.empty_like(mel), torch.empty_like(post), torch.empty_like(gate)
    _lib.check(lib.gvx_tacotron2_loss_backward(_p(mel), _p(post), _p(gate), _p(mel_t), _p(gate_t), B, M, T, _p(dmel), _p(dpost), _p(dgate),
                                               torch.cuda.current_stream(dev).cuda_stream))
    return dmel, dpost, dgate


def postnet_train_step_slice(model, decoder_mel: torch.Tensor, batch: Dict[str, torch.Tensor], gate_outputs: torch.Tensor,
                             keeps: Sequence[torch.Tensor]):
    """The tail of the reference's training step that is complete on this side: Postnet in training mode on the decoder's
    (unmasked) mel output, residual add, padding mask (models/tts/tacotron2.py:463-473), criterion backward, Postnet backward.
    Returns (outputs, parameter gradients of the Postnet, d loss / d decoder mel output summed over all three paths,
    d loss / d gate_outputs) - the last two are what a decoder BPTT would start from."""
    dev = decoder_mel.device
    y, tape = convstack_train_forward(model, "postnet.convolutions", postnet_acts(model), decoder_mel, keeps)
    mel = decoder_mel.clone()
    post = decoder_mel + y
    gate = gate_outputs.to(dev).clone()
    if model.model_config.mask_padding:
        lens = batch["mel_lengths"].to(dev)
        pad = torch.arange(mel.shape[2], device=dev)[None, :] >= lens[:, None]       # [B, T]
        mel.masked_fill_(pad[:, None, :], 0.0)
        post.masked_fill_(pad[:, None, :], 0.0)
        gate.masked_fill_(pad, 1e3)
    outputs = {"mel_outputs": mel, "mel_outputs_postnet": post, "gate_outputs": gate}
    dmel, dpost, dgate = loss_backward(batch, outputs)
    # the reference masks mel_outputs IN PLACE (.data) after the Postnet has run: autograd's saved input of the first Postnet
    # convolution is that storage, so its weight gradient is computed from the masked tensor
    dx, grads = convstack_train_backward(tape, dpost, first_layer_wgrad_input=mel)
    return outputs, grads, dmel + dpost + dx, dgate


# =====================================================================================================================
# The whole backward of the training step (loss.backward() of the reference, models/tts/tacotron2.py:519): host side.
# Walks the decoder loop and the encoder BiLSTM backwards exactly as oracle/train_ref.py::train_backward states it; every
# formula runs in a HIP kernel behind the C ABI (csrc/train.hip: gvx_train_*), torch only allocates, slices and reshapes.
# =====================================================================================================================
class _Ops:
    """Thin wrappers over the gvx_train_* primitives (row-major fp32 tensors whose last stride is 1)."""

    def __init__(self, dev):
        self.lib = _lib.load()
        self.dev = dev
        self.st = torch.cuda.current_stream(dev).cuda_stream
        self.scratch = torch.empty(8 << 20, device=dev)   # 32 MB of split-K partial tiles for the small-batch products

    @staticmethod
    def _ld(t: torch.Tensor) -> int:
        assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32, (t.shape, t.stride(), t.dtype)
        return t.stride(0)

    def new(self, *shape):
        return torch.empty(*shape, device=self.dev)

    def zeros(self, *shape):
        return torch.zeros(*shape, device=self.dev)

    def gemm_nt(self, A, W, bias=None, out=None):
        """out[m][n] = sum_k A[m][k] W[n][k] (+ bias[n]);  K % 4 == 0."""
        M, K = A.shape
        N, K2 = W.shape
        assert K == K2 and K % 4 == 0, (A.shape, W.shape)
        out = self.new(M, N) if out is None else out
        _lib.check(self.lib.gvx_train_gemm_nt(_p(A), self._ld(A), _p(W), self._ld(W), _p(out), self._ld(out), M, N, K, _p(bias),
                                              _p(self.scratch), self.scratch.numel() * 4, self.st))
        return out

    def transpose(self, X, pad_to: int = 4):
        """X [R, C] (dense) -> [C, R_p] with R_p = R rounded up to a multiple of pad_to (zero filled)."""
        X = X.contiguous()
        R, Cc = X.shape
        Rp = -(-R // pad_to) * pad_to
        out = self.new(Cc, Rp)
        _lib.check(self.lib.gvx_train_transpose(_p(X), Cc, _p(out), R, Cc, Rp, self.st))
        return out

    def mm_tn(self, A, Bm):
        """A^T @ Bm for A [R, M], Bm [R, N] -> [M, N] (the weight-gradient form: the K-major GEMM reads both as they lie)."""
        R, M = A.shape
        R2, N = Bm.shape
        assert R == R2, (A.shape, Bm.shape)
        out = self.new(M, N)
        _lib.check(self.lib.gvx_train_gemm_tn(_p(A), self._ld(A), _p(Bm), self._ld(Bm), _p(out), N, M, N, R, _p(self.scratch),
                                              self.scratch.numel() * 4, self.st))
        return out

    def colsum(self, X):
        X = X.contiguous()
        out = self.new(X.shape[1])
        _lib.check(self.lib.gvx_train_colsum(_p(X), X.shape[0], X.shape[1], _p(out), self.st))
        return out

    def axpby(self, a, alpha=1.0, b=None, beta=1.0, out=None):
        rows, cols = a.shape
        out = self.new(rows, cols) if out is None else out
        _lib.check(self.lib.gvx_train_axpby(_p(a), self._ld(a), float(alpha), _p(b), self._ld(b) if b is not None else 0, float(beta),
                                            _p(out), self._ld(out), rows, cols, self.st))
        return out


_SIDE: Dict[str, object] = {}


def _side_stream(dev):
    """The second stream of the training backward (one per device), or None when switched off."""
    if os.environ.get("GVX_TRAIN_SIDE_STREAM", "1") == "0":
        return None
    key = ("stream", str(dev))
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


def _side_ops(dev) -> "_Ops":
    """_Ops bound to the side stream (own split-K scratch); call with that stream current."""
    key = ("ops", str(dev))
    if key not in _SIDE:
        _SIDE[key] = _Ops(dev)
    return _SIDE[key]


def _unblock(ops: _Ops, blocked: torch.Tensor, n_slots: int, B: int, K: int) -> torch.Tensor:
    out = ops.new(n_slots, B, K)
    _lib.check(ops.lib.gvx_train_unblock(_p(blocked), _p(out), n_slots, B, K, ops.st))
    return out


def _accumulate(ops: "_Ops", total: Dict[str, torch.Tensor], part: Dict[str, torch.Tensor]) -> None:
    """total[k] += part[k] (gradients of the row chunks are added in chunk order)."""
    for k, v in part.items():
        if k in total:
            t = total[k]
            ops.axpby(t.reshape(1, -1), 1.0, v.reshape(1, -1), 1.0, out=t.reshape(1, -1))
        else:
            total[k] = v.contiguous()


def _recurrent_backward(model, ops: "_Ops", ch: dict, dmel: torch.Tensor, dgate: torch.Tensor, x: torch.Tensor, align: torch.Tensor
                        ) -> Tuple[Dict[str, torch.Tensor], torch.Tensor]:
    """The part of the backward that never mixes batch rows, for one chunk of at most 32 rows: projection, decoder loop
    (gvx_train_decoder_bptt), Prenet, memory layer, encoder BiLSTM (gvx_train_encoder_lstm_bptt).  dmel [Bc, M, T] and dgate
    [Bc, T] are d loss / d of the decoder's mel / gate outputs, x [Bc, L, E] the BiLSTM's input, align [Bc, T, L] the
    alignments.  Returns (parameter gradients of this chunk's rows, d loss / d x as [Bc * L, E])."""
    mc = model.model_config
    dev = dmel.device
    lib, st = ops.lib, ops.st
    P_ = dict(model.named_parameters())
    W = lambda name: P_[name].data
    att = "decoder.attention_layer."
    B, M, T = dmel.shape
    L = x.shape[1]
    A, D, E, Pn, a = mc.attention_rnn_dim, mc.decoder_rnn_dim, mc.encoder_embedding_dim, mc.prenet_dim, mc.attention_dim
    F_, kl = mc.attention_location_n_filters, mc.attention_location_kernel_size
    g: Dict[str, torch.Tensor] = {}
    # ---- projection: rows (t, b); [dmel | dgate | 0-pad] [T B, Mp] x [Wp ; Wg]
    Mp = -(-(M + 1) // 4) * 4
    dmg = ops.zeros(T * B, Mp)
    dmg[:, :M] = dmel.permute(2, 0, 1).reshape(T * B, M)          # (torch: layout plumbing only)
    dmg[:, M] = dgate.t().reshape(T * B)
    hc = ch["hc_all"]                                             # [T+1, B, D+E] rows: slot t + 1 = [h_d(t) ; ctx(t)]
    hc_t = hc[1:].reshape(T * B, D + E)
    Wpg = ops.zeros(Mp, D + E)
    Wpg[:M] = W("decoder.linear_projection.linear_layer.weight")
    Wpg[M] = W("decoder.gate_layer.linear_layer.weight")[0]
    dWpg = ops.mm_tn(dmg, hc_t)                                   # [Mp, D+E]
    g["decoder.linear_projection.linear_layer.weight"], g["decoder.gate_layer.linear_layer.weight"] = dWpg[:M].contiguous(), dWpg[M:M + 1].contiguous()
    db = ops.colsum(dmg)
    g["decoder.linear_projection.linear_layer.bias"], g["decoder.gate_layer.linear_layer.bias"] = db[:M].contiguous(), db[M:M + 1].contiguous()
    dhc_all = ops.gemm_nt(dmg, ops.transpose(Wpg)).reshape(T, B, D + E)       # d loss / d [h_d(t) ; ctx(t)] through the projection

    # ---- operands of the weight gradients (inputs of both cells at every step) and the attention queries
    p2 = ch["p2"]                                                 # [T+1, B, P]
    ha = ch["h_a_all"]                                            # [T+1, B, A]  slot t + 1 = dropped h_a(t)
    ctx_prev = hc[:T, :, D:]                                      # ctx(t-1), slot t
    xa = torch.cat((p2[:T], ctx_prev), dim=2).reshape(T * B, Pn + E).contiguous()
    ha_prev = ha[:T].reshape(T * B, A)
    Wia, Wha = W("decoder.attention_rnn.weight_ih"), W("decoder.attention_rnn.weight_hh")
    Wid, Whd = W("decoder.decoder_rnn.weight_ih"), W("decoder.decoder_rnn.weight_hh")
    xd = torch.cat((ha[1:], hc[1:, :, D:]), dim=2).reshape(T * B, A + E).contiguous()   # [h_a(t) ; ctx(t)]
    hd_prev = hc[:T, :, :D].reshape(T * B, D).contiguous()
    pre_a, pre_d = ch["pre_a"], ch["pre_d"]                       # [T, B, H, 4]: the gate pre-activations the forward computed
    wq, v = W(att + "query_layer.linear_layer.weight"), W(att + "v.linear_layer.weight")
    lw, ld = W(att + "location_layer.location_conv.conv.weight").contiguous(), W(att + "location_layer.location_dense.linear_layer.weight")
    wm = W(att + "memory_layer.linear_layer.weight")
    q_all = ops.gemm_nt(ha[1:].reshape(T * B, A), wq).reshape(T, B, a)
    memory, pm = ch["memory"], ch["pm"]
    w_all = align.permute(1, 0, 2).contiguous()                   # [T, B, L]
    c_a, c_d = ch["c_a_all"], ch["c_d_all"]                       # [T+1, B, H]
    # ---- back-propagation through the decoder loop: one C-ABI call (three launches per step, csrc/train.hip)
    dga_all, dgd_all = ops.new(T, B, 4 * A), ops.new(T, B, 4 * D)
    dq_all, dctx_all = ops.new(T, B, a), ops.new(T, B, E)
    dpm, dmemory = ops.new(B, L, a), ops.new(B, L, E)
    dv, dld, dlw = ops.new(a), ops.new(a, F_), ops.new(F_, 2, kl)
    ba_ = _lib.gvx_bptt_decoder_args()
    ba_.B, ba_.L, ba_.T, ba_.A, ba_.D, ba_.E, ba_.P, ba_.a, ba_.F, ba_.kl = B, L, T, A, D, E, Pn, a, F_, kl
    ba_.att_scale, ba_.dec_scale = 1.0 / (1.0 - mc.p_attention_dropout), 1.0 / (1.0 - mc.p_decoder_dropout)
    keep = []   # (tensors whose only reference would be the raw pointer in the argument block)
    def ptr(t):
        t = t.contiguous()
        keep.append(t)
        return t.data_ptr()
    ba_.dhc_all, ba_.pre_a, ba_.pre_d = ptr(dhc_all), ptr(pre_a), ptr(pre_d)
    ba_.c_a_all, ba_.c_d_all = ptr(c_a), ptr(c_d)
    ba_.att_keep, ba_.dec_keep, ba_.q_all = ptr(ch["att_keep"]), ptr(ch["dec_keep"]), ptr(q_all)
    ba_.ctx_all = hc.data_ptr() + 4 * (B * (D + E) + D)          # ctx(t) = hc[t + 1, b, D:]
    ba_.ctx_ts, ba_.ctx_bs = B * (D + E), D + E
    ba_.w_all, ba_.memory, ba_.pm = ptr(w_all), ptr(memory), ptr(pm)
    ba_.w_ih_a, ba_.w_hh_a, ba_.w_ih_d, ba_.w_hh_d = ptr(Wia), ptr(Wha), ptr(Wid), ptr(Whd)
    ba_.wq, ba_.v, ba_.loc_conv, ba_.loc_dense = ptr(wq), ptr(v), ptr(lw), ptr(ld)
    ba_.dga_all, ba_.dgd_all, ba_.dq_all, ba_.dctx_all = dga_all.data_ptr(), dgd_all.data_ptr(), dq_all.data_ptr(), dctx_all.data_ptr()
    ba_.dpm, ba_.dmemory, ba_.dv, ba_.dloc_dense, ba_.dloc_conv = dpm.data_ptr(), dmemory.data_ptr(), dv.data_ptr(), dld.data_ptr(), dlw.data_ptr()
    wsb = lib.gvx_train_decoder_bptt_workspace_bytes(C.byref(ba_))
    if wsb == 0:
        _lib.check(1)
    ws_bptt = torch.empty(wsb, dtype=torch.uint8, device=dev)
    _lib.check(lib.gvx_train_decoder_bptt(C.byref(ba_), ws_bptt.data_ptr(), wsb, st))
    # ---- Off the chain to the encoder: the loop's weight gradients and the whole Prenet backward (dense products, ~5 ms at
    # 32 x 200 frames) go to a second stream; the caller's stream goes on with the memory gradient and the encoder's BiLSTM walk
    # (a launch per position on a quarter of the chip) and waits for them at the end.  One fork and one join per chunk (an edge
    # costs ~5 us: tools/micro/chain_sidestream_bench.hip); every tensor is produced by the same launches in the same order as on
    # one stream, so results do not change.  GVX_TRAIN_SIDE_STREAM=0: everything on the caller's stream.
    main_stream = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    if side is not None:
        fork = torch.cuda.Event(); fork.record(main_stream)
        side.wait_event(fork)
    with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
        ops_s = _side_ops(dev) if side is not None else ops
        st_s = ops_s.st
        dpm2 = dpm.reshape(B * L, a)
        # the Prenet columns of the attention LSTM are not on the recurrence: one product over all steps
        dp2 = ops_s.zeros(T + 1, B, Pn)
        ops_s.gemm_nt(dga_all.reshape(T * B, 4 * A), ops_s.transpose(Wia)[:Pn], out=dp2[:T].reshape(T * B, Pn))
        # ---- weight gradients of the loop, one product over all (t, b) rows each
        dga2, dgd2 = dga_all.reshape(T * B, 4 * A), dgd_all.reshape(T * B, 4 * D)
        g["decoder.attention_rnn.weight_ih"], g["decoder.attention_rnn.weight_hh"] = ops_s.mm_tn(dga2, xa), ops_s.mm_tn(dga2, ha_prev.contiguous())
        g["decoder.attention_rnn.bias_ih"] = ops_s.colsum(dga2)
        g["decoder.attention_rnn.bias_hh"] = g["decoder.attention_rnn.bias_ih"].clone()
        g["decoder.decoder_rnn.weight_ih"], g["decoder.decoder_rnn.weight_hh"] = ops_s.mm_tn(dgd2, xd), ops_s.mm_tn(dgd2, hd_prev)
        g["decoder.decoder_rnn.bias_ih"] = ops_s.colsum(dgd2)
        g["decoder.decoder_rnn.bias_hh"] = g["decoder.decoder_rnn.bias_ih"].clone()
        g[att + "query_layer.linear_layer.weight"] = ops_s.mm_tn(dq_all.reshape(T * B, a), ha[1:].reshape(T * B, A).contiguous())
        g[att + "v.linear_layer.weight"] = dv[None, :]
        g[att + "location_layer.location_dense.linear_layer.weight"] = dld
        g[att + "location_layer.location_conv.conv.weight"] = dlw
        g[att + "memory_layer.linear_layer.weight"] = ops_s.mm_tn(dpm2, memory.reshape(B * L, E))
        # ---- Prenet (relu then dropout, twice; models/tts/tacotron2.py:140-144)
        pk = ch["prenet_keep"]                                        # [2, T+1, B, P] uint8
        p1 = ch["p1"]
        w1 = W("decoder.prenet.layers.1.linear_layer.weight")
        n_rows = (T + 1) * B
        dz2 = ops_s.new(n_rows, Pn)
        _lib.check(lib.gvx_train_relu_dropout_backward(_p(dp2), _p(p2.contiguous()), _p(pk[1].contiguous()), 2.0, n_rows * Pn, _p(dz2), st_s))
        g["decoder.prenet.layers.1.linear_layer.weight"] = ops_s.mm_tn(dz2, p1.reshape(n_rows, Pn).contiguous())
        dp1 = ops_s.gemm_nt(dz2, ops_s.transpose(w1))
        dz1 = ops_s.new(n_rows, Pn)
        _lib.check(lib.gvx_train_relu_dropout_backward(_p(dp1), _p(p1.contiguous()), _p(pk[0].contiguous()), 2.0, n_rows * Pn, _p(dz1), st_s))
        g["decoder.prenet.layers.0.linear_layer.weight"] = ops_s.mm_tn(dz1, ch["frames"].reshape(n_rows, M).contiguous())
        if side is not None:
            join = torch.cuda.Event(); join.record(side)
    dmemory = ops.axpby(dmemory.reshape(B * L, E), 1.0, ops.gemm_nt(dpm2, ops.transpose(wm)), 1.0).reshape(B, L, E)
    # ---- encoder BiLSTM, packed-sequence semantics (models/tts/tacotron2.py:239-245): one C-ABI call, a launch per time step
    H = E // 2
    x = x.contiguous()
    c_enc = ch["enc_cell_states"]                                 # [B, L, E] (forward direction in channels [0, H))
    dx_enc = ops.zeros(B * L, E)
    sfxs = ("", "_reverse")
    w_ih2 = [W("encoder.lstm.weight_ih_l0" + sfx) for sfx in sfxs]
    w_hh2 = torch.stack([W("encoder.lstm.weight_hh_l0" + sfx) for sfx in sfxs]).contiguous()        # [2, 4H, H]
    xg = torch.empty(2, B * L, 4 * H, device=dev)
    for d_, sfx in enumerate(sfxs):
        bsum = ops.axpby(W("encoder.lstm.bias_ih_l0" + sfx)[None, :], 1.0, W("encoder.lstm.bias_hh_l0" + sfx)[None, :], 1.0)[0]
        ops.gemm_nt(x.reshape(B * L, E), w_ih2[d_], bias=bsum, out=xg[d_])
    dg_pos, hprev_pos = torch.empty(2, B * L, 4 * H, device=dev), torch.empty(2, B * L, H, device=dev)
    wse = torch.empty(lib.gvx_train_encoder_lstm_bptt_workspace_bytes(B, H), dtype=torch.uint8, device=dev)
    tl32 = ch["token_lengths"].to(device=dev, dtype=torch.int32).contiguous()
    # (one resident launch for the whole walk unless the model has seen it time out: encoder_bptt_timed_out below)
    walk = lib.gvx_train_encoder_lstm_bptt_resident if getattr(model, "_enc_bptt_resident", True) else lib.gvx_train_encoder_lstm_bptt
    _lib.check(walk(_p(xg), _p(memory.contiguous()), _p(c_enc.contiguous()), _p(dmemory.contiguous()), _p(w_hh2), _p(tl32),
                    B, L, H, _p(dg_pos), _p(hprev_pos), _p(wse), wse.numel(), st))
    if not hasattr(model, "_enc_bptt_workspaces"):
        model._enc_bptt_workspaces = []
    model._enc_bptt_workspaces.append((wse, B, H))   # (kept until the step's gradients have been looked at)
    for d_, sfx in enumerate(sfxs):
        dg2 = dg_pos[d_]
        g["encoder.lstm.weight_ih_l0" + sfx] = ops.mm_tn(dg2, x.reshape(B * L, E))
        g["encoder.lstm.weight_hh_l0" + sfx] = ops.mm_tn(dg2, hprev_pos[d_])
        g["encoder.lstm.bias_ih_l0" + sfx] = ops.colsum(dg2)
        g["encoder.lstm.bias_hh_l0" + sfx] = g["encoder.lstm.bias_ih_l0" + sfx].clone()
        dx_enc = ops.axpby(dx_enc, 1.0, ops.gemm_nt(dg2, ops.transpose(w_ih2[d_])), 1.0)
    if side is not None:
        main_stream.wait_event(join)
    return g, dx_enc


def train_backward(model, batch: Dict[str, torch.Tensor], outputs: Dict[str, torch.Tensor], tape: dict) -> Dict[str, torch.Tensor]:
    """Gradients of loss = Tacotron2Loss(batch, outputs) w.r.t. every parameter (keys = the reference's state_dict names),
    from the tape of ``Tacotron2._forward_train``.  What mixes batch rows - the criterion's means and the BatchNorm batch
    statistics of both convolution stacks - runs on the whole batch; the recurrent part runs per chunk of at most 32 rows
    (``_recurrent_backward``), its parameter gradients added in chunk order."""
    dev = outputs["mel_outputs"].device
    ops = _Ops(dev)
    lib, st = ops.lib, ops.st
    P_ = dict(model.named_parameters())
    tok = batch["token_padded"].to(device=dev, dtype=torch.int64).contiguous()
    B, L = tok.shape
    M, T = outputs["mel_outputs"].shape[1], outputs["mel_outputs"].shape[2]
    E = model.model_config.encoder_embedding_dim
    g: Dict[str, torch.Tensor] = {}
    # ---- criterion and Postnet (whole batch)
    dmel_direct, dpost, dgate = loss_backward(batch, outputs)
    dx, pg = convstack_train_backward(tape["postnet"], dpost, first_layer_wgrad_input=outputs["mel_outputs"])
    g.update(pg)
    dmel = ops.axpby(dmel_direct.reshape(B * M, T), 1.0, dpost.reshape(B * M, T), 1.0)
    dmel = ops.axpby(dmel, 1.0, dx.reshape(B * M, T), 1.0).reshape(B, M, T)
    # ---- the recurrent part, chunk by chunk
    x_all = tape["enc_conv_out"].permute(0, 2, 1).contiguous()    # [B, L, E]: the BiLSTM's input
    dx_enc = ops.new(B * L, E)
    for ch in tape["chunks"]:
        lo, hi = ch["rows"]
        cg, dxe = _recurrent_backward(model, ops, ch, dmel[lo:hi], dgate[lo:hi], x_all[lo:hi], outputs["alignments"][lo:hi])
        _accumulate(ops, g, cg)
        dx_enc[lo * L:hi * L] = dxe
    # ---- encoder convolution stack and embedding (whole batch)
    dconv, eg = convstack_train_backward(tape["encoder"], dx_enc.reshape(B, L, E).permute(0, 2, 1).contiguous())
    g.update(eg)
    demb = torch.empty_like(P_["embedding.weight"].data)
    dtok = dconv.permute(0, 2, 1).contiguous()                    # [B, L, E]
    _lib.check(lib.gvx_train_embedding_backward(_p(tok), _p(dtok), B * L, E, demb.shape[0], _p(demb), st))
    g["embedding.weight"] = demb
    return g


def encoder_bptt_timed_out(model) -> bool:
    """True when a resident walk of the encoder BiLSTM (gvx_train_encoder_lstm_bptt_resident) of this step gave up on a hand-off -
    its workgroups could not run at the same time; its outputs are NaN then.  Reads the status word of every workspace the step
    used (one stream synchronisation each) and forgets them."""
    import ctypes as C

    lib = _lib.load()
    hit = False
    for wse, B, H in getattr(model, "_enc_bptt_workspaces", []):
        code = C.c_int(0)
        _lib.check(lib.gvx_train_encoder_lstm_bptt_status(_p(wse), wse.numel(), B, H, C.byref(code), torch.cuda.current_stream(wse.device).cuda_stream))
        hit = hit or code.value != 0
    model._enc_bptt_workspaces = []
    return hit


def _device_table(entries, dev) -> torch.Tensor:
    """A ctypes array of reference structs as device bytes (what the many-tensor kernels index by blockIdx.y)."""
    raw = bytes(entries)
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)


def clip_grad_norm(grads: Dict[str, torch.Tensor], max_norm: float) -> Tuple[float, float]:
    """torch.nn.utils.clip_grad_norm_ (models/tts/tacotron2.py:521): returns (total norm, scale to apply to every gradient).
    One launch over all gradients + one that adds the partial sums in a fixed order (reproducible bit for bit)."""
    lib = _lib.load()
    gs = [v if v.is_contiguous() else v.contiguous() for v in grads.values()]
    dev = gs[0].device
    st = torch.cuda.current_stream(dev).cuda_stream
    refs = (_lib.gvx_tensor_ref * len(gs))()
    for i, v in enumerate(gs):
        refs[i] = _lib.gvx_tensor_ref(v.data_ptr(), v.numel())
    table = _device_table(refs, dev)
    scratch = torch.empty(lib.gvx_train_sqnorm_scratch_bytes(len(gs)), dtype=torch.uint8, device=dev)
    out = torch.empty(1, dtype=torch.float64, device=dev)
    _lib.check(lib.gvx_train_sqnorm_many(_p(table), len(gs), _p(scratch), _p(out), st))
    total = float(out.sqrt().item())
    coef = max_norm / (total + 1e-6)
    return total, (coef if coef < 1.0 else 1.0)


class Adam:
    """torch.optim.Adam as the reference configures it (models/tts/tacotron2.py:506-513): one HIP launch for all parameters."""

    def __init__(self, model, lr: float, weight_decay: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-8):
        self.model, self.lr, self.wd, self.betas, self.eps = model, lr, weight_decay, betas, eps
        self.step_count = 0
        self.state: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}

    def step(self, grads: Dict[str, torch.Tensor], grad_scale: float = 1.0) -> None:
        lib = _lib.load()
        self.step_count += 1
        params = list(self.model.named_parameters())
        refs = (_lib.gvx_adam_ref * len(params))()
        keep = []
        for i, (name, p) in enumerate(params):
            gk = grads[name].contiguous()
            keep.append(gk)
            if name not in self.state:
                self.state[name] = (torch.zeros_like(p.data), torch.zeros_like(p.data))
            m, v = self.state[name]
            refs[i] = _lib.gvx_adam_ref(p.data.data_ptr(), gk.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel())
        dev = params[0][1].device
        table = _device_table(refs, dev)
        _lib.check(lib.gvx_train_adam_step_many(_p(table), len(params), float(grad_scale), float(self.lr), float(self.wd), float(self.betas[0]),
                                                float(self.betas[1]), float(self.eps), self.step_count, torch.cuda.current_stream(dev).cuda_stream))
        # the parameters changed through raw pointers: the packed blob of the forward kernels is stale
        self.model._packed_key = None
