"""Training-mode pieces of the MI355X path - the first slice of SURVEY.md section 8f rank 4 (the reference's
``Tacotron2.train_step``, models/tts/tacotron2.py:515-522).

What exists: the convolution stacks of the encoder and of the Postnet exactly as the reference runs them under ``.train()``
(``nn.Conv1d`` + ``nn.BatchNorm1d`` with batch statistics and its running-statistics update + activation + ``F.dropout``;
models/tts/tacotron2.py:149-199, :207-220, :234-235), forward and backward, and the backward of ``Tacotron2Loss``
(:598-615) - HIP kernels behind the C ABI (``gvx_conv_bn_act_train_forward`` / ``_backward``, ``gvx_tacotron2_loss_backward``;
csrc/train.hip), working on the model's parameters in place, in the reference's own layouts.  Pinned by
``tests/golden/train_small.npz``: the reference's own ``loss.backward()`` gradients.

What does not exist yet: the backward through the LSTMs and the attention (BPTT over the decoder loop and the encoder
BiLSTM) and the optimiser step; ``Tacotron2.train_step`` therefore still raises.  Dropout keep masks are explicit inputs
(uint8, the shape of the dropout's input), as for the Prenet: parity is defined for given masks.
"""
from __future__ import annotations

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
