"""ORACLE — test infrastructure, never shipped on the product path.

CPU restatement (plain torch fp32 tensor math, explicit formulas, no autograd) of the TRAINING-mode pieces of the
reference's Tacotron2 that the build covers so far (first slice): the convolution stacks
(conv1d + BatchNorm1d with batch statistics + activation + dropout) of the encoder (models/tts/tacotron2.py:207-220,
:234-235) and of the Postnet (:149-199), their backward, and the backward of the criterion (Tacotron2Loss, :598-615).

Parity pinning: tests/golden/train_small.npz is generated from the reference itself in .train() mode (forward, loss,
loss.backward(); tests/golden/make_fixtures.py train); tests/test_oracle_golden.py holds this file to it.
Dropout is part of the function: keep masks are explicit inputs (uint8 {0,1}, the shape of the dropout's input).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def conv_bn_act_drop_forward(x, w, b, gamma, beta, act: str, keep, p: float = 0.5):
    """One layer, training mode.  x [B, Cin, T] -> y [B, Cout, T].  Returns y and what its backward needs.
    BatchNorm1d in training mode normalises with the BIASED batch variance over (B, T) (torch.nn.BatchNorm1d)."""
    k = w.shape[2]
    z = F.conv1d(x, w, b, padding=(k - 1) // 2)                      # [B, Cout, T]
    mean = z.mean(dim=(0, 2))
    var = z.var(dim=(0, 2), unbiased=False)
    invstd = torch.rsqrt(var + BN_EPS)
    xhat = (z - mean[None, :, None]) * invstd[None, :, None]
    u = xhat * gamma[None, :, None] + beta[None, :, None]
    a = torch.tanh(u) if act == "tanh" else (torch.relu(u) if act == "relu" else u)
    y = a * keep.to(a.dtype) / (1.0 - p)
    return y, dict(x=x, xhat=xhat, invstd=invstd, a=a, mean=mean, var=var)


def conv_bn_act_drop_backward(dy, saved, w, gamma, act: str, keep, p: float = 0.5):
    """Backward of one layer: returns dx and the gradients of (conv weight, conv bias, BN weight, BN bias)."""
    a, xhat, invstd, x = saved["a"], saved["xhat"], saved["invstd"], saved["x"]
    da = dy * keep.to(dy.dtype) / (1.0 - p)
    du = da * (1.0 - a * a) if act == "tanh" else (da * (a > 0).to(da.dtype) if act == "relu" else da)
    dbeta = du.sum(dim=(0, 2))
    dgamma = (du * xhat).sum(dim=(0, 2))
    n = du.shape[0] * du.shape[2]
    dz = (gamma * invstd)[None, :, None] * (du - dbeta[None, :, None] / n - xhat * dgamma[None, :, None] / n)
    k = w.shape[2]
    pad = (k - 1) // 2
    db = dz.sum(dim=(0, 2))
    xp = F.pad(x, (pad, pad))                                         # [B, Cin, T + 2 pad]
    # dW[co][ci][j] = sum_{b,t} dz[b][co][t] * xp[b][ci][t + j]
    dw = torch.stack([torch.einsum("bot,bit->oi", dz, xp[:, :, j:j + dz.shape[2]]) for j in range(k)], dim=2)
    # dx[b][ci][t] = sum_{co,j} dz[b][co][t + pad - j] * w[co][ci][j]
    dzp = F.pad(dz, (pad, pad))
    dx = sum(torch.einsum("bot,oi->bit", dzp[:, :, 2 * pad - j:2 * pad - j + dz.shape[2]], w[:, :, j]) for j in range(k))
    return dx, dw, db, dgamma, dbeta


def _stack_params(sd, prefix: str, n: int):
    return [(sd[f"{prefix}.{i}.0.conv.weight"], sd[f"{prefix}.{i}.0.conv.bias"], sd[f"{prefix}.{i}.1.weight"], sd[f"{prefix}.{i}.1.bias"])
            for i in range(n)]


def convstack_train_forward(sd, prefix: str, n: int, acts: List[str], x, keeps):
    saved = []
    for i, (w, b, g, be) in enumerate(_stack_params(sd, prefix, n)):
        x, s = conv_bn_act_drop_forward(x, w, b, g, be, acts[i], keeps[i])
        saved.append(s)
    return x, saved


def convstack_train_backward(sd, prefix: str, n: int, acts: List[str], dy, saved, keeps) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    grads = {}
    for i in reversed(range(n)):
        w, b, g, be = _stack_params(sd, prefix, n)[i]
        dy, dw, db, dg, dbe = conv_bn_act_drop_backward(dy, saved[i], w, g, acts[i], keeps[i])
        grads[f"{prefix}.{i}.0.conv.weight"], grads[f"{prefix}.{i}.0.conv.bias"] = dw, db
        grads[f"{prefix}.{i}.1.weight"], grads[f"{prefix}.{i}.1.bias"] = dg, dbe
    return dy, grads


def encoder_acts(n: int) -> List[str]:
    return ["relu"] * n                                               # models/tts/tacotron2.py:234-235


def postnet_acts(n: int) -> List[str]:
    return ["tanh"] * (n - 1) + ["none"]                              # models/tts/tacotron2.py:194-199


def bn_running_update(running_mean, running_var, mean, var, n: int, momentum: float = 0.1):
    """Side effect of a training-mode BatchNorm1d forward: running statistics (unbiased variance), torch semantics."""
    return (1 - momentum) * running_mean + momentum * mean, (1 - momentum) * running_var + momentum * var * n / (n - 1)


def loss_backward(batch, outputs):
    """d loss / d (mel_outputs [direct term], mel_outputs_postnet, gate_outputs) of Tacotron2Loss (models/tts/tacotron2.py:598-615):
    two mean-squared errors over all B * n_mels * T elements and BCE-with-logits over all B * T elements.  (The reference
    masks the outputs in place outside autograd, :466-473: the padded positions enter with their masked values.)"""
    mel_t, gate_t = batch["mel_padded"], batch["gate_padded"]
    mel, post, gate = outputs["mel_outputs"], outputs["mel_outputs_postnet"], outputs["gate_outputs"]
    n = mel.numel()
    return 2.0 * (mel - mel_t) / n, 2.0 * (post - mel_t) / n, (torch.sigmoid(gate) - gate_t) / gate.numel()
