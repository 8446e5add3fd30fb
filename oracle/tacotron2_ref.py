"""ORACLE — test infrastructure, never shipped on the product path.

CPU restatement (plain torch fp32 tensor math, no nn.Module, no autograd) of the
reference's Tacotron2 text->mel forward.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this file; ``genvox_amd`` never does.

Parity pinning: the reference has no tests or golden vectors of its own
(SURVEY.md section 4), so this restatement is pinned by fixtures generated from the
reference itself, imported unchanged in the build container by
``tests/golden/make_fixtures.py`` (committed with its outputs under ``tests/golden/``);
``tests/test_oracle_golden.py`` checks this file against them.

Each function cites the reference lines it restates (paths relative to the reference root).
Weights come in as a dict with the reference's ``state_dict`` key names.
Prenet dropout is part of the function (always on in the reference), so the two
Bernoulli keep-masks are explicit inputs (uint8 {0,1}).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # torch.nn.BatchNorm1d default, used by models/tts/tacotron2.py:161,:176,:190,:218


def _bn_eval(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    """BatchNorm1d in eval mode on [B, C, T] (running statistics)."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    mu, var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    scale = w / torch.sqrt(var + BN_EPS)
    return (x - mu[None, :, None]) * scale[None, :, None] + b[None, :, None]


def _conv_bn(x: torch.Tensor, sd, prefix: str) -> torch.Tensor:
    """ConvNorm + BatchNorm1d pair (models/generic.py:23-54; tacotron2.py:149-192, :207-220)."""
    w = sd[prefix + ".0.conv.weight"]
    y = F.conv1d(x, w, sd[prefix + ".0.conv.bias"], padding=(w.shape[2] - 1) // 2)
    return _bn_eval(y, sd, prefix + ".1")


def _lstm_cell(gates: torch.Tensor, c: torch.Tensor):
    """torch LSTM gate order i, f, g, o (used by nn.LSTM :222 and nn.LSTMCell :286/:294)."""
    i, f, g, o = gates.chunk(4, dim=-1)
    c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    return torch.sigmoid(o) * torch.tanh(c_new), c_new


def encoder(sd, tokens: torch.Tensor, lengths: Optional[torch.Tensor]) -> torch.Tensor:
    """Embedding + 3x(conv,BN,relu) + BiLSTM with packed-sequence semantics.

    tacotron2.py:459/:486 (embedding + transpose), :231-246 (Encoder.forward, packed) and
    :248-256 (Encoder.inference: no lengths -> every row runs the full L).
    The convs run over padded positions unmasked; the reverse direction starts at each
    row's own last token; outputs past a row's length are zero.
    """
    x = sd["embedding.weight"][tokens.long()].transpose(1, 2)  # [B, E, L]
    n_conv = 0
    while f"encoder.convolutions.{n_conv}.0.conv.weight" in sd:
        n_conv += 1
    for i in range(n_conv):
        x = torch.relu(_conv_bn(x, sd, f"encoder.convolutions.{i}"))
    x = x.transpose(1, 2).contiguous()  # [B, L, E]
    B, L, E = x.shape
    H = sd["encoder.lstm.weight_hh_l0"].shape[1]
    lens = torch.full((B,), L, dtype=torch.long) if lengths is None else lengths.long().cpu()
    out = torch.zeros(B, L, 2 * H, dtype=x.dtype)
    rows = torch.arange(B)
    for d, sfx in enumerate(("", "_reverse")):
        w_ih, w_hh = sd["encoder.lstm.weight_ih_l0" + sfx], sd["encoder.lstm.weight_hh_l0" + sfx]
        bias = sd["encoder.lstm.bias_ih_l0" + sfx] + sd["encoder.lstm.bias_hh_l0" + sfx]
        xg = x @ w_ih.t() + bias  # [B, L, 4H]
        h = torch.zeros(B, H, dtype=x.dtype)
        c = torch.zeros(B, H, dtype=x.dtype)
        for s in range(L):
            active = s < lens  # [B]
            t_idx = torch.full((B,), s, dtype=torch.long) if d == 0 else (lens - 1 - s).clamp(min=0)
            h_new, c_new = _lstm_cell(xg[rows, t_idx] + h @ w_hh.t(), c)
            h = torch.where(active[:, None], h_new, h)
            c = torch.where(active[:, None], c_new, c)
            sel = rows[active]
            out[sel, t_idx[active], d * H:(d + 1) * H] = h_new[active]
    return out


def prenet(sd, x: torch.Tensor, keep0: torch.Tensor, keep1: torch.Tensor) -> torch.Tensor:
    """tacotron2.py:140-144: relu(linear) then dropout(p=0.5, always on) twice.
    keep masks are {0,1}; kept units are scaled by 1/(1-p) = 2."""
    w0 = sd["decoder.prenet.layers.0.linear_layer.weight"]
    w1 = sd["decoder.prenet.layers.1.linear_layer.weight"]
    h = torch.relu(x @ w0.t()) * (keep0.to(x.dtype) * 2.0)
    return torch.relu(h @ w1.t()) * (keep1.to(x.dtype) * 2.0)


class _DecoderState:
    """State created by Decoder.initialize_decoder_states (tacotron2.py:303-315)."""

    def __init__(self, sd, memory: torch.Tensor, mask: Optional[torch.Tensor]):
        B, L, _ = memory.shape
        A = sd["decoder.attention_rnn.weight_hh"].shape[1]
        D = sd["decoder.decoder_rnn.weight_hh"].shape[1]
        z = lambda n: torch.zeros(B, n, dtype=memory.dtype)
        self.h_a, self.c_a, self.h_d, self.c_d = z(A), z(A), z(D), z(D)
        self.w, self.w_cum, self.ctx = z(L), z(L), z(memory.shape[2])
        self.memory = memory
        self.pm = memory @ sd["decoder.attention_layer.memory_layer.linear_layer.weight"].t()
        self.mask = mask  # True = padded position


def _decode_step(sd, st: _DecoderState, prenet_out: torch.Tensor):
    """Decoder.decode (tacotron2.py:333-363) with Attention.forward (:106-129),
    get_alignment_energies (:89-104) and LocationLayer.forward (:48-53). Eval mode:
    the hidden-state dropouts at :341/:358 are identities."""
    att = "decoder.attention_layer."
    x = torch.cat((prenet_out, st.ctx), dim=-1)
    gates = (x @ sd["decoder.attention_rnn.weight_ih"].t() + sd["decoder.attention_rnn.bias_ih"]
             + st.h_a @ sd["decoder.attention_rnn.weight_hh"].t() + sd["decoder.attention_rnn.bias_hh"])
    st.h_a, st.c_a = _lstm_cell(gates, st.c_a)

    w_cat = torch.stack((st.w, st.w_cum), dim=1)  # [B, 2, L]
    lw = sd[att + "location_layer.location_conv.conv.weight"]
    loc = F.conv1d(w_cat, lw, None, padding=(lw.shape[2] - 1) // 2).transpose(1, 2)  # [B, L, F]
    loc = loc @ sd[att + "location_layer.location_dense.linear_layer.weight"].t()  # [B, L, a]
    q = st.h_a @ sd[att + "query_layer.linear_layer.weight"].t()  # [B, a]
    e = torch.tanh(q[:, None, :] + loc + st.pm) @ sd[att + "v.linear_layer.weight"].t()  # [B, L, 1]
    e = e.squeeze(-1)
    if st.mask is not None:
        e = e.masked_fill(st.mask, float("-inf"))
    st.w = torch.softmax(e, dim=1)
    st.ctx = torch.bmm(st.w[:, None, :], st.memory).squeeze(1)
    st.w_cum = st.w_cum + st.w

    x = torch.cat((st.h_a, st.ctx), dim=-1)
    gates = (x @ sd["decoder.decoder_rnn.weight_ih"].t() + sd["decoder.decoder_rnn.bias_ih"]
             + st.h_d @ sd["decoder.decoder_rnn.weight_hh"].t() + sd["decoder.decoder_rnn.bias_hh"])
    st.h_d, st.c_d = _lstm_cell(gates, st.c_d)

    hc = torch.cat((st.h_d, st.ctx), dim=-1)
    mel = hc @ sd["decoder.linear_projection.linear_layer.weight"].t() + sd["decoder.linear_projection.linear_layer.bias"]
    gate = hc @ sd["decoder.gate_layer.linear_layer.weight"].t() + sd["decoder.gate_layer.linear_layer.bias"]
    return mel, gate.squeeze(1), st.w


def postnet(sd, mel: torch.Tensor) -> torch.Tensor:
    """Postnet.forward, eval mode (tacotron2.py:194-200): tanh on all but the last layer."""
    n = 0
    while f"postnet.convolutions.{n}.0.conv.weight" in sd:
        n += 1
    x = mel
    for i in range(n):
        x = _conv_bn(x, sd, f"postnet.convolutions.{i}")
        if i < n - 1:
            x = torch.tanh(x)
    return x


def lengths_to_pad_mask(lengths: torch.Tensor, max_len: Optional[int] = None) -> torch.Tensor:
    """get_mask_from_lengths (tacotron2.py:17-21): True where index >= length."""
    n = int(lengths.max()) if max_len is None else max_len
    return torch.arange(n)[None, :] >= lengths.long()[:, None]


@torch.no_grad()
def tacotron2_forward(sd, batch: Dict[str, torch.Tensor], keep_masks: torch.Tensor,
                      mask_padding: bool = True, taps: bool = False) -> Dict[str, torch.Tensor]:
    """Teacher-forced Tacotron2.forward (tacotron2.py:450-481) with Decoder.forward (:365-388).

    keep_masks: uint8 [2, (T+1)*B, P], row index = t*B + b (the reference applies one
    dropout draw over the whole [T+1, B, P] prenet activation, :373)."""
    tokens, tok_len = batch["token_padded"], batch["token_lengths"]
    mel_in, mel_len = batch["mel_padded"].float(), batch["mel_lengths"]
    B, M, T = mel_in.shape
    memory = encoder(sd, tokens, tok_len)
    frames = torch.cat((torch.zeros(1, B, M), mel_in.permute(2, 0, 1)), dim=0)  # [T+1, B, M]
    P = sd["decoder.prenet.layers.0.linear_layer.weight"].shape[0]
    km = keep_masks.reshape(2, T + 1, B, P)
    pre = prenet(sd, frames, km[0], km[1])  # [T+1, B, P]
    st = _DecoderState(sd, memory, lengths_to_pad_mask(tok_len, tokens.shape[1]))
    mels, gates, aligns, ctxs = [], [], [], []
    for t in range(T):
        m, g, w = _decode_step(sd, st, pre[t])
        mels.append(m); gates.append(g); aligns.append(w)
        if taps:
            ctxs.append(st.ctx)
    mel_out = torch.stack(mels, dim=2)  # [B, M, T]
    gate_out = torch.stack(gates, dim=1)  # [B, T]
    align = torch.stack(aligns, dim=1)  # [B, T, L]
    mel_post = mel_out + postnet(sd, mel_out)
    if mask_padding:  # tacotron2.py:466-473
        pad = lengths_to_pad_mask(mel_len, T)
        mel_out = mel_out.masked_fill(pad[:, None, :], 0.0)
        mel_post = mel_post.masked_fill(pad[:, None, :], 0.0)
        gate_out = gate_out.masked_fill(pad, 1e3)
    out = {"mel_outputs": mel_out, "mel_outputs_postnet": mel_post, "gate_outputs": gate_out, "alignments": align}
    if taps:
        out.update({"encoder_outputs": memory, "processed_memory": st.pm, "prenet_outputs": pre,
                    "attention_contexts": torch.stack(ctxs, dim=1)})
    return out


@torch.no_grad()
def tacotron2_inference(sd, tokens: torch.Tensor, keep_masks: torch.Tensor, gate_threshold: float,
                        max_decoder_steps: int, token_length: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """Autoregressive Tacotron2.inference (tacotron2.py:483-499) with Decoder.inference (:390-414).

    Batch 1 only, like the reference (its stop test at :405 is a scalar truth test).
    keep_masks: uint8 [2, max_decoder_steps, P]; step t uses [0, t] then [1, t].
    token_length (not in the reference's inference, which never pads): the row is one row of a PADDED batch;
    it is then treated exactly as Tacotron2.forward treats padded rows (:459-462): convs run over the pad-token
    embeddings unmasked, packed BiLSTM, -inf attention mask past the length.  This is the semantics the batched
    autoregressive extension is checked against."""
    assert tokens.shape[0] == 1
    if token_length is None:
        memory = encoder(sd, tokens, None)
        st = _DecoderState(sd, memory, None)
    else:
        lens = torch.tensor([token_length])
        memory = encoder(sd, tokens, lens)
        st = _DecoderState(sd, memory, lengths_to_pad_mask(lens, tokens.shape[1]))
    M = sd["decoder.linear_projection.linear_layer.weight"].shape[0]
    frame = torch.zeros(1, M)
    mels, gates, aligns = [], [], []
    while True:
        t = len(mels)
        m, g, w = _decode_step(sd, st, prenet(sd, frame, keep_masks[0, t][None], keep_masks[1, t][None]))
        mels.append(m); gates.append(g); aligns.append(w)
        if torch.sigmoid(g).item() > gate_threshold or len(mels) >= max_decoder_steps:
            break
        frame = m
    mel_out = torch.stack(mels, dim=2)
    mel_post = mel_out + postnet(sd, mel_out)
    return {"mel_outputs": mel_out, "mel_outputs_postnet": mel_post,
            "gate_outputs": torch.stack(gates, dim=1), "alignments": torch.stack(aligns, dim=1)}


def tacotron2_loss(batch, outputs):
    """Restates Tacotron2Loss (models/tts/tacotron2.py:598-615): MSE(mel_outputs, target) + MSE(mel_outputs_postnet,
    target) (means over all B*n_mels*T elements, padding included) + BCE-with-logits(gate_outputs, gate target) (mean
    over B*T).  float32 tensors in, dict of 0-dim float32 tensors out."""
    mel_t = batch["mel_padded"].to(torch.float32)
    gate_t = batch["gate_padded"].to(torch.float32).reshape(-1)
    mel_loss = ((outputs["mel_outputs"] - mel_t) ** 2).mean() + ((outputs["mel_outputs_postnet"] - mel_t) ** 2).mean()
    x = outputs["gate_outputs"].reshape(-1)
    gate_loss = (torch.clamp(x, min=0) - x * gate_t + torch.log1p(torch.exp(-x.abs()))).mean()
    return {"loss": mel_loss + gate_loss, "mel_loss": mel_loss, "gate_loss": gate_loss}
