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


# =====================================================================================================================
# The whole training step in explicit formulas (no autograd): forward with everything the backward needs recorded, then
# back-propagation through time over the decoder loop (attention LSTM, location-sensitive attention, decoder LSTM,
# projections), the Prenet, the encoder BiLSTM, its convolution stack and the embedding - what the reference's
# loss.backward() computes (models/tts/tacotron2.py:515-520).  Gradient clipping and Adam follow at the end of the file.
# =====================================================================================================================
def _sig(x):
    return torch.sigmoid(x)


def _lstm_cell_fwd(gates, c_prev):
    i, f, g, o = gates.chunk(4, dim=-1)
    i, f, g, o = _sig(i), _sig(f), torch.tanh(g), _sig(o)
    c = f * c_prev + i * g
    tc = torch.tanh(c)
    return o * tc, c, (i, f, g, o, tc, c_prev)


def _lstm_cell_bwd(dh, dc_next, saved):
    """Gradients of the four gate pre-activations (torch order i, f, g, o) and of the previous cell state."""
    i, f, g, o, tc, c_prev = saved
    do = dh * tc
    dc = dh * o * (1 - tc * tc) + dc_next
    dgates = torch.cat((dc * g * i * (1 - i), dc * c_prev * f * (1 - f), dc * i * (1 - g * g), do * o * (1 - o)), dim=-1)
    return dgates, dc * f


def train_forward(sd, batch, masks, mc):
    """Training-mode forward.  masks: dict with encoder [n][B,E,L], prenet [2][T+1,B,P], attention_rnn [T][B,A],
    decoder_rnn [T][B,D], postnet list.  mc: the model config (dropout probabilities, mask_padding).  Returns outputs + tape."""
    from oracle import tacotron2_ref as R

    tok, tl = batch["token_padded"].long(), batch["token_lengths"].long()
    mel_in, ml = batch["mel_padded"], batch["mel_lengths"].long()
    B, L = tok.shape
    _, M, T = mel_in.shape
    ne, npn = mc.encoder_n_convolutions, mc.postnet_n_convolutions
    tape = {}
    emb = sd["embedding.weight"][tok].transpose(1, 2)                                     # [B, E, L]
    xc, tape["enc_convs"] = convstack_train_forward(sd, "encoder.convolutions", ne, encoder_acts(ne), emb, masks["encoder"])
    x = xc.transpose(1, 2).contiguous()                                                   # [B, L, E]
    # BiLSTM, packed semantics (models/tts/tacotron2.py:239-245)
    H = sd["encoder.lstm.weight_hh_l0"].shape[1]
    memory = torch.zeros(B, L, 2 * H)
    rows = torch.arange(B)
    enc = []
    for d, sfx in enumerate(("", "_reverse")):
        w_ih, w_hh = sd["encoder.lstm.weight_ih_l0" + sfx], sd["encoder.lstm.weight_hh_l0" + sfx]
        bias = sd["encoder.lstm.bias_ih_l0" + sfx] + sd["encoder.lstm.bias_hh_l0" + sfx]
        xg = x @ w_ih.t() + bias
        h, c = torch.zeros(B, H), torch.zeros(B, H)
        steps = []
        for s in range(L):
            active = s < tl
            t_idx = torch.full((B,), s, dtype=torch.long) if d == 0 else (tl - 1 - s).clamp(min=0)
            h_new, c_new, sv = _lstm_cell_fwd(xg[rows, t_idx] + h @ w_hh.t(), c)
            steps.append((active, t_idx, sv, h))
            h = torch.where(active[:, None], h_new, h)
            c = torch.where(active[:, None], c_new, c)
            memory[rows[active], t_idx[active], d * H:(d + 1) * H] = h_new[active]
        enc.append(steps)
    tape.update(enc_lstm=enc, x=x, memory=memory)
    # decoder
    att = "decoder.attention_layer."
    A, D = sd["decoder.attention_rnn.weight_hh"].shape[1], sd["decoder.decoder_rnn.weight_hh"].shape[1]
    E = memory.shape[2]
    frames = torch.cat((torch.zeros(1, B, M), mel_in.permute(2, 0, 1)), 0)                # [T+1, B, M]
    w0, w1 = sd["decoder.prenet.layers.0.linear_layer.weight"], sd["decoder.prenet.layers.1.linear_layer.weight"]
    p1 = torch.relu(frames @ w0.t()) * (masks["prenet"][0].float() * 2)
    p2 = torch.relu(p1 @ w1.t()) * (masks["prenet"][1].float() * 2)                       # [T+1, B, P]
    tape.update(frames=frames, p1=p1, p2=p2)
    pm = memory @ sd[att + "memory_layer.linear_layer.weight"].t()
    pad_mask = torch.arange(L)[None, :] >= tl[:, None]
    h_a, c_a, h_d, c_d = torch.zeros(B, A), torch.zeros(B, A), torch.zeros(B, D), torch.zeros(B, D)
    w, wcum, ctx = torch.zeros(B, L), torch.zeros(B, L), torch.zeros(B, E)
    lw, ld = sd[att + "location_layer.location_conv.conv.weight"], sd[att + "location_layer.location_dense.linear_layer.weight"]
    wq, v = sd[att + "query_layer.linear_layer.weight"], sd[att + "v.linear_layer.weight"]
    sa, sdp = 1.0 / (1.0 - mc.p_attention_dropout), 1.0 / (1.0 - mc.p_decoder_dropout)
    steps, mels, gates_out, aligns = [], [], [], []
    for t in range(T):
        xa = torch.cat((p2[t], ctx), -1)
        ga = (xa @ sd["decoder.attention_rnn.weight_ih"].t() + sd["decoder.attention_rnn.bias_ih"]
              + h_a @ sd["decoder.attention_rnn.weight_hh"].t() + sd["decoder.attention_rnn.bias_hh"])
        h_a_prev = h_a
        ht, c_a, sva = _lstm_cell_fwd(ga, c_a)
        ka = masks["attention_rnn"][t].float() * sa
        h_a = ht * ka
        loc_in = torch.stack((w, wcum), 1)                                                # [B, 2, L]
        locf = F.conv1d(loc_in, lw, None, padding=(lw.shape[2] - 1) // 2)                 # [B, F, L]
        th = torch.tanh((h_a @ wq.t())[:, None, :] + locf.transpose(1, 2) @ ld.t() + pm)  # [B, L, a]
        e = (th @ v.t()).squeeze(-1).masked_fill(pad_mask, float("-inf"))
        w = torch.softmax(e, 1)
        ctx = torch.bmm(w[:, None, :], memory).squeeze(1)
        wcum = wcum + w
        xd = torch.cat((h_a, ctx), -1)
        gd = (xd @ sd["decoder.decoder_rnn.weight_ih"].t() + sd["decoder.decoder_rnn.bias_ih"]
              + h_d @ sd["decoder.decoder_rnn.weight_hh"].t() + sd["decoder.decoder_rnn.bias_hh"])
        h_d_prev = h_d
        ht, c_d, svd = _lstm_cell_fwd(gd, c_d)
        kd = masks["decoder_rnn"][t].float() * sdp
        h_d = ht * kd
        hc = torch.cat((h_d, ctx), -1)
        mels.append(hc @ sd["decoder.linear_projection.linear_layer.weight"].t() + sd["decoder.linear_projection.linear_layer.bias"])
        gates_out.append((hc @ sd["decoder.gate_layer.linear_layer.weight"].t() + sd["decoder.gate_layer.linear_layer.bias"]).squeeze(1))
        aligns.append(w)
        steps.append(dict(xa=xa, h_a_prev=h_a_prev, sva=sva, ka=ka, h_a=h_a, loc_in=loc_in, locf=locf, th=th, w=w, xd=xd,
                          h_d_prev=h_d_prev, svd=svd, kd=kd, hc=hc))
    dec_mel = torch.stack(mels, 2)                                                        # [B, M, T]
    gate = torch.stack(gates_out, 1)
    tape.update(steps=steps, pm=pm, pad_mask=pad_mask)
    y, tape["post_convs"] = convstack_train_forward(sd, "postnet.convolutions", npn, postnet_acts(npn), dec_mel, masks["postnet"])
    mel_out, post = dec_mel.clone(), dec_mel + y
    if mc.mask_padding:
        pm_t = torch.arange(T)[None, :] >= ml[:, None]
        mel_out = mel_out.masked_fill(pm_t[:, None, :], 0.0)
        post = post.masked_fill(pm_t[:, None, :], 0.0)
        gate = gate.masked_fill(pm_t, 1e3)
    outputs = {"mel_outputs": mel_out, "mel_outputs_postnet": post, "gate_outputs": gate, "alignments": torch.stack(aligns, 1)}
    return outputs, tape


def train_backward(sd, batch, masks, mc, outputs, tape) -> Dict[str, torch.Tensor]:
    """Gradients of every parameter (keys = state_dict names) of loss = Tacotron2Loss(batch, outputs)."""
    att = "decoder.attention_layer."
    tok, tl = batch["token_padded"].long(), batch["token_lengths"].long()
    B, L = tok.shape
    T = batch["mel_padded"].shape[2]
    ne, npn = mc.encoder_n_convolutions, mc.postnet_n_convolutions
    g: Dict[str, torch.Tensor] = {}
    dmel_direct, dpost, dgate = loss_backward(batch, outputs)
    tape["post_convs"][0]["x"] = outputs["mel_outputs"]       # the reference's in-place masking quirk (see convstack test)
    dx, pg = convstack_train_backward(sd, "postnet.convolutions", npn, postnet_acts(npn), dpost, tape["post_convs"], masks["postnet"])
    g.update(pg)
    dmel = dmel_direct + dpost + dx                                                      # [B, M, T]
    Wp, Wg = sd["decoder.linear_projection.linear_layer.weight"], sd["decoder.gate_layer.linear_layer.weight"]
    Wia, Wha = sd["decoder.attention_rnn.weight_ih"], sd["decoder.attention_rnn.weight_hh"]
    Wid, Whd = sd["decoder.decoder_rnn.weight_ih"], sd["decoder.decoder_rnn.weight_hh"]
    lw, ld = sd[att + "location_layer.location_conv.conv.weight"], sd[att + "location_layer.location_dense.linear_layer.weight"]
    wq, v = sd[att + "query_layer.linear_layer.weight"], sd[att + "v.linear_layer.weight"]
    wm = sd[att + "memory_layer.linear_layer.weight"]
    memory = tape["memory"]
    A, D, E, P = Wha.shape[1], Whd.shape[1], memory.shape[2], Wia.shape[1] - memory.shape[2]
    pad = (lw.shape[2] - 1) // 2
    z = torch.zeros_like
    for k in ("decoder.linear_projection.linear_layer.weight", "decoder.linear_projection.linear_layer.bias", "decoder.gate_layer.linear_layer.weight",
              "decoder.gate_layer.linear_layer.bias", "decoder.attention_rnn.weight_ih", "decoder.attention_rnn.weight_hh", "decoder.attention_rnn.bias_ih",
              "decoder.decoder_rnn.weight_ih", "decoder.decoder_rnn.weight_hh", "decoder.decoder_rnn.bias_ih", att + "location_layer.location_conv.conv.weight",
              att + "location_layer.location_dense.linear_layer.weight", att + "query_layer.linear_layer.weight", att + "v.linear_layer.weight"):
        g[k] = z(sd[k])
    dmemory, dpm = z(memory), z(tape["pm"])
    dh_a_n, dc_a_n, dh_d_n, dc_d_n = torch.zeros(B, A), torch.zeros(B, A), torch.zeros(B, D), torch.zeros(B, D)
    dctx_n, dw_n, G = torch.zeros(B, E), torch.zeros(B, L), torch.zeros(B, L)          # G: gradient w.r.t. the cumulative weights
    dp2 = torch.zeros(T + 1, B, P)
    for t in reversed(range(T)):
        s = tape["steps"][t]
        dm, dg_ = dmel[:, :, t], dgate[:, t:t + 1]
        g["decoder.linear_projection.linear_layer.weight"] += dm.t() @ s["hc"]
        g["decoder.linear_projection.linear_layer.bias"] += dm.sum(0)
        g["decoder.gate_layer.linear_layer.weight"] += dg_.t() @ s["hc"]
        g["decoder.gate_layer.linear_layer.bias"] += dg_.sum(0)
        dhc = dm @ Wp + dg_ @ Wg
        dh_d, dctx = dhc[:, :D] + dh_d_n, dhc[:, D:] + dctx_n
        dgd, dc_d_n = _lstm_cell_bwd(dh_d * s["kd"], dc_d_n, s["svd"])
        g["decoder.decoder_rnn.weight_ih"] += dgd.t() @ s["xd"]
        g["decoder.decoder_rnn.weight_hh"] += dgd.t() @ s["h_d_prev"]
        g["decoder.decoder_rnn.bias_ih"] += dgd.sum(0)
        dxd = dgd @ Wid
        dh_d_n = dgd @ Whd
        dh_a = dxd[:, :A] + dh_a_n
        dctx = dctx + dxd[:, A:]
        # attention: ctx = w . memory, w = softmax(e), e = v . tanh(q + dense(conv([w_prev ; w_cum_prev])) + pm)
        G = G                                                                            # grad w.r.t. wcum_t (uses at steps > t)
        dw = dw_n + G + torch.einsum("be,ble->bl", dctx, memory)
        dmemory += s["w"][:, :, None] * dctx[:, None, :]
        de = s["w"] * (dw - (s["w"] * dw).sum(1, keepdim=True))
        th = s["th"]
        g[att + "v.linear_layer.weight"] += torch.einsum("bl,bla->a", de, th)[None, :]
        du = de[:, :, None] * v[0][None, None, :] * (1 - th * th)                        # [B, L, a]
        dq = du.sum(1)
        g[att + "query_layer.linear_layer.weight"] += dq.t() @ s["h_a"]
        dh_a = dh_a + dq @ wq
        dpm += du
        g[att + "location_layer.location_dense.linear_layer.weight"] += torch.einsum("bla,bfl->af", du, s["locf"])
        dlocf = torch.einsum("bla,af->bfl", du, ld)                                      # [B, F, L]
        lin_p = F.pad(s["loc_in"], (pad, pad))
        g[att + "location_layer.location_conv.conv.weight"] += torch.stack(
            [torch.einsum("bfl,bcl->fc", dlocf, lin_p[:, :, j:j + L]) for j in range(lw.shape[2])], 2)
        dlf_p = F.pad(dlocf, (pad, pad))
        dloc_in = sum(torch.einsum("bfl,fc->bcl", dlf_p[:, :, 2 * pad - j:2 * pad - j + L], lw[:, :, j]) for j in range(lw.shape[2]))
        dw_n = dloc_in[:, 0]                                                             # w_{t-1} enters as channel 0 ...
        G = G + dloc_in[:, 1]                                                            # ... and wcum_{t-1} as channel 1 (and through wcum_t)
        # attention LSTM
        dga, dc_a_n = _lstm_cell_bwd(dh_a * s["ka"], dc_a_n, s["sva"])
        g["decoder.attention_rnn.weight_ih"] += dga.t() @ s["xa"]
        g["decoder.attention_rnn.weight_hh"] += dga.t() @ s["h_a_prev"]
        g["decoder.attention_rnn.bias_ih"] += dga.sum(0)
        dxa = dga @ Wia
        dh_a_n = dga @ Wha
        dp2[t] = dxa[:, :P]
        dctx_n = dxa[:, P:]
    g["decoder.attention_rnn.bias_hh"] = g["decoder.attention_rnn.bias_ih"].clone()
    g["decoder.decoder_rnn.bias_hh"] = g["decoder.decoder_rnn.bias_ih"].clone()
    # memory projection and Prenet
    g[att + "memory_layer.linear_layer.weight"] = dpm.reshape(-1, dpm.shape[2]).t() @ memory.reshape(-1, E)
    dmemory += dpm @ wm
    w0, w1 = sd["decoder.prenet.layers.0.linear_layer.weight"], sd["decoder.prenet.layers.1.linear_layer.weight"]
    dz2 = dp2 * (masks["prenet"][1].float() * 2) * (tape["p2"] > 0).float()
    g["decoder.prenet.layers.1.linear_layer.weight"] = dz2.reshape(-1, P).t() @ tape["p1"].reshape(-1, tape["p1"].shape[2])
    dz1 = (dz2 @ w1) * (masks["prenet"][0].float() * 2) * (tape["p1"] > 0).float()
    g["decoder.prenet.layers.0.linear_layer.weight"] = dz1.reshape(-1, dz1.shape[2]).t() @ tape["frames"].reshape(-1, tape["frames"].shape[2])
    # encoder BiLSTM (packed semantics: inactive steps pass the state through)
    H = sd["encoder.lstm.weight_hh_l0"].shape[1]
    x = tape["x"]
    dx = torch.zeros_like(x)
    rows = torch.arange(B)
    for d, sfx in enumerate(("", "_reverse")):
        w_ih, w_hh = sd["encoder.lstm.weight_ih_l0" + sfx], sd["encoder.lstm.weight_hh_l0" + sfx]
        dw_ih, dw_hh, db = z(w_ih), z(w_hh), torch.zeros(4 * H)
        dh, dc = torch.zeros(B, H), torch.zeros(B, H)
        for s in reversed(range(L)):
            active, t_idx, sv, h_prev = tape["enc_lstm"][d][s]
            am = active[:, None].float()
            dh_tot = dh + dmemory[rows, t_idx, d * H:(d + 1) * H] * am
            dgt, dc_prev = _lstm_cell_bwd(dh_tot * am, dc * am, sv)
            dgt = dgt * am
            dw_ih += dgt.t() @ x[rows, t_idx]
            dw_hh += dgt.t() @ h_prev
            db += dgt.sum(0)
            dx[rows, t_idx] += dgt @ w_ih
            dh = dgt @ w_hh + dh * (1 - am)
            dc = dc_prev * am + dc * (1 - am)
        g["encoder.lstm.weight_ih_l0" + sfx], g["encoder.lstm.weight_hh_l0" + sfx] = dw_ih, dw_hh
        g["encoder.lstm.bias_ih_l0" + sfx], g["encoder.lstm.bias_hh_l0" + sfx] = db, db.clone()
    dconv, eg = convstack_train_backward(sd, "encoder.convolutions", ne, encoder_acts(ne), dx.transpose(1, 2).contiguous(), tape["enc_convs"], masks["encoder"])
    g.update(eg)
    demb = torch.zeros_like(sd["embedding.weight"])
    demb.index_add_(0, tok.reshape(-1), dconv.transpose(1, 2).reshape(-1, demb.shape[1]))
    g["embedding.weight"] = demb
    return g


def clip_grad_norm_(grads: Dict[str, torch.Tensor], max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_ (models/tts/tacotron2.py:521): total 2-norm over all gradients; scale by
    max_norm / (norm + 1e-6) when that is below one.  Returns the norm before clipping."""
    total = float(torch.sqrt(sum((v.double() ** 2).sum() for v in grads.values())))
    coef = max_norm / (total + 1e-6)
    if coef < 1.0:
        for k in grads:
            grads[k] = grads[k] * coef
    return total


def adam_step(params, grads, state, lr: float, weight_decay: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-8):
    """torch.optim.Adam as the reference configures it (models/tts/tacotron2.py:506-513): L2 weight decay added to the
    gradient, bias-corrected moments."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    for k, p in params.items():
        gk = grads[k] + weight_decay * p
        m = state.setdefault("m." + k, torch.zeros_like(p))
        vv = state.setdefault("v." + k, torch.zeros_like(p))
        m.mul_(betas[0]).add_(gk, alpha=1 - betas[0])
        vv.mul_(betas[1]).addcmul_(gk, gk, value=1 - betas[1])
        denom = (vv.sqrt() / (1 - betas[1] ** t) ** 0.5).add_(eps)
        p.addcdiv_(m, denom, value=-lr / (1 - betas[0] ** t))
