#!/usr/bin/env python3
"""Developer diagnostic (GPU box): relative error of every gradient of one training step against the CPU oracle, sorted, at a
given shape (default: the benchmark batch shape 32 x 128 tokens, 10 steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvox_amd import training as tr, weights as gw
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2
from oracle import train_ref

B, L, T = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 128, 10)))
mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, log_func="np.log"), TextConfig(n_tokens=40)
rng = np.random.default_rng(11)
tl = np.sort(rng.integers(min(20, L), L + 1, B))[::-1].copy(); tl[0] = L
ml = rng.integers(min(3, T), T + 1, B); ml[min(5, B - 1)] = T
sd = gw.generate_state_dict(mc, ac, tc, seed=6, peaky_attention=True)
inp = gw.synthetic_inputs(B, L, T, tc.n_tokens, ac.n_mels, seed=41, token_lengths=tl, mel_lengths=ml)
batch = {k: torch.from_numpy(v) for k, v in inp.items()}
gen = torch.Generator().manual_seed(80)
E, A, D, P, C, M, n = mc.encoder_embedding_dim, mc.attention_rnn_dim, mc.decoder_rnn_dim, mc.prenet_dim, mc.postnet_embedding_dim, ac.n_mels, mc.postnet_n_convolutions
bern = lambda shape, p: (torch.rand(shape, generator=gen) >= p).to(torch.uint8)
masks = {"encoder": [bern((B, E, L), 0.5) for _ in range(mc.encoder_n_convolutions)], "prenet": bern((2, T + 1, B, P), 0.5),
         "attention_rnn": bern((T, B, A), mc.p_attention_dropout), "decoder_rnn": bern((T, B, D), mc.p_decoder_dropout),
         "postnet": [bern((B, C if i < n - 1 else M, T), 0.5) for i in range(n)]}
wo, wt = train_ref.train_forward(sd, batch, masks, mc)
want = train_ref.train_backward(sd, batch, masks, mc, wo, wt)
gb = dict(batch)
gb["train_keep_masks"] = {k: masks[k] for k in ("encoder", "attention_rnn", "decoder_rnn", "postnet")}
gb["prenet_keep_masks"] = masks["prenet"]
m = Tacotron2(mc, ac, tc); m.load_state_dict(sd); m = m.to("cuda:0"); m.train()
out, tape = m._forward_train(gb)
g = tr.train_backward(m, gb, out, tape)
rows = []
for k, ref in want.items():
    sc = float(ref.abs().max())
    rows.append((float((g[k].cpu() - ref).abs().max()) / max(sc, 1e-12), sc, k))
for e, sc, k in sorted(rows, reverse=True)[:16]:
    print(f"{e:9.2e}  scale {sc:9.2e}  {k}")
for k in ("alignments", "mel_outputs", "mel_outputs_postnet"):
    print(k, float((out[k].cpu() - wo[k]).abs().max()))
