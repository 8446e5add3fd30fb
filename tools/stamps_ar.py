#!/usr/bin/env python3
"""Developer diagnostic (GPU box): where step 20 of the autoregressive resident loop spends its time - per workgroup stamps of the
tile kernel (dec_resident.hip, decoder_ar_resident_kernel) and of the attention rows (attn_persist.hip, AR role), relative to the
moment the first Prenet workgroup saw every row's layer 1 of the step.
    python -m genvox_amd.build --stamps && GVX_LIB=libgenvox_amd_stamps.so python tools/stamps_ar.py [B] [L]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from genvox_amd import _lib, weights as gw
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L = int(sys.argv[2]) if len(sys.argv) > 2 else 128
mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, log_func="np.log"), TextConfig(n_tokens=40)
mc.max_decoder_steps = 60
mc.gate_threshold = 1.0
m = Tacotron2(mc, ac, tc)
m.load_state_dict(gw.generate_state_dict(mc, ac, tc, 0))
m = m.to("cuda:0")
tok = torch.from_numpy(gw.synthetic_inputs(B, L, 8, tc.n_tokens, ac.n_mels, seed=3)["token_padded"]).cuda()
assert m.ar_loop_kind(B, L) == 2
for _ in range(3):
    m.inference({"tokens": tok})
    torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
wg = (C.c_ulonglong * 896)(); rows = (C.c_ulonglong * 512)()
assert lib.gvx_debug_read_wg_stamps_resident(wg, rows) == 0
wa = (C.c_ulonglong * 896)(); ra = (C.c_ulonglong * 256)()
assert lib.gvx_debug_read_stamps_ar(wa, ra) == 0
w = np.array(list(wg), dtype=np.int64).reshape(224, 4)
wx = np.array(list(wa), dtype=np.int64).reshape(224, 4)
r = np.array(list(rows), dtype=np.int64).reshape(64, 8)[:B]
rx = np.array(list(ra), dtype=np.int64).reshape(64, 4)[:B]
z = int(wx[64:72, 0].min())   # first Prenet workgroup has seen y1 of step 20


def span(a):
    return f"{(a.min() - z) * 10:7d} .. {(a.max() - z) * 10:7d} ns (median {int((np.median(a) - z) * 10)})"


print(f"autoregressive resident loop, B = {B}, L = {L}, step 20, relative to 'first Prenet workgroup saw y1':")
print(f"  Prenet workgroups (8)    y1 seen            {span(wx[64:72, 0])}")
print(f"                           flag stored        {span(wx[64:72, 1])}")
for name, sl in (("attention LSTM 48 rows", slice(0, 64)), ("attention LSTM 32 rows", slice(64, 96))):
    print(f"  {name:24s} step begins        {span(w[sl, 0])}")
    print(f"  {'':24s} Prenet seen        {span(wx[sl, 2])}")
    print(f"  {'':24s} Prenet columns done {span(wx[sl, 3])}")
    print(f"  {'':24s} flag stored        {span(w[sl, 3])}")
print(f"  attention rows           slabs (q) seen     {span(r[:, 1])}")
print(f"                           ctx stored         {span(r[:, 5])}")
print(f"                           ctx flag acked     {span(r[:, 7])}")
print(f"                           loc features done  {span(r[:, 6])}")
print(f"  decoder LSTM             step begins        {span(w[96:, 0])}")
print(f"                           gate 1 (h)         {span(w[96:, 1])}")
print(f"                           gate 2 (ctx)       {span(w[96:, 2])}")
print(f"                           flag stored        {span(w[96:, 3])}")
print(f"  attention rows           p slabs seen       {span(rx[:, 0])}")
print(f"                           frame summed       {span(rx[:, 1])}")
print(f"                           layer 1 stored     {span(rx[:, 2])}")
buf = (C.c_ulonglong * 480)()
assert lib.gvx_debug_read_stamps_resident(buf) == 0
ev = ["step begins", "gate 1 (h)", "part 1 done", "gate 2 (ctx)", "parts done", "sums synced", "cells synced", "stores issued", "stores drained", "barrier 3"]
for k, name in enumerate(("attention LSTM 48 rows (block 0)", "attention LSTM 32 rows + Prenet (block 64)", "decoder LSTM (block 96)")):
    print(name)
    for wv in range(8):
        v = [buf[(k * 10 + wv) * 16 + e] for e in range(10)]
        print(f"   wave {wv}: " + "  ".join(f"{(x - z) * 10:6d}" for x in v))
    print("   events:  " + " | ".join(ev))
