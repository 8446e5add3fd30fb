#!/usr/bin/env python3
"""Developer diagnostic (GPU box): phase timestamps of the resident decoder kernel (dec_resident.hip) at decoder step 20, one
workgroup of each kind, every wave; the attention kernel's phases of the same step.
    python -m genvox_amd.build --stamps && GVX_LIB=libgenvox_amd_stamps.so GVX_TF_RESIDENT=1 python tools/stamps_resident.py [B] [T]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from genvox_amd import _lib, weights as gw
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
L = 128
mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, log_func="np.log"), TextConfig(n_tokens=40)
m = Tacotron2(mc, ac, tc)
m.load_state_dict(gw.generate_state_dict(mc, ac, tc, 0))
m = m.to("cuda:0")
batch = {k: torch.from_numpy(v).cuda() for k, v in gw.synthetic_inputs(B, L, T, 40, 80).items()}
lib = C.CDLL(_lib.LIB_PATH)
for _ in range(3):
    m.forward(batch)
    torch.cuda.synchronize()
buf = (C.c_ulonglong * 480)()
assert lib.gvx_debug_read_stamps_resident(buf) == 0
ev = ["step begins", "gate 1 (h)", "part 1 done", "gate 2 (ctx)", "part 2 done", "sums synced", "cells synced", "stores issued", "stores drained", "barrier 3"]
kinds = ["attention LSTM 48 rows (block 0)", "attention LSTM 32 rows (block 64)", "decoder LSTM (block 96)"]
t0 = min(buf[(k * 10 + w) * 16] for k in range(2) for w in range(8))
for k in range(3):
    print(kinds[k])
    base = t0 if k < 2 else min(buf[(k * 10 + w) * 16] for w in range(8))
    for w in range(8):
        v = [buf[(k * 10 + w) * 16 + e] for e in range(10)]
        print(f"   wave {w}: " + "  ".join(f"{(x - base) * 10:6d}" for x in v))
    print("   events:  " + " | ".join(ev))
pb = (C.c_ulonglong * 96)()
assert lib.gvx_debug_read_stamps_persist(pb) == 0
names = ["step begins", "slabs flagged + barrier", "q summed", "energies", "softmax + context partials", "context published", "location features"]
v = [pb[i] for i in range(len(names))]
print(f"persistent attention, row 0, step 20 (begin at {(v[0] - t0) * 10} ns after the attention-LSTM step began)")
for i in range(1, len(v)):
    print(f"   {names[i]:28s} +{(v[i] - v[i - 1]) * 10} ns   (t = {(v[i] - t0) * 10} ns)")

wg = (C.c_ulonglong * 896)()
rows = (C.c_ulonglong * 512)()
assert lib.gvx_debug_read_wg_stamps_resident(wg, rows) == 0
import numpy as np
w = np.array(list(wg), dtype=np.int64).reshape(224, 4)
r = np.array(list(rows), dtype=np.int64).reshape(64, 8)[:B]
z = int(w[:96, 3].min())   # first attention-LSTM flag of the step
def span(a): return f"{(a.min() - z) * 10:7d} .. {(a.max() - z) * 10:7d} ns (median {int((np.median(a) - z) * 10)})"
print("per workgroup, step 20, relative to the first attention-LSTM flag of the step:")
for name, sl in (("attention LSTM 48 rows", slice(0, 64)), ("attention LSTM 32 rows", slice(64, 96)), ("decoder LSTM", slice(96, 224))):
    print(f"  {name:24s} step begins {span(w[sl, 0])}")
    print(f"  {'':24s} gate 1      {span(w[sl, 1])}")
    print(f"  {'':24s} gate 2      {span(w[sl, 2])}")
    print(f"  {'':24s} flag stored {span(w[sl, 3])}")
print(f"  attention rows: step begins {span(r[:, 0])}")
print(f"  {'':14s}  slabs seen  {span(r[:, 1])}")
print(f"  {'':14s}  ctx stored  {span(r[:, 5])}")
print(f"  {'':14s}  flag acked  {span(r[:, 7])}")


ls = (C.c_ulonglong * 32)()
if lib.gvx_debug_read_loc_stamps(ls) == 0:
    print("location features of step 20, row 0, waves 0 / 5 / 10 / 15: conv done | conv barrier passed | dense done | last barrier passed (ns after wave 0's conv)")
    base = ls[0]
    for w in range(4):
        print(f"   wave {5 * w:2d}: " + "  ".join(f"{(ls[w * 8 + e] - base) * 10:6d}" for e in range(4)))
