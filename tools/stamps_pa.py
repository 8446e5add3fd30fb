#!/usr/bin/env python3
"""Developer diagnostic (GPU box): phase timestamps of the persistent attention kernel at decoder step 20 (stamps build), next
to the last LSTM launch's (workgroup 0).
    python -m genvox_amd.build --stamps && GVX_LIB=libgenvox_amd_stamps.so python tools/stamps_pa.py [B]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from genvox_amd import _lib, weights as gw
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T, L = 60, int(sys.argv[2]) if len(sys.argv) > 2 else 128
mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, log_func="np.log"), TextConfig(n_tokens=40)
m = Tacotron2(mc, ac, tc)
m.load_state_dict(gw.generate_state_dict(mc, ac, tc, 0))
m = m.to("cuda:0")
batch = {k: torch.from_numpy(v).cuda() for k, v in gw.synthetic_inputs(B, L, T, 40, 80).items()}
for _ in range(2):
    m.forward(batch)
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 96)()
assert lib.gvx_debug_read_stamps_persist(buf) == 0
names = ["step begins", "slabs signalled + barrier", "q summed", "energies", "softmax + context partials", "context published", "location features"]
v = [buf[i] for i in range(len(names))]
print("persistent attention, row 0, step 20")
for i in range(1, len(v)):
    print(f"   {names[i]:28s} +{(v[i] - v[i - 1]) * 10} ns   (t = {(v[i] - v[0]) * 10} ns)")
