#!/usr/bin/env python3
"""Developer diagnostic (GPU box): phase timestamps of the two BPTT kernels of the training step (stamps build), workgroup
(0, 0) of the launch at mid-sequence.
    python -m genvox_amd.build --stamps && GVX_LIB=libgenvox_amd_stamps.so python tools/stamps_train.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from genvox_amd import _lib, weights as gw
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2

B, L, T = 32, 128, 60
mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, log_func="np.log"), TextConfig(n_tokens=40)
m = Tacotron2(mc, ac, tc)
m.load_state_dict(gw.generate_state_dict(mc, ac, tc, 0))
m = m.to("cuda:0")
batch = {k: torch.from_numpy(v).cuda() for k, v in gw.synthetic_inputs(B, L, T, 40, 80, seed=3).items()}
opt = m.get_optimizer()
for _ in range(2):
    m.train_step(batch, m.get_criterion(), opt)
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 96)()
assert lib.gvx_debug_read_stamps_train(buf) == 0
for k, (title, names) in enumerate((
        ("bptt_attention_kernel", ["start", "operands staged", "row sum s", "dw / de", "location features", "energies", "dq / dv", "d dense",
                                   "dlocf", "t1", "d conv", "end"]),
        ("encoder_bptt_step_kernel", ["start", "weights staged", "dgates walked", "h_prev walked", "combined", "end"]))):
    v = [buf[32 * k + i] for i in range(len(names))]
    print(title)
    for i in range(1, len(v)):
        print(f"   {names[i]:22s} +{(v[i] - v[i - 1]) * 10:6d} ns   (t = {(v[i] - v[0]) * 10} ns)")
