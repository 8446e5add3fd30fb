#!/usr/bin/env python3
"""Developer diagnostic (GPU box): phase timestamps inside the decoder-step kernels (stamps build).
    python -m genvox_amd.build --stamps && GVX_LIB=libgenvox_amd_stamps.so python tools/stamps.py [B]"""
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
# reset the launch-span slots (row 2, slots 20 / 21) by hand is not possible from here: they are min / max accumulators
# over every launch since the module was loaded; a fresh process per measurement keeps them meaningful
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 96)()
assert lib.gvx_debug_read_stamps(buf) == 0
one_launch = os.environ.get("GVX_ATTN_SPLIT", "0") != "1"
names = {0: ["start", "mainloop done", "red synced", "cell done", "qslab done"],
         1: (["start", "loads issued", "q summed", "energies done + synced", "softmax done + synced", "ctx written"] if one_launch else
             ["start", "loads issued", "q summed + synced", "tanh partials done", "reduced + stored"]),
         2: [] if one_launch else ["start", "softmax done", "ctx partial", "ctx written"]}
for k, title in enumerate(["decoder_lstm_step (wg 0)", "attn_step (row 0, slice 0)" if one_launch else "attn_energy (wg 0,0)", "attn_context (wg 0,0)"]):
    if not names[k]:
        continue
    v = [buf[k * 32 + i] for i in range(len(names[k]))]
    print(title)
    for i in range(1, len(v)):
        print(f"   {names[k][i]:22s} +{(v[i] - v[i - 1]) * 10} ns   (t = {(v[i] - v[0]) * 10} ns)")

# per-wave main-loop end times (rows 1/2, slots 8..15 are written by the LSTM kernel's block 0 / first decoder tile)
import ctypes as C2
lib2 = C.CDLL(_lib.LIB_PATH)
raw = (C.c_ulonglong * 96)()
lib2.gvx_debug_read_stamps_skinny.argtypes = [C.c_void_p]
if lib2.gvx_debug_read_stamps_skinny(raw) == 0:
    # NOTE: block 0 of EVERY skinny launch writes row 1 (and the first tile of a launch's second job row 2): after a whole
    # forward these are the stamps of the LAST launch that had such a block - the decoder-LSTM drain launch -, not of a
    # mid-sequence step; the per-workgroup spans below are the ones to read for the steady-state launch
    for row, nm in ((1, "block 0 of the last skinny launch"), (2, "first tile of the second job of the last 2-job launch")):
        # row 2 carries its own reference: block 0's start stamp of the launch the decoder tile ran in (slot 7); the last
        # launch of a sequence has decoder tiles only, so row 0 may belong to a different launch
        t0 = raw[2 * 32 + 7] if row == 2 else raw[0]
        ends = [(raw[row * 32 + 8 + w] - t0) * 10 for w in range(8)]
        print(f"{nm}: per-wave main-loop end (ns after block 0 start): {ends}  spread {max(ends) - min(ends)} ns")


spans = (C.c_ulonglong * 1024)()
lib2.gvx_debug_read_wg_spans.argtypes = [C.c_void_p]
if lib2.gvx_debug_read_wg_spans(spans) == 0:
    beg = [spans[i] for i in range(512)]
    end = [spans[512 + i] for i in range(512)]
    live = [i for i in range(512) if beg[i] and end[i]]
    if live:
        t0 = min(beg[i] for i in live)
        def grp(name, ids):
            ids = [i for i in ids if i in live]
            if ids:
                b = [(beg[i] - t0) * 10 for i in ids]
                e = [(end[i] - t0) * 10 for i in ids]
                worst = max(ids, key=lambda i: end[i])
                print(f"{name:16s} {len(ids):4d} workgroups: begin {min(b):6d}..{max(b):6d} ns, end {min(e):6d}..{max(e):6d} ns (last: block {worst})")
        print("last multi-job LSTM launch, per workgroup (ns after the first workgroup began):")
        if os.environ.get("GVX_ATTN_PERSISTENT", "1") != "0" and B > 32:
            grp("attention (2 bt)", range(0, 128))
            grp("decoder partial", range(128, 256))
            grp("decoder final", range(256, 384))
            for lo in range(0, 384, 32):   # in dispatch order: which blocks start late / end late
                grp(f"blocks {lo}..{lo + 31}", range(lo, lo + 32))
        elif os.environ.get("GVX_ATTN_PERSISTENT", "1") != "0" and B <= 32 and L > 128:
            grp("attention pairs", range(0, 64))
            grp("decoder tiles", range(64, 192))
        elif os.environ.get("GVX_ATTN_PERSISTENT", "1") != "0" and B <= 32:
            grp("attention 48 rows", range(0, 64))
            grp("attention 32 rows", range(64, 96))
            grp("decoder tiles", range(96, 224))
        else:
            grp("attention tiles", range(0, 128))
            grp("decoder tiles", range(128, 256))
            grp("location", range(256, 512))
