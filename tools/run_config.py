#!/usr/bin/env python3
"""Run ONE configuration a few times, for use under rocprofv3 (kernel trace / PMC passes):
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/run_config.py ar 64
    configurations: tf <B> | ar <B> | gl <B> | postnet <B>"""
import contextlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from genvox_amd import weights as gw
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2

kind, B = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, hop_length=256, log_func="np.log"), TextConfig(n_tokens=40)
L, T = 128, 800
if kind in ("tf", "ar", "postnet"):
    m = Tacotron2(mc, ac, tc)
    m.load_state_dict(gw.generate_state_dict(mc, ac, tc, seed=0))
    m = m.to("cuda:0")
if kind == "tf":
    batch = {k: torch.from_numpy(v).cuda() for k, v in gw.synthetic_inputs(B, L, T, 40, 80, seed=3).items()}
    fn = lambda: m.forward(batch)
elif kind == "ar":
    mc.gate_threshold = 1.0
    tok = torch.from_numpy(gw.synthetic_inputs(B, L, 8, 40, 80, seed=3)["token_padded"]).cuda()
    fn = lambda: m.inference({"tokens": tok})
elif kind == "postnet":
    mel = torch.randn(B, 80, T, device="cuda")
    fn = lambda: m.postnet_residual(mel)
elif kind == "gl":
    from genvox_amd.audio import AudioProcessor
    ap = AudioProcessor(ac)
    mag = ap.mel_to_magnitude(torch.randn(B, 80, T, device="cuda") * 1.5 - 4.0)
    fn = lambda: ap.griffin_lim(mag, n_iter=60, want_phase=False)
import time
with contextlib.redirect_stdout(sys.stderr):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
print(f"{kind} B={B}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call")
