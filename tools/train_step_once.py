#!/usr/bin/env python3
"""A few training steps with pauses in between (for kernel timelines: rocprofv3 --kernel-trace ... -- python3 tools/train_step_once.py,
then python3 tools/kernel_gaps.py DIR)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvox_amd import weights as gw
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2
mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, hop_length=256, log_func="np.log"), TextConfig(n_tokens=40)
m = Tacotron2(mc, ac, tc); m.load_state_dict(gw.generate_state_dict(mc, ac, tc, seed=0)); m = m.to("cuda:0")
opt = m.get_optimizer()
B, L, T = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 128, 200)))
batch = {k: torch.from_numpy(v).cuda() for k, v in gw.synthetic_inputs(B, L, T, 40, 80, seed=3).items()}
for i in range(4):
    torch.cuda.synchronize(); time.sleep(0.1); t0 = time.perf_counter()
    m.train_step(batch, m.get_criterion(), opt)
    torch.cuda.synchronize(); print(f"step {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", file=sys.stderr)
