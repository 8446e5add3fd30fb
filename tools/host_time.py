#!/usr/bin/env python3
"""Developer diagnostic (GPU box): host time of one asynchronous Tacotron2.forward call (everything the call enqueues) next to
the GPU time per forward - how far the host runs ahead of the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvox_amd import weights as gw
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2

B, L, T = 32, 128, 800
mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, log_func="np.log"), TextConfig(n_tokens=40)
m = Tacotron2(mc, ac, tc)
m.load_state_dict(gw.generate_state_dict(mc, ac, tc, 0))
m = m.to("cuda:0").eval()
batch = {k: torch.from_numpy(v).cuda() for k, v in gw.synthetic_inputs(B, L, T, 40, 80).items()}
for _ in range(3):
    m.forward(batch)
torch.cuda.synchronize()
m.enable_stage_timing(True)
host, t0 = [], time.perf_counter()
for _ in range(10):
    h0 = time.perf_counter(); m.forward(batch); host.append(time.perf_counter() - h0)
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / 10
print(f"host time per forward call: {[round(h * 1e3, 2) for h in host]} ms; wall per forward {total * 1e3:.2f} ms; stages {m.stage_times_ms()}")
