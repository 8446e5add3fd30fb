#!/usr/bin/env python3
"""Time Tacotron2.train_step on synthetic batches (profiles/r03_train_step_timing.txt)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvox_amd import weights as gw
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2

mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, hop_length=256, log_func="np.log"), TextConfig(n_tokens=40)
m = Tacotron2(mc, ac, tc)
m.load_state_dict(gw.generate_state_dict(mc, ac, tc, seed=0))
m = m.to("cuda:0")
opt = m.get_optimizer()
for B, L, T in ((8, 64, 100), (32, 128, 200), (64, 128, 200), (32, 128, 800)):
    batch = {k: torch.from_numpy(v).cuda() for k, v in gw.synthetic_inputs(B, L, T, 40, 80, seed=3).items()}
    times, losses = [], []
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.train_step(batch, m.get_criterion(), opt)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0); losses.append(round(m.loss_items["loss"], 4))
    print(f"train_step B={B} L={L} T={T}: {[round(t, 3) for t in times]} s per step, loss {losses}, grad_norm {m.grad_norm_val:.3f}, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
