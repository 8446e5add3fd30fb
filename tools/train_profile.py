#!/usr/bin/env python3
"""Where a training step spends its time (phase by phase, synchronising between phases)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvox_amd import training, weights as gw
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2

mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, hop_length=256, log_func="np.log"), TextConfig(n_tokens=40)
m = Tacotron2(mc, ac, tc)
m.load_state_dict(gw.generate_state_dict(mc, ac, tc, seed=0))
m = m.to("cuda:0")
opt = m.get_optimizer()
B, L, T = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 128, 200)))
batch = {k: torch.from_numpy(v).cuda() for k, v in gw.synthetic_inputs(B, L, T, 40, 80, seed=3).items()}
m.train_step(batch, m.get_criterion(), opt)   # warm-up
def tick(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, time.perf_counter() - t0
m.train()
_, t_pack = tick(lambda: m._ensure_packed())
(out_tape), t_fwd = tick(lambda: m._forward_train(batch))
outputs, tape = out_tape
loss, t_loss = tick(lambda: m.get_criterion()["loss"](batch, outputs))
grads, t_bwd = tick(lambda: training.train_backward(m, batch, outputs, tape))
(norm_scale), t_clip = tick(lambda: training.clip_grad_norm(grads, mc.grad_clip_thresh))
_, t_adam = tick(lambda: opt["optimizer"].step(grads, norm_scale[1]))
print(f"B={B} L={L} T={T}: re-pack {t_pack*1e3:.1f} ms | forward {t_fwd*1e3:.1f} | loss {t_loss*1e3:.1f} | backward {t_bwd*1e3:.1f} | clip {t_clip*1e3:.1f} | adam {t_adam*1e3:.1f}")
