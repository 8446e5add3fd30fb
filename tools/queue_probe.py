#!/usr/bin/env python3
"""Developer diagnostic (GPU box): per-call times of the teacher-forced forward and the autoregressive decode, call by call
(stream / hardware-queue placement problems show up as alternating or bimodal times)."""
import contextlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvox_amd import weights as gw
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2

mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, hop_length=256, log_func="np.log"), TextConfig(n_tokens=40)
mc.max_decoder_steps, mc.gate_threshold = 200, 1.0
m = Tacotron2(mc, ac, tc)
m.load_state_dict(gw.generate_state_dict(mc, ac, tc, seed=0))
m = m.to("cuda:0")
T, L = 200, 128
def t_ms(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
b32 = {k: torch.from_numpy(v).cuda() for k, v in gw.synthetic_inputs(32, L, T, 40, 80, seed=3).items()}
tok1 = torch.from_numpy(gw.synthetic_inputs(1, L, 8, 40, 80, seed=3)["token_padded"]).cuda()
tok64 = torch.from_numpy(gw.synthetic_inputs(64, L, 8, 40, 80, seed=3)["token_padded"]).cuda()
with contextlib.redirect_stdout(sys.stderr):
    tf = [round(t_ms(lambda: m.forward(b32)), 2) for _ in range(8)]
    ar1 = [round(t_ms(lambda: m.inference({"tokens": tok1})), 2) for _ in range(6)]
    ar64 = [round(t_ms(lambda: m.inference({"tokens": tok64})), 2) for _ in range(6)]
    tf2 = [round(t_ms(lambda: m.forward(b32)), 2) for _ in range(4)]
m.check_status()
print(os.environ.get("TAG", ""), "tf 32x200 ms:", tf, "| ar1 x200 ms:", ar1, "| ar64 x200 ms:", ar64, "| tf again:", tf2)
