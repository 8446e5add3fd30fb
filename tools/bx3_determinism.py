#!/usr/bin/env python3
"""Developer diagnostic (GPU box): is the bf16x3 GEMM bitwise reproducible?  The Postnet (five convolutions as GEMMs) and the
encoder run N times on fixed inputs; every run is compared with the first bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvox_amd import weights as gw
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2

mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, log_func="np.log"), TextConfig(n_tokens=40)
m = Tacotron2(mc, ac, tc); m.load_state_dict(gw.generate_state_dict(mc, ac, tc, 0)); m = m.to("cuda:0").eval()
B, T, L = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 800, 128
mel = torch.randn(B, 80, T, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
tok = torch.randint(0, 40, (B, L), device="cuda", generator=torch.Generator("cuda").manual_seed(2))
ref_p, ref_e = m.postnet_residual(mel).clone(), m.encode(tok, None).clone()
bad_p = bad_e = 0
for i in range(20):
    bad_p += int(not torch.equal(m.postnet_residual(mel), ref_p))
    bad_e += int(not torch.equal(m.encode(tok, None), ref_e))
d = (m.postnet_residual(mel) - ref_p).abs().max().item()
print(f"B={B}: postnet runs differing from the first: {bad_p}/20 (last max abs diff {d:.3e}); encoder: {bad_e}/20")
