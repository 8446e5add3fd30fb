#!/usr/bin/env python3
"""Developer diagnostic (GPU box): the bf16x3 GEMM from TWO streams at once (two host threads, each with its own model handle and
inputs, like the two 32-row lanes of a 64-row batch): every result is compared with that lane's single-stream reference."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvox_amd import weights as gw
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2

what = sys.argv[1] if len(sys.argv) > 1 else "postnet"
mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, log_func="np.log"), TextConfig(n_tokens=40)
models, inputs, refs = [], [], []
for lane in range(2):
    m = Tacotron2(mc, ac, tc); m.load_state_dict(gw.generate_state_dict(mc, ac, tc, lane)); m = m.to("cuda:0").eval()
    g = torch.Generator("cuda").manual_seed(10 + lane)
    x = torch.randn(32, 80, 800, device="cuda", generator=g) if what == "postnet" else torch.randint(0, 40, (32, 128), device="cuda", generator=g)
    f = (lambda m_=m, x_=x: m_.postnet_residual(x_)) if what == "postnet" else (lambda m_=m, x_=x: m_.encode(x_, None))
    models.append(f); refs.append(f().clone())
torch.cuda.synchronize()
bad = [0, 0]
def work(lane):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(30):
            out = models[lane]()
            st.synchronize()
            bad[lane] += int(not torch.equal(out, refs[lane]))
ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
[t.start() for t in ths]; [t.join() for t in ths]
print(f"{what}: runs differing from the single-stream reference: lane 0 {bad[0]}/30, lane 1 {bad[1]}/30")
