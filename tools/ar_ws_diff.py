#!/usr/bin/env python3
"""Developer diagnostic (GPU box): which workspace buffers of a short autoregressive decode (STEPS env, default 1) differ from
an undisturbed run when another model's GEMMs are in flight on a second stream?  Run with GVX_DEBUG_PLAN=1 (the library then
prints the byte offsets of the workspace buffers to stderr; this tool re-reads them from its own stderr copy)."""
import os, sys, threading, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if os.environ.get("AR_WS_CHILD") != "1":   # parent: run the child with the plan print captured
    env = dict(os.environ, AR_WS_CHILD="1", GVX_DEBUG_PLAN="1")
    r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
    print(r.stdout, end="")
    sys.stderr.write("\n".join(l for l in r.stderr.splitlines() if not l.startswith("wsplan ")) + "\n")
    sys.exit(r.returncode)

import torch
from genvox_amd import weights as gw
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2

S = int(os.environ.get("STEPS", "1"))
mc, ac, tc = Tacotron2Config(gate_threshold=1.0, max_decoder_steps=S), AudioConfig(filter_length=1024, hop_length=256, log_func="np.log"), TextConfig(n_tokens=40)
a = Tacotron2(mc, ac, tc); a.load_state_dict(gw.generate_state_dict(mc, ac, tc, seed=0)); a = a.to("cuda:0")
b = Tacotron2(mc, ac, tc); b.load_state_dict(gw.generate_state_dict(mc, ac, tc, seed=1)); b = b.to("cuda:0").eval()
tok = torch.from_numpy(gw.synthetic_inputs(32, 128, 8, 40, 80, seed=3)["token_padded"])
masks = torch.from_numpy(gw.prenet_keep_masks(S * 32, mc.prenet_dim, seed=11)).reshape(2, S, 32, mc.prenet_dim)
inp = {"tokens": tok, "prenet_keep_masks": masks}
import tempfile
tmp = tempfile.TemporaryFile(mode="w+")
saved = os.dup(2); os.dup2(tmp.fileno(), 2)
for _ in range(3): a.inference(inp)
os.dup2(saved, 2)
tmp.seek(0)
torch.cuda.synchronize()
ref = a._workspace.clone().view(torch.int32)
PLAN = sorted((int(l.split()[2]), l.split()[1]) for l in tmp.read().splitlines() if l.startswith("wsplan ") and l.split()[1] != "total")
mel = torch.randn(32, 80, 800, device="cuda")
stop = False
def neighbour():
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        while not stop:
            b.postnet_residual(mel); st.synchronize()
th = threading.Thread(target=neighbour); th.start()
seen = 0
for i in range(40):
    a.inference(inp); torch.cuda.synchronize()
    cur = a._workspace.view(torch.int32)
    d = (cur != ref).nonzero().flatten()
    if d.numel() == 0: continue
    seen += 1
    if seen > 3: continue
    print(f"run {i}: {d.numel()} differing words")
    for (o, n), (o2, _) in zip(PLAN, PLAN[1:] + [(cur.numel() * 4, "end")]):
        if o2 <= o: continue
        x, y = cur[o // 4:o2 // 4], ref[o // 4:o2 // 4]
        ne = x != y
        k = int(ne.sum())
        if k == 0: continue
        xf, yf = x.view(torch.float32)[ne], y.view(torch.float32)[ne]
        zero_flip = int(((xf == 0) != (yf == 0)).sum())
        first = int(ne.nonzero()[0])
        print(f"   {n:10s} {k:7d} of {ne.numel():9d} words differ (first at word {first}); max |float diff| {float((xf - yf).abs().max()):.3e}; zero<->nonzero {zero_flip}")
stop = True; th.join()
print(f"{seen}/40 runs left a workspace that differs from the undisturbed one")
