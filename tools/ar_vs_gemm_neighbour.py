#!/usr/bin/env python3
"""Developer diagnostic (GPU box): one thread decodes 32 rows autoregressively again and again (graph replays), a second thread
keeps dense products of ANOTHER model in flight on its own stream (Postnet: GEMMs only).  Does the decode stay bit-identical?"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvox_amd import weights as gw
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2

S = 200
mc, ac, tc = Tacotron2Config(gate_threshold=1.0, max_decoder_steps=S), AudioConfig(filter_length=1024, hop_length=256, log_func="np.log"), TextConfig(n_tokens=40)
a = Tacotron2(mc, ac, tc); a.load_state_dict(gw.generate_state_dict(mc, ac, tc, seed=0)); a = a.to("cuda:0")
b = Tacotron2(mc, ac, tc); b.load_state_dict(gw.generate_state_dict(mc, ac, tc, seed=1)); b = b.to("cuda:0").eval()
tok = torch.from_numpy(gw.synthetic_inputs(32, 128, 8, 40, 80, seed=3)["token_padded"])
masks = torch.from_numpy(gw.prenet_keep_masks(S * 32, mc.prenet_dim, seed=11)).reshape(2, S, 32, mc.prenet_dim)
inp = {"tokens": tok, "prenet_keep_masks": masks}
VICTIM = os.environ.get("VICTIM", "ar")
if VICTIM == "tf":    # teacher-forced forward, 32 x 200 frames
    a.eval()
    if os.environ.get("VICTIM_RESIDENT") == "0":
        from genvox_amd import _lib
        _lib.check(_lib.load().gvx_model_set_persistent_attention(a._handle, 0))
    raw = gw.synthetic_inputs(32, 128, S, 40, 80, seed=3)
    batch = {k: torch.from_numpy(v) for k, v in raw.items()}
    batch["prenet_keep_masks"] = torch.from_numpy(gw.prenet_keep_masks((S + 1) * 32, mc.prenet_dim, seed=11))
    run_victim = lambda: a.forward(batch)
    KEY = "mel_outputs_postnet"
else:
    run_victim = lambda: a.inference(inp)
    KEY = "alignments"
ref = run_victim(); ref = run_victim(); ref = {k: v.clone() for k, v in run_victim().items() if torch.is_tensor(v)}
mel = torch.randn(32, 80, 800, device="cuda")
if os.environ.get("NEIGHBOUR") == "gl":
    from genvox_amd.audio import AudioProcessor
    ap = AudioProcessor(ac, device="cuda:0")
    mag_gl = torch.rand(32, 513, 800, device="cuda")
tok_dev = tok.cuda()
stop = False
def neighbour():
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        kind = os.environ.get("NEIGHBOUR", "postnet")
        x16 = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16); x32 = torch.randn(4096, 4096, device="cuda")
        while not stop:
            if kind == "postnet": b.postnet_residual(mel)
            elif kind == "mm_bf16": torch.mm(x16, x16)
            elif kind == "mm_f32": torch.mm(x32, x32)
            elif kind == "elementwise": torch.tanh(x32)
            elif kind == "small_mm_bf16": torch.mm(x16[:256, :256], x16[:256, :256])
            elif kind == "encoder": b.encode(tok_dev, None)
            elif kind == "gl": ap.griffin_lim(mag_gl, n_iter=8)
            st.synchronize()
th = threading.Thread(target=neighbour); th.start()
bad = 0
for i in range(12):
    out = run_victim()
    same = torch.equal(out[KEY], ref[KEY])
    bad += int(not same)
    if not same and os.environ.get("VERBOSE") and VICTIM == "ar":
        da = (out["alignments"] != ref["alignments"]).any(dim=2)          # [B, T]
        dm = (out["mel_outputs"] != ref["mel_outputs"]).any(dim=1)        # [B, T]
        dg = out["gate_outputs"] != ref["gate_outputs"]
        ta = int(da.any(dim=0).float().argmax()); tm = int(dm.any(dim=0).float().argmax()); tg = int(dg.any(dim=0).float().argmax())
        print(f"  run {i}: first differing step: alignments {ta} rows {da[:, ta].nonzero().flatten().tolist()} "
              f"max |d| {float((out['alignments'][:, ta] - ref['alignments'][:, ta]).abs().max()):.3e}; mel {tm} rows {dm[:, tm].nonzero().flatten().tolist()} "
              f"max |d| {float((out['mel_outputs'][:, :, tm] - ref['mel_outputs'][:, :, tm]).abs().max()):.3e}; gate {tg}", flush=True)
stop = True; th.join()
print(f"decodes differing from the undisturbed reference with GEMMs of another model in flight: {bad}/12")
