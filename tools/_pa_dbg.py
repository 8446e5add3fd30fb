import os, sys
sys.path.insert(0, os.getcwd())
import torch
from genvox_amd import weights as gw
from genvox_amd.tacotron2 import Tacotron2
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, hop_length=256, log_func="np.log"), TextConfig(n_tokens=40)
m = Tacotron2(mc, ac, tc); m.load_state_dict(gw.generate_state_dict(mc, ac, tc, seed=0)); m = m.to("cuda:0")
B, L, T = int(os.environ.get("DBG_B", "32")), 128, 800
batch = {k: torch.from_numpy(v).cuda() for k, v in gw.synthetic_inputs(B, L, T, 40, 80, seed=3).items()}
masks = torch.from_numpy(gw.prenet_keep_masks((T + 1) * B, mc.prenet_dim, seed=11)).reshape(2, T + 1, B, mc.prenet_dim).cuda()
batch["prenet_keep_masks"] = masks
from genvox_amd import _lib
lib = _lib.load()
m._ensure_handle(); outs = {}
for tag, en in (("pa1", 1), ("pa2", 1), ("off", 0), ("off2", 0)):
    _lib.check(lib.gvx_model_set_persistent_attention(m._handle, en))
    o = m.forward(batch); torch.cuda.synchronize()
    import ctypes as C
    st = (C.c_int32 * 2)()
    lib.gvx_workspace_status(m._handle, m._workspace.data_ptr(), m._workspace.numel(), None, st)
    print(tag, "status", st[0], hex(st[1]))
    outs[tag] = {k: v.clone() for k, v in o.items()}
for a, b in (("pa1", "pa2"), ("pa1", "off"), ("off", "off2")):
    for k in outs[a]:
        d = (outs[a][k] - outs[b][k]).abs()
        print(a, b, k, "max", float(d.max()), "ndiff", int((d > 0).sum()))
al1, al0 = outs["pa1"]["alignments"], outs["off"]["alignments"]
al2 = outs["pa2"]["alignments"]
for t in range(6):
    print("step", t, "pa1 vs pa2 align maxdiff", float((al1[:, t] - al2[:, t]).abs().max()))
r = 0
print("row0 step1 pa1-off:", (al1[r, 1] - al0[r, 1])[:16].tolist())
print("row0 step1 off    :", al0[r, 1][:16].tolist())
print("row0 step0 off    :", al0[r, 0][:16].tolist())
for t in range(6):
    d = (al1[:, t] - al0[:, t]).abs()
    print("step", t, "align maxdiff", float(d.max()), "rows", torch.nonzero(d.amax(dim=1) > 1e-6).flatten().tolist()[:40])
    g = (outs["pa1"]["gate_outputs"][:, t] - outs["off"]["gate_outputs"][:, t]).abs()
    print("        gate maxdiff", float(g.max()), "rows", torch.nonzero(g > 1e-5).flatten().tolist()[:40])
