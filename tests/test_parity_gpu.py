"""GPU parity tests: the HIP path (through the C ABI) against the reference-generated golden fixtures
and against the oracle on the same seeded inputs.  Tolerance 1e-3 fp32 (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from genvox_amd import weights as gw
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2
from oracle import tacotron2_ref
from tests.golden.cases import AR_CASES, TF_CASES, case_configs
from tests.helpers import TOL, case_state_dict, load_fixture, max_abs_diff, tf_batch, unpack_masks

pytestmark = pytest.mark.gpu
KEYS = ("alignments", "gate_outputs", "mel_outputs", "mel_outputs_postnet")


def gpu_model(name, case, **overrides):
    mc, ac, tc = case_configs(case)
    for k, v in overrides.items():
        setattr(mc, k, v)
    m = Tacotron2(mc, ac, tc)
    m.load_state_dict(case_state_dict(name))
    return m.to("cuda:0"), (mc, ac, tc)


@pytest.mark.parametrize("name", list(TF_CASES))
def test_teacher_forced_matches_reference_fixture(name):
    case, fx = TF_CASES[name], load_fixture(name)
    m, (mc, ac, tc) = gpu_model(name, case)
    B, T = case["B"], case["T"]
    masks = unpack_masks(fx["keep_masks_packed"], (2, (T + 1) * B, mc.prenet_dim))
    mem = m.encode(torch.from_numpy(fx["token_padded"]), torch.from_numpy(fx["token_lengths"]))
    assert max_abs_diff(mem, fx["encoder_outputs"]) <= TOL
    out = m.forward({**tf_batch(fx), "prenet_keep_masks": masks})
    for k in KEYS:
        assert out[k].shape == fx[k].shape
        d = max_abs_diff(out[k], fx[k])
        assert d <= TOL, f"{name}/{k}: {d}"
    # padding semantics of the reference (models/tts/tacotron2.py:466-473, packed encoder, masked softmax)
    for b in range(B):
        ml, tl = int(fx["mel_lengths"][b]), int(fx["token_lengths"][b])
        assert torch.all(out["mel_outputs"][b, :, ml:] == 0) and torch.all(out["mel_outputs_postnet"][b, :, ml:] == 0)
        assert torch.all(out["gate_outputs"][b, ml:] == 1e3)
        assert torch.all(out["alignments"][b, :, tl:] == 0)
        assert torch.all(mem[b, tl:] == 0)


@pytest.mark.parametrize("name", list(TF_CASES))
def test_teacher_forced_matches_oracle_and_stages(name):
    case, fx = TF_CASES[name], load_fixture(name)
    m, (mc, ac, tc) = gpu_model(name, case)
    sd = case_state_dict(name)
    B, T = case["B"], case["T"]
    masks = unpack_masks(fx["keep_masks_packed"], (2, (T + 1) * B, mc.prenet_dim))
    want = tacotron2_ref.tacotron2_forward(sd, tf_batch(fx), masks, mask_padding=True)
    got = m.forward({**tf_batch(fx), "prenet_keep_masks": masks})
    for k in KEYS:
        assert max_abs_diff(got[k], want[k]) <= TOL, k
    mel = torch.from_numpy(fx["mel_outputs"])
    assert max_abs_diff(m.postnet_residual(mel), mel + tacotron2_ref.postnet(sd, mel)) <= TOL


@pytest.mark.parametrize("name", list(AR_CASES))
def test_autoregressive_matches_reference_fixture(name):
    case, fx = AR_CASES[name], load_fixture(name)
    steps = int(fx["max_decoder_steps"])
    m, (mc, ac, tc) = gpu_model(name, case, max_decoder_steps=steps, gate_threshold=float(fx["gate_threshold"]))
    masks = unpack_masks(fx["keep_masks_packed"], (2, steps, mc.prenet_dim)).reshape(2, steps, 1, mc.prenet_dim)
    out = m.inference({"tokens": torch.from_numpy(fx["tokens"]).int(), "prenet_keep_masks": masks})
    assert out["mel_outputs"].shape == fx["mel_outputs"].shape, "stop step differs from the reference"
    for k in KEYS:
        d = max_abs_diff(out[k], fx[k])
        assert d <= TOL, f"{name}/{k}: {d}"


def test_batched_autoregressive_rows_match_batch1_runs():
    """The reference decodes batch 1 only; the batched extension must give every row what a batch-1 run gives."""
    name = "ar_full_gate"
    case, fx = AR_CASES[name], load_fixture(name)
    steps = 24
    m, (mc, ac, tc) = gpu_model(name, case, max_decoder_steps=steps, gate_threshold=float(fx["gate_threshold"]))
    sd = case_state_dict(name)
    lens = [30, 19, 7]
    B, L = len(lens), max(lens)
    tok = (gw.hashed_uniform(77, "bar", B * L) * tc.n_tokens).astype(np.int64).reshape(B, L)
    for b, n in enumerate(lens):
        tok[b, n:] = 0
    masks = torch.from_numpy(gw.prenet_keep_masks(steps * B, mc.prenet_dim, seed=5)).reshape(2, steps, B, mc.prenet_dim)
    out = m.inference({"tokens": torch.from_numpy(tok), "token_lengths": torch.tensor(lens), "prenet_keep_masks": masks})
    n_frames = out["mel_lengths"].cpu().tolist()
    for b, n in enumerate(lens):
        # padded-row semantics of the reference's batched forward (conv over pad embeddings, packed LSTM, masked softmax)
        want = tacotron2_ref.tacotron2_inference(sd, torch.from_numpy(tok[b:b + 1]), masks[:, :, b],
                                                 float(fx["gate_threshold"]), steps, token_length=n)
        nf = want["mel_outputs"].shape[2]
        assert n_frames[b] == nf, (b, n_frames[b], nf)
        assert max_abs_diff(out["mel_outputs"][b:b + 1, :, :nf], want["mel_outputs"]) <= TOL
        assert max_abs_diff(out["gate_outputs"][b:b + 1, :nf], want["gate_outputs"]) <= TOL
        assert max_abs_diff(out["alignments"][b:b + 1, :nf], want["alignments"]) <= TOL


def test_full_size_properties_and_long_horizon_parity():
    """BASELINE config 2 (B=32, T=800, L=128): size-independent properties on the whole batch, and direct
    oracle parity over the full 800-step horizon on two rows (drift check)."""
    mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, log_func="np.log"), TextConfig(n_tokens=40)
    sd = gw.generate_state_dict(mc, ac, tc, seed=0)
    m = Tacotron2(mc, ac, tc)
    m.load_state_dict(sd)
    m = m.to("cuda:0")
    B, L, T, P = 32, 128, 800, mc.prenet_dim
    tl = np.sort(128 - (np.arange(B) * 3) % 50)[::-1].copy()
    ml = 800 - (np.arange(B) * 37) % 300
    ml[0] = 800
    inp = gw.synthetic_inputs(B, L, T, tc.n_tokens, ac.n_mels, seed=3, token_lengths=tl, mel_lengths=ml)
    batch = {k: torch.from_numpy(v) for k, v in inp.items()}
    masks = torch.from_numpy(gw.prenet_keep_masks((T + 1) * B, P, seed=11)).reshape(2, T + 1, B, P)
    out = m.forward({**batch, "prenet_keep_masks": masks})
    out2 = m.forward({**batch, "prenet_keep_masks": masks})
    for k in KEYS:  # deterministic: no atomics, fixed reduction order
        assert torch.equal(out[k], out2[k]), k
    a = out["alignments"]
    assert torch.all(torch.isfinite(out["mel_outputs_postnet"]))
    assert float((a.sum(dim=2) - 1).abs().max()) <= 1e-5
    for b in range(B):
        assert torch.all(a[b, :, int(tl[b]):] == 0)
    # batch rows are independent: a sub-batch run alone reproduces its rows
    rows = [0, 5, 17, 31]
    sub = {k: v[rows] for k, v in batch.items()}
    sub["token_padded"] = sub["token_padded"][:, :int(tl[rows].max())]
    out_sub = m.forward({**sub, "prenet_keep_masks": masks[:, :, rows].contiguous()})
    for k in ("gate_outputs", "mel_outputs", "mel_outputs_postnet"):
        assert max_abs_diff(out_sub[k], out[k][rows]) <= 1e-4, k
    assert max_abs_diff(out_sub["alignments"], out["alignments"][rows][:, :, :int(tl[rows].max())]) <= 1e-4
    # long-horizon parity against the oracle on two rows
    rows = [0, 31]
    sub = {k: v[rows] for k, v in batch.items()}
    want = tacotron2_ref.tacotron2_forward(sd, sub, masks[:, :, rows].reshape(2, -1, P), mask_padding=True)
    for k in KEYS:
        d = max_abs_diff(out[k][rows], want[k])
        assert d <= TOL, f"{k}: {d}"


@pytest.mark.parametrize("B,L,T,dims", [
    (40, 50, 6, "full"),      # B > 32: two batch tiles per MFMA pass (MT = 2), G = 2 attention groups
    (64, 16, 4, "full"),      # maximum rows per call, minimum useful L (G = 2)
    (2, 300, 5, "full"),      # long rows: several passes per attention workgroup, context prefetch reloads
    (1, 7, 3, "small"),       # single short row, reduced dims
    (33, 21, 3, "small"),     # MT = 2 with reduced dims and ragged lengths
])
def test_shapes_against_oracle(B, L, T, dims):
    case = TF_CASES["tf_full" if dims == "full" else "tf_small"]
    name = "tf_full" if dims == "full" else "tf_small"
    m, (mc, ac, tc) = gpu_model(name, case)
    sd = case_state_dict(name)
    rng = np.random.default_rng(B * 1000 + L)
    tl = np.sort(rng.integers(max(1, L // 3), L + 1, size=B))[::-1].copy()
    tl[0] = L
    ml = rng.integers(1, T + 1, size=B)
    ml[rng.integers(0, B)] = T
    inp = gw.synthetic_inputs(B, L, T, tc.n_tokens, ac.n_mels, seed=9, token_lengths=tl, mel_lengths=ml)
    batch = {k: torch.from_numpy(v) for k, v in inp.items()}
    masks = torch.from_numpy(gw.prenet_keep_masks((T + 1) * B, mc.prenet_dim, seed=B + L))
    want = tacotron2_ref.tacotron2_forward(sd, batch, masks, mask_padding=True)
    got = m.forward({**batch, "prenet_keep_masks": masks})
    for k in KEYS:
        d = max_abs_diff(got[k], want[k])
        assert d <= TOL, f"B={B} L={L} T={T} {k}: {d}"


def test_autoregressive_two_batch_tiles_in_one_call():
    """B = 36 rows in ONE C-ABI call (two batch tiles per MFMA pass, MT = 2, of the autoregressive launches) - the host
    mirror would chunk this batch, so the call is made directly; first, middle and last row against batch-1 oracle runs."""
    import ctypes as C

    from genvox_amd import _lib
    name = "ar_small_gate"
    case, fx = AR_CASES[name], load_fixture(name)
    steps = 10
    m, (mc, ac, tc) = gpu_model(name, case, max_decoder_steps=steps, gate_threshold=1.0)
    sd = case_state_dict(name)
    B, L = 36, 11
    tok = (gw.hashed_uniform(5, "ar36", B * L) * tc.n_tokens).astype(np.int64).reshape(B, L)
    masks = torch.from_numpy(gw.prenet_keep_masks(steps * B, mc.prenet_dim, seed=3)).reshape(2, steps, B, mc.prenet_dim)
    lib = _lib.load()
    m._ensure_packed()
    dev = torch.device("cuda:0")
    ws = m._get_workspace(B, L, steps)
    tk, mk = torch.from_numpy(tok).to(dev), masks.to(dev).contiguous()
    memory = torch.empty(B, L, mc.encoder_embedding_dim, device=dev)
    mel = torch.zeros(B, ac.n_mels, steps, device=dev)
    gate = torch.zeros(B, steps, device=dev)
    align = torch.zeros(B, steps, L, device=dev)
    nf = torch.zeros(B, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ran = C.c_int(0)
    _lib.check(lib.gvx_encoder_forward(m._handle, tk.data_ptr(), None, B, L, memory.data_ptr(), ws.data_ptr(), ws.numel(), st))
    _lib.check(lib.gvx_decoder_autoregressive(m._handle, memory.data_ptr(), None, B, L, steps, 1.0, mk.data_ptr(), mel.data_ptr(),
                                              gate.data_ptr(), align.data_ptr(), nf.data_ptr(), C.byref(ran), ws.data_ptr(), ws.numel(), st))
    assert ran.value == steps and nf.cpu().tolist() == [steps] * B
    got = {"mel_outputs": mel, "gate_outputs": gate, "alignments": align}
    for b in (0, 17, 35):
        want = tacotron2_ref.tacotron2_inference(sd, torch.from_numpy(tok[b:b + 1]), masks[:, :, b], 1.0, steps)
        for k in got:
            assert max_abs_diff(got[k][b:b + 1], want[k]) <= TOL, (b, k)


def test_autoregressive_batch_over_32_rows():
    """B = 36 autoregressive rows through the host mirror (two chunks on two lanes), first and last row against batch-1 oracle runs."""
    name = "ar_small_gate"
    case, fx = AR_CASES[name], load_fixture(name)
    steps = 10
    m, (mc, ac, tc) = gpu_model(name, case, max_decoder_steps=steps, gate_threshold=1.0)
    sd = case_state_dict(name)
    B, L = 36, 11
    tok = (gw.hashed_uniform(5, "ar36", B * L) * tc.n_tokens).astype(np.int64).reshape(B, L)
    masks = torch.from_numpy(gw.prenet_keep_masks(steps * B, mc.prenet_dim, seed=3)).reshape(2, steps, B, mc.prenet_dim)
    out = m.inference({"tokens": torch.from_numpy(tok), "prenet_keep_masks": masks})
    assert out["mel_outputs"].shape == (B, ac.n_mels, steps)
    for b in (0, 17, 35):
        want = tacotron2_ref.tacotron2_inference(sd, torch.from_numpy(tok[b:b + 1]), masks[:, :, b], 1.0, steps)
        for k in KEYS:
            assert max_abs_diff(out[k][b:b + 1], want[k]) <= TOL, (b, k)


def test_persistent_attention_equals_per_step_attention():
    """The teacher-forced loop runs its attention as ONE kernel beside the LSTM launches when the shape allows it (default
    layer sizes, B <= 32, L <= 128: attn_persist.hip, 224-workgroup launch layout in skinny.hip) and as a launch per step
    otherwise; both must give the reference's numbers (models/tts/tacotron2.py:243-262, :317-348).  Ragged lengths, fixed
    dropout masks; the hand-off status word must stay clear; two runs must agree bit for bit (fixed reduction orders, no
    data-dependent scheduling in the results), also when the second one replays the hipGraph of the LSTM chain."""
    from genvox_amd import _lib

    mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, log_func="np.log"), TextConfig(n_tokens=40)
    sd = gw.generate_state_dict(mc, ac, tc, seed=0)
    m = Tacotron2(mc, ac, tc)
    m.load_state_dict(sd)
    m = m.to("cuda:0")
    B, L, T, P = 7, 100, 33, mc.prenet_dim
    tl = np.array([100, 97, 64, 50, 33, 2, 1])
    ml = np.array([33, 30, 33, 7, 20, 1, 33])
    inp = gw.synthetic_inputs(B, L, T, tc.n_tokens, ac.n_mels, seed=9, token_lengths=tl, mel_lengths=ml)
    batch = {k: torch.from_numpy(v) for k, v in inp.items()}
    masks = torch.from_numpy(gw.prenet_keep_masks((T + 1) * B, P, seed=2)).reshape(2, T + 1, B, P)
    batch["prenet_keep_masks"] = masks
    lib = _lib.load()
    handle = m._ensure_handle()
    outs = {}
    for tag, on in (("persistent", 1), ("persistent_again", 1), ("persistent_graph", 1), ("per_step", 0)):
        _lib.check(lib.gvx_model_set_persistent_attention(handle, on))
        outs[tag] = {k: v.clone() for k, v in m.forward(batch).items()}
        m.check_status()   # raises if an in-launch wait gave up
    _lib.check(lib.gvx_model_set_persistent_attention(handle, 1))
    for k in KEYS:
        assert torch.equal(outs["persistent"][k], outs["persistent_again"][k]), k
        assert torch.equal(outs["persistent"][k], outs["persistent_graph"][k]), k
        assert max_abs_diff(outs["persistent"][k], outs["per_step"][k]) <= 2e-5, k
    want = tacotron2_ref.tacotron2_forward(sd, batch, masks.reshape(2, -1, P), mask_padding=True)
    for k in KEYS:
        assert max_abs_diff(outs["persistent"][k], want[k]) <= TOL, k


@pytest.mark.parametrize("B,L,T", [(1, 1, 1), (1, 128, 3), (32, 5, 2), (17, 127, 9), (2, 64, 1), (31, 128, 17), (32, 128, 1)])
def test_persistent_attention_edge_shapes(B, L, T):
    """Edge shapes of the persistent-attention loop (single row / token / step, maximum rows and tokens, ragged lengths):
    same numbers as the launch-per-step loop (which test_shapes_against_oracle pins to the oracle), repeatable bit for bit,
    finite, no hand-off time-out."""
    from genvox_amd import _lib

    mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, log_func="np.log"), TextConfig(n_tokens=40)
    m = Tacotron2(mc, ac, tc)
    m.load_state_dict(gw.generate_state_dict(mc, ac, tc, seed=0))
    m = m.to("cuda:0")
    rng = np.random.default_rng(B * 1000 + L * 10 + T)
    tl = np.sort(rng.integers(1, L + 1, B))[::-1].copy()
    tl[0] = L
    ml = rng.integers(1, T + 1, B)
    ml[0] = T
    inp = gw.synthetic_inputs(B, L, T, tc.n_tokens, ac.n_mels, seed=7, token_lengths=tl, mel_lengths=ml)
    batch = {k: torch.from_numpy(v) for k, v in inp.items()}
    batch["prenet_keep_masks"] = torch.from_numpy(gw.prenet_keep_masks((T + 1) * B, mc.prenet_dim, seed=2)).reshape(2, T + 1, B, mc.prenet_dim)
    lib, handle = _lib.load(), m._ensure_handle()
    outs = []
    for on in (1, 1, 0):
        _lib.check(lib.gvx_model_set_persistent_attention(handle, on))
        outs.append({k: v.clone() for k, v in m.forward(batch).items()})
        m.check_status()
    _lib.check(lib.gvx_model_set_persistent_attention(handle, 1))
    for k in KEYS:
        assert torch.isfinite(outs[0][k]).all(), k
        assert torch.equal(outs[0][k], outs[1][k]), k
        assert max_abs_diff(outs[0][k], outs[2][k]) <= 2e-5, k
