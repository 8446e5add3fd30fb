"""GPU: BASELINE.json's full-size configurations run under -m gpu (configs[2] autoregressive B=64 x 1000 steps,
configs[3] Postnet + 60-iteration Griffin-Lim at B=256 x 800), the weight re-binding flow of a serving process
(checkpoint A, then checkpoint B on the same model object), and the hipGraph replay path.

Sizes the oracle cannot cover in seconds are checked on sampled rows / prefixes (the decoder is causal: the first
n steps of a 1000-step run are what an n-step run gives) and through size-independent properties (determinism,
softmax rows summing to one, row independence)."""
import numpy as np
import pytest
import torch

from genvox_amd import weights as gw
from genvox_amd.audio import AudioProcessor
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2
from oracle import audio_ref, tacotron2_ref
from tests.golden.cases import AR_CASES, TF_CASES, case_configs
from tests.helpers import TOL, case_state_dict, load_fixture, max_abs_diff, tf_batch, unpack_masks

pytestmark = pytest.mark.gpu
KEYS = ("alignments", "gate_outputs", "mel_outputs", "mel_outputs_postnet")


def full_configs(**over):
    mc = Tacotron2Config(**over)
    return mc, AudioConfig(filter_length=1024, hop_length=256, log_func="np.log"), TextConfig(n_tokens=40)


# --------------------------------------------------------------------------------------------------------------------
def test_rebinding_weights_never_replays_stale_graphs(monkeypatch):
    """forward / inference / encode with weights A, load_state_dict(B), same calls again on the SAME model object and
    workspace: must equal a fresh model that only ever saw B.  GVX_GRAPH_FIRST=1 makes the very first call capture its
    step loops, so the second round would replay graphs holding the old blob's addresses if they survived the re-bind."""
    monkeypatch.setenv("GVX_GRAPH_FIRST", "1")
    name = "tf_full"
    case, fx = TF_CASES[name], load_fixture(name)
    mc, ac, tc = case_configs(case)
    mc.max_decoder_steps, mc.gate_threshold = 20, 1.0
    B, T = case["B"], case["T"]
    sd_a = gw.generate_state_dict(mc, ac, tc, seed=7)
    sd_b = case_state_dict(name)
    masks = unpack_masks(fx["keep_masks_packed"], (2, (T + 1) * B, mc.prenet_dim))
    batch = {**tf_batch(fx), "prenet_keep_masks": masks}
    tok = torch.from_numpy(fx["token_padded"][:2])
    ar_masks = torch.from_numpy(gw.prenet_keep_masks(20 * 2, mc.prenet_dim, seed=3)).reshape(2, 20, 2, mc.prenet_dim)
    ar_in = {"tokens": tok, "prenet_keep_masks": ar_masks}

    m = Tacotron2(mc, ac, tc)
    m.load_state_dict(sd_a)
    m = m.to("cuda:0")
    out_a = m.forward(batch)
    ar_a = m.inference(ar_in)
    blob_a = m._blob.data_ptr()
    m.load_state_dict(sd_b)                      # checkpoint B into the same object (Synthesizer / evaluation flow)
    out_b = m.forward(batch)
    ar_b = m.inference(ar_in)
    mem_b = m.encode(tok, None)
    del blob_a   # the allocator may or may not hand the new blob the old address; both cases must work

    fresh = Tacotron2(mc, ac, tc)
    fresh.load_state_dict(sd_b)
    fresh = fresh.to("cuda:0")
    want, ar_want, mem_want = fresh.forward(batch), fresh.inference(ar_in), fresh.encode(tok, None)
    for k in KEYS:
        assert torch.equal(out_b[k], want[k]), f"teacher-forced {k} differs after re-binding"
        assert torch.equal(ar_b[k], ar_want[k]), f"autoregressive {k} differs after re-binding"
        assert max_abs_diff(out_b[k], fx[k]) <= TOL, k          # and B's outputs are the reference's
    assert torch.equal(mem_b, mem_want)
    assert not torch.equal(out_a["mel_outputs"], out_b["mel_outputs"]) and not torch.equal(ar_a["mel_outputs"], ar_b["mel_outputs"])
    # in-place edits of a parameter are picked up as well (the weights key tracks tensor versions)
    with torch.no_grad():
        m.decoder.gate_layer.linear_layer.bias.add_(1.0)
    shifted = m.forward(batch)
    live = torch.from_numpy(fx["mel_lengths"])
    for b in range(B):
        n = int(live[b])
        assert max_abs_diff(shifted["gate_outputs"][b, :n] - 1.0, want["gate_outputs"][b, :n]) <= 1e-5


@pytest.mark.parametrize("name", ["ar_full_gate", "tf_full_peaky"])
def test_graph_replay_equals_eager(name, monkeypatch):
    """The hipGraph replay of the step loops is bit-identical to the eager launches it captured: first call (eager, the key
    is only remembered), second call (captured + replayed), third (replayed)."""
    monkeypatch.delenv("GVX_GRAPH_FIRST", raising=False)
    if name in AR_CASES:
        case, fx = AR_CASES[name], load_fixture(name)
        steps = int(fx["max_decoder_steps"])
        mc, ac, tc = case_configs(case)
        mc.max_decoder_steps, mc.gate_threshold = steps, float(fx["gate_threshold"])
        m = Tacotron2(mc, ac, tc)
        m.load_state_dict(case_state_dict(name))
        m = m.to("cuda:0")
        masks = unpack_masks(fx["keep_masks_packed"], (2, steps, mc.prenet_dim)).reshape(2, steps, 1, mc.prenet_dim)
        run = lambda: m.inference({"tokens": torch.from_numpy(fx["tokens"]).int(), "prenet_keep_masks": masks})
    else:
        case, fx = TF_CASES[name], load_fixture(name)
        mc, ac, tc = case_configs(case)
        m = Tacotron2(mc, ac, tc)
        m.load_state_dict(case_state_dict(name))
        m = m.to("cuda:0")
        masks = unpack_masks(fx["keep_masks_packed"], (2, (case["T"] + 1) * case["B"], mc.prenet_dim))
        run = lambda: m.forward({**tf_batch(fx), "prenet_keep_masks": masks})
    first, second, third = run(), run(), run()
    for k in KEYS:
        assert torch.equal(first[k], second[k]) and torch.equal(second[k], third[k]), k
        assert max_abs_diff(third[k], fx[k]) <= TOL, k


# --------------------------------------------------------------------------------------------------------------------
def test_config3_autoregressive_b64_1000_steps():
    """BASELINE configs[2]: B = 64, L = 128, gate_threshold = 1.0 -> exactly 1000 steps.  Rows 0 / 31 / 63 against batch-1
    oracle runs over the first 250 steps (Decoder.inference, models/tts/tacotron2.py:390-414), determinism across two
    runs (second one replays the chunk graphs), softmax rows, stop bookkeeping."""
    mc, ac, tc = full_configs(gate_threshold=1.0)
    S, B, L, P = mc.max_decoder_steps, 64, 128, mc.prenet_dim
    assert S == 1000
    sd = gw.generate_state_dict(mc, ac, tc, seed=0)
    m = Tacotron2(mc, ac, tc)
    m.load_state_dict(sd)
    m = m.to("cuda:0")
    tok = torch.from_numpy(gw.synthetic_inputs(B, L, 8, tc.n_tokens, ac.n_mels, seed=3)["token_padded"])
    masks = torch.from_numpy(gw.prenet_keep_masks(S * B, P, seed=11)).reshape(2, S, B, P)
    out = m.inference({"tokens": tok, "prenet_keep_masks": masks})
    out2 = m.inference({"tokens": tok, "prenet_keep_masks": masks})
    assert out["mel_outputs"].shape == (B, ac.n_mels, S) and out["alignments"].shape == (B, S, L)
    assert out["mel_lengths"].cpu().tolist() == [S] * B
    for k in KEYS:
        assert torch.equal(out[k], out2[k]), k
        assert torch.isfinite(out[k]).all(), k
    assert float((out["alignments"].sum(dim=2) - 1).abs().max()) <= 1e-5
    n = 250
    for b in (0, 31, 63):
        want = tacotron2_ref.tacotron2_inference(sd, tok[b:b + 1], masks[:, :n, b], 1.0, n)
        assert max_abs_diff(out["mel_outputs"][b:b + 1, :, :n], want["mel_outputs"]) <= TOL, b
        assert max_abs_diff(out["gate_outputs"][b:b + 1, :n], want["gate_outputs"]) <= TOL, b
        assert max_abs_diff(out["alignments"][b:b + 1, :n], want["alignments"]) <= TOL, b
    # the Postnet over the full 1000 frames of one row (a prefix would see a different right edge)
    mel0 = out["mel_outputs"][0:1].cpu()
    assert max_abs_diff(out["mel_outputs_postnet"][0:1], mel0 + tacotron2_ref.postnet(sd, mel0)) <= TOL


def test_batched_inference_rows_are_postprocessed_at_their_own_length():
    """Rows that stop early keep decoding until the longest row is done; everything the caller sees of row b must still be
    what a batch-1 run of the reference gives: Postnet with zero padding right after the row's last frame, padding values
    (mel 0, gate 1e3, alignment 0) past it."""
    name = "ar_full_gate"
    case, fx = AR_CASES[name], load_fixture(name)
    steps, thr = 30, float(fx["gate_threshold"])
    mc, ac, tc = case_configs(case)
    mc.max_decoder_steps, mc.gate_threshold = steps, thr
    sd = case_state_dict(name)
    m = Tacotron2(mc, ac, tc)
    m.load_state_dict(sd)
    m = m.to("cuda:0")
    lens = [33, 30, 19, 7]
    B, L = len(lens), max(lens)
    tok = (gw.hashed_uniform(78, "postlen", B * L) * tc.n_tokens).astype(np.int64).reshape(B, L)
    tok[0] = fx["tokens"][0]            # the fixture's utterance: its gate fires in the middle of the run
    for b, n in enumerate(lens):
        tok[b, n:] = 0
    masks = torch.from_numpy(gw.prenet_keep_masks(steps * B, mc.prenet_dim, seed=6)).reshape(2, steps, B, mc.prenet_dim)
    # pick a threshold that stops the rows at different steps: gate tracks of free-running oracle rows, then the value
    # (with a safety margin to every track sample) that gives the most distinct first crossings
    tracks = [torch.sigmoid(tacotron2_ref.tacotron2_inference(sd, torch.from_numpy(tok[b:b + 1]), masks[:, :, b], 1.0, steps,
                                                              token_length=n)["gate_outputs"][0]).numpy() for b, n in enumerate(lens)]
    vals = np.sort(np.concatenate(tracks))
    best = None
    for lo, hi in zip(vals[:-1], vals[1:]):
        if hi - lo < 2e-4:   # 1e-4 of room on both sides; the GPU's gate logits are within ~1e-5 of the oracle's
            continue
        c = 0.5 * (lo + hi)
        stops = [int(np.argmax(tr > c)) + 1 if (tr > c).any() else steps for tr in tracks]
        score = (len(set(stops)), -max(stops))
        if min(stops) >= 3 and (best is None or score > best[0]):
            best = (score, float(c))
    assert best is not None and best[0][0] > 1, "no threshold separates the rows' stop steps"
    thr = best[1]
    m.model_config.gate_threshold = thr
    out = m.inference({"tokens": torch.from_numpy(tok), "token_lengths": torch.tensor(lens), "prenet_keep_masks": masks})
    nfs = out["mel_lengths"].cpu().tolist()
    assert len(set(nfs)) > 1, f"rows should stop at different steps for this test to mean anything: {nfs}"
    Tn = out["mel_outputs"].shape[2]
    assert Tn == max(nfs)
    for b, n in enumerate(lens):
        want = tacotron2_ref.tacotron2_inference(sd, torch.from_numpy(tok[b:b + 1]), masks[:, :, b], thr, steps, token_length=n)
        nf = want["mel_outputs"].shape[2]
        assert nfs[b] == nf, (b, nfs[b], nf)
        for k in KEYS:
            got = out[k][b:b + 1, :nf] if k in ("gate_outputs", "alignments") else out[k][b:b + 1, :, :nf]
            assert max_abs_diff(got, want[k]) <= TOL, (b, k)
        assert torch.all(out["mel_outputs"][b, :, nf:] == 0) and torch.all(out["mel_outputs_postnet"][b, :, nf:] == 0)
        assert torch.all(out["gate_outputs"][b, nf:] == 1e3) and torch.all(out["alignments"][b, nf:] == 0)


# --------------------------------------------------------------------------------------------------------------------
def test_config4_postnet_b256x800_rows_match_oracle():
    """BASELINE configs[3], Postnet half: one call on [256, 80, 800] (no per-call batch limit on the GEMM-only path);
    sampled rows against the oracle (Postnet.forward + residual, models/tts/tacotron2.py:194-200, :464)."""
    mc, ac, tc = full_configs()
    sd = gw.generate_state_dict(mc, ac, tc, seed=0)
    m = Tacotron2(mc, ac, tc)
    m.load_state_dict(sd)
    m = m.to("cuda:0")
    B, T = 256, 800
    g = torch.Generator().manual_seed(21)
    mel = torch.randn(B, ac.n_mels, T, generator=g)
    out = m.postnet_residual(mel)
    assert out.shape == mel.shape and torch.isfinite(out).all()
    for b in (0, 63, 64, 200, 255):
        assert max_abs_diff(out[b:b + 1], mel[b:b + 1] + tacotron2_ref.postnet(sd, mel[b:b + 1])) <= TOL, b
    # per-row lengths: each row equals a run on its own truncated sequence, zeros past it
    lens = torch.full((B,), T, dtype=torch.int32)
    lens[3], lens[64], lens[255] = 1, 517, 2
    out_l = m.postnet_residual(mel, lens)
    for b in (3, 64, 255, 0):
        n = int(lens[b])
        want = mel[b:b + 1, :, :n] + tacotron2_ref.postnet(sd, mel[b:b + 1, :, :n])
        assert max_abs_diff(out_l[b:b + 1, :, :n], want) <= TOL, b
        assert torch.all(out_l[b, :, n:] == 0)
    small = m.postnet_residual(mel[5:7, :, :3])          # T = 3 with kernel 5: halo rows outnumber... (T >= halo path edge)
    assert max_abs_diff(small, mel[5:7, :, :3] + tacotron2_ref.postnet(sd, mel[5:7, :, :3])) <= TOL
    one = m.postnet_residual(mel[5:6, :, :1])            # T = 1 < halo: separate halo launch
    assert max_abs_diff(one, mel[5:6, :, :1] + tacotron2_ref.postnet(sd, mel[5:6, :, :1])) <= TOL


def _weighted_phase_diff(a, b, mag):
    w = np.abs(mag) / np.abs(mag).sum()
    return float((np.abs(np.exp(1j * a) - np.exp(1j * b)) * w).sum())


def _spectral_convergence(mag, phase, n_fft, hop):
    spec = (mag * (np.cos(phase) + 1j * np.sin(phase))).astype(np.complex64)
    back = audio_ref.stft(audio_ref.istft(spec, n_fft, hop), n_fft, hop)
    return float(np.linalg.norm(np.abs(back) - np.abs(mag)) / np.linalg.norm(mag))


def test_config4_griffin_lim_800_frames_60_iterations():
    """BASELINE configs[3], vocoder half, on the fused in-LDS FFT kernels (n_fft 1024 / hop 256):
      * the phase estimate after each of the first 4 iterations of an 800-frame utterance against the oracle's
        (griffin_lim, utils/audio/base.py:147-162) - tight, before float32 rounding has been amplified;
      * 60 iterations (the benchmark's count): the result's spectral-convergence metric against the oracle's 60-iteration
        result (phases of energy-free bins are arbitrary by then, the metric is not: it agrees to 1e-5 between a float32
        and a float64 run of the oracle);
      * a B = 256 x 800 batch in one call: every row equals its batch-1 run bit for bit (rows never interact)."""
    ac = AudioConfig(filter_length=1024, hop_length=256, log_func="np.log")
    ap = AudioProcessor(ac)
    n_fft, hop, T = 1024, 256, 800
    g = torch.Generator().manual_seed(4)
    mel = torch.randn(1, 80, T, generator=g) * 1.5 - 4.0           # the benchmark's synthetic mel (dB)
    mag_t = ap.mel_to_magnitude(mel)
    mag = mag_t.cpu().numpy()[0]
    for k, tol in ((1, 2e-4), (2, 4e-4), (3, 1e-3), (4, 2e-3)):
        phase, _ = ap.griffin_lim(mag_t, n_iter=k)
        want = audio_ref.griffin_lim(mag, n_fft, hop, n_iter=k)
        d = _weighted_phase_diff(phase.cpu().numpy()[0], want, mag)
        assert d <= tol, (k, d)
    phase60, wav60 = ap.griffin_lim(mag_t, n_iter=60)
    want60 = audio_ref.griffin_lim(mag, n_fft, hop, n_iter=60)
    got_sc = _spectral_convergence(mag, phase60.cpu().numpy()[0], n_fft, hop)
    want_sc = _spectral_convergence(mag, want60, n_fft, hop)
    sc1 = _spectral_convergence(mag, audio_ref.griffin_lim(mag, n_fft, hop, n_iter=1), n_fft, hop)
    assert abs(got_sc - want_sc) <= 2e-3 and got_sc < 0.9 * sc1, (got_sc, want_sc, sc1)
    spec = (mag * (np.cos(phase60.cpu().numpy()[0]) + 1j * np.sin(phase60.cpu().numpy()[0]))).astype(np.complex64)
    ref_wav = audio_ref.istft(spec, n_fft, hop)
    interior = slice(n_fft, -n_fft)
    assert np.abs(wav60.cpu().numpy()[0][interior] - ref_wav[interior]).max() <= 1e-4 * max(1.0, float(np.abs(ref_wav).max()))
    # full batch of the benchmark in one call
    B = 256
    melB = torch.randn(B, 80, T, generator=g) * 1.5 - 4.0
    melB[17] = mel[0]
    magB = ap.mel_to_magnitude(melB)
    _, wavB = ap.griffin_lim(magB, n_iter=60, want_phase=False)
    assert torch.isfinite(wavB).all()
    assert torch.equal(wavB[17], wav60[0])
    _, w200 = ap.griffin_lim(magB[200:201], n_iter=60, want_phase=False)
    assert torch.equal(wavB[200], w200[0])


# --------------------------------------------------------------------------------------------------------------------
def test_config1_single_utterance_b1_l100_t568():
    """BASELINE configs[0]: one LJSpeech-sized utterance, B = 1, L = 100 tokens, T = 568 frames,
    teacher-forced (models/tts/tacotron2.py:450-481) - the whole utterance against the oracle, all four outputs."""
    mc, ac, tc = full_configs()
    sd = gw.generate_state_dict(mc, ac, tc, seed=0)
    m = Tacotron2(mc, ac, tc)
    m.load_state_dict(sd)
    m = m.to("cuda:0")
    B, L, T, P = 1, 100, 568, mc.prenet_dim
    inp = gw.synthetic_inputs(B, L, T, tc.n_tokens, ac.n_mels, seed=3)
    batch = {k: torch.from_numpy(v) for k, v in inp.items()}
    masks = torch.from_numpy(gw.prenet_keep_masks((T + 1) * B, P, seed=11))
    got = m.forward({**batch, "prenet_keep_masks": masks})
    m.check_status()
    again = m.forward({**batch, "prenet_keep_masks": masks})
    want = tacotron2_ref.tacotron2_forward(sd, batch, masks, mask_padding=True)
    for k in KEYS:
        assert got[k].shape == want[k].shape
        assert torch.equal(got[k], again[k]), k
        d = max_abs_diff(got[k], want[k])
        assert d <= TOL, f"{k}: {d}"
    assert float((got["alignments"].sum(dim=2) - 1).abs().max()) <= 1e-5


@pytest.mark.parametrize("mode", ["one_call_64_rows", "two_lanes_of_32", "two_chunks_of_32_in_turn"])
def test_north_star_batch_64x800(mode, monkeypatch):
    """north_star's target batch (64 x 800 frames, 128 tokens) through Tacotron2.forward, both ways it can run: as ONE call -
    the 64-row loop beside the resident attention kernel (GVX_TF_ROWS64=1: two batch tiles per workgroup, decoder cell cut
    along K and finished one launch later) -, as two 32-row chunks on two HIP streams with a C-ABI handle each
    (GVX_TF_LANES=1: launch-per-step attention) and, by default, as two 32-row chunks one after the other on the caller's
    stream, each beside the resident attention kernel.  Deterministic, softmax rows sum to one, rows independent of the batch they run in
    (rows 0 / 32 / 63 run alone reproduce themselves), and the first 100 steps of rows 0 / 32 / 63 equal what the oracle
    gives on a 100-frame run (the teacher-forced decoder is causal; the Postnet prefix is compared where its receptive
    field - 5 layers x 2 frames - lies inside the prefix)."""
    if mode == "one_call_64_rows":
        monkeypatch.setenv("GVX_TF_ROWS64", "1")   # opt-in (the chunks are faster: gvx_api.hip)
    if mode == "two_lanes_of_32":
        monkeypatch.setenv("GVX_TF_LANES", "1")
    mc, ac, tc = full_configs()
    sd = gw.generate_state_dict(mc, ac, tc, seed=0)
    m = Tacotron2(mc, ac, tc)
    m.load_state_dict(sd)
    m = m.to("cuda:0")
    B, L, T, P = 64, 128, 800, mc.prenet_dim
    tl = np.sort(128 - (np.arange(B) * 5) % 60)[::-1].copy()
    ml = 800 - (np.arange(B) * 29) % 350
    ml[0] = 800
    inp = gw.synthetic_inputs(B, L, T, tc.n_tokens, ac.n_mels, seed=5, token_lengths=tl, mel_lengths=ml)
    batch = {k: torch.from_numpy(v) for k, v in inp.items()}
    masks = torch.from_numpy(gw.prenet_keep_masks((T + 1) * B, P, seed=13)).reshape(2, T + 1, B, P)
    out = m.forward({**batch, "prenet_keep_masks": masks})
    m.check_status()
    assert m.last_forward_lanes == (mode == "two_lanes_of_32")
    assert m.last_forward_sequential32 == (mode == "two_chunks_of_32_in_turn")
    out2 = m.forward({**batch, "prenet_keep_masks": masks})
    for k in KEYS:
        assert torch.isfinite(out[k]).all(), k
        assert torch.equal(out[k], out2[k]), k
    a = out["alignments"]
    assert float((a.sum(dim=2) - 1).abs().max()) <= 1e-5
    for b in range(B):
        assert torch.all(a[b, :, int(tl[b]):] == 0)
        assert torch.all(out["mel_outputs"][b, :, int(ml[b]):] == 0) and torch.all(out["gate_outputs"][b, int(ml[b]):] == 1e3)
    rows = [0, 32, 63]
    sub = {k: v[rows] for k, v in batch.items()}
    alone = m.forward({**sub, "prenet_keep_masks": masks[:, :, rows].contiguous()})
    for k in KEYS:
        assert max_abs_diff(alone[k], out[k][rows]) <= 1e-4, k
    n = 100
    pre = {"token_padded": sub["token_padded"], "token_lengths": sub["token_lengths"], "mel_padded": sub["mel_padded"][:, :, :n].contiguous(),
           "gate_padded": sub["gate_padded"][:, :n].contiguous(), "mel_lengths": torch.full((3,), n, dtype=sub["mel_lengths"].dtype)}
    want = tacotron2_ref.tacotron2_forward(sd, pre, masks[:, :n + 1, rows].reshape(2, -1, P), mask_padding=False)
    assert int(ml[rows].min()) >= n   # the prefix lies inside every sampled row's unmasked part
    for k in ("mel_outputs", "gate_outputs", "alignments"):
        got = out[k][rows][..., :n] if k != "alignments" else out[k][rows][:, :n]
        d = max_abs_diff(got, want[k])
        assert d <= TOL, f"{k}: {d}"
    assert max_abs_diff(out["mel_outputs_postnet"][rows][..., :n - 10], want["mel_outputs_postnet"][..., :n - 10]) <= TOL
