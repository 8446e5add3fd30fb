"""GPU parity of the vocoder path (STFT / iSTFT / mel->magnitude / Griffin-Lim / convert_mel2wav) against the
reference-generated fixture tests/golden/audio.npz and the NumPy oracle.

Griffin-Lim is an iterative phase retrieval: in bins that carry no energy the phase is numerically arbitrary (the
oracle run in float64 instead of float32 flips such bins by up to pi after 8 iterations), so phases are compared on
the unit circle weighted by magnitude, with tolerances taken from that float32-vs-float64 self-drift
(8e-5 / 1.4e-4 / 1.9e-2 after 1 / 2 / 32 iterations), and the spectral-convergence metric of the result is compared
as well (it agrees to 1e-5 between float32 and float64)."""
import numpy as np
import pytest
import torch

from genvox_amd.audio import AudioProcessor
from genvox_amd.configs import AudioConfig
from oracle import audio_ref
from tests.golden.cases import AUDIO_CASE
from tests.helpers import load_fixture, max_abs_diff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ap():
    c = AUDIO_CASE
    return AudioProcessor(AudioConfig(sampling_rate=c["fs"], filter_length=c["n_fft"], hop_length=c["hop"], n_mels=c["n_mels"],
                                      mel_fmin=c["fmin"], mel_fmax=c["fmax"], log_func=c["log_func"], ref_level_db=c["ref"]))


def weighted_phase_diff(phase_a, phase_b, mag):
    w = np.abs(mag) / np.abs(mag).sum()
    return float((np.abs(np.exp(1j * phase_a) - np.exp(1j * phase_b)) * w).sum())


def istft_error(got, want, n_fft, hop):
    """max |diff| weighted by the summed squared window: the first / last samples of the reference's un-centred
    iSTFT are divided by a window sum of ~1e-10, which amplifies any FFT rounding difference by 1/w^2 (the
    reference trims 500 samples at both ends before use, core/processors.py:93)."""
    w2 = audio_ref.hann_window(n_fft) ** 2
    wss = np.zeros(want.shape[0], np.float32)
    for t in range((want.shape[0] - n_fft) // hop + 1):
        wss[t * hop: t * hop + n_fft] += w2
    return float((np.abs(got - want) * np.minimum(1.0, wss)).max())


def spectral_convergence(mag, phase, n_fft, hop):
    spec = (mag * (np.cos(phase) + 1j * np.sin(phase))).astype(np.complex64)
    back = audio_ref.stft(audio_ref.istft(spec, n_fft, hop), n_fft, hop)
    return float(np.linalg.norm(np.abs(back) - np.abs(mag)) / np.linalg.norm(mag))


def test_host_constants_match_reference(ap):
    fx = load_fixture("audio")
    assert np.array_equal(ap.mel_basis, fx["mel_basis"])
    assert max_abs_diff(ap.inverse_mel_basis, fx["inverse_mel_basis"]) <= 1e-6
    assert np.array_equal(ap.window, audio_ref.hann_window(AUDIO_CASE["n_fft"]))


def test_stft_istft_match_reference(ap):
    fx = load_fixture("audio")
    sig = torch.from_numpy(fx["signal"])
    spec = ap.stft(torch.stack([sig, 0.5 * sig.flip(0)])).cpu().numpy()
    ref = fx["stft_real"] + 1j * fx["stft_imag"]
    scale = np.abs(ref).max()
    assert spec.shape == (2,) + ref.shape
    assert np.abs(spec[0] - ref).max() <= 2e-6 * scale
    ref2 = audio_ref.stft(0.5 * fx["signal"][::-1].copy(), AUDIO_CASE["n_fft"], AUDIO_CASE["hop"])
    assert np.abs(spec[1] - ref2).max() <= 2e-6 * np.abs(ref2).max()
    back = ap.istft(torch.from_numpy(np.stack([ref, ref2]).astype(np.complex64))).cpu().numpy()
    assert istft_error(back[0], fx["istft"], AUDIO_CASE["n_fft"], AUDIO_CASE["hop"]) <= 3e-6
    assert istft_error(back[1], audio_ref.istft(ref2, AUDIO_CASE["n_fft"], AUDIO_CASE["hop"]), AUDIO_CASE["n_fft"], AUDIO_CASE["hop"]) <= 3e-6
    interior = slice(AUDIO_CASE["n_fft"], -AUDIO_CASE["n_fft"])
    assert np.abs(back[0][interior] - fx["istft"][interior]).max() <= 3e-6
    # non-multiple-of-4 signal length takes the unaligned framing path
    odd = fx["signal"][:5001].copy()
    s_odd = ap.stft(torch.from_numpy(odd)[None]).cpu().numpy()[0]
    r_odd = audio_ref.stft(odd, AUDIO_CASE["n_fft"], AUDIO_CASE["hop"])
    assert s_odd.shape == r_odd.shape and np.abs(s_odd - r_odd).max() <= 2e-6 * np.abs(r_odd).max()


def test_mel_to_magnitude_matches_reference(ap):
    fx = load_fixture("audio")
    mag = ap.mel_to_magnitude(torch.from_numpy(fx["mel_db"])[None]).cpu().numpy()[0]
    assert mag.shape == fx["mag"].shape
    assert np.abs(mag - fx["mag"]).max() <= 1e-4 * np.abs(fx["mag"]).max()
    assert (fx["mag"] < 0).any() and np.array_equal(mag < -1e-6, fx["mag"] < -1e-6)  # pinv magnitudes can be negative


@pytest.mark.parametrize("n_iter,tol", [(1, 1e-3), (2, 1e-3), (32, 0.1)])
def test_griffin_lim_matches_reference(ap, n_iter, tol):
    fx = load_fixture("audio")
    c = AUDIO_CASE
    mag = fx["mag"]
    phase, wav = ap.griffin_lim(torch.from_numpy(mag)[None], n_iter=n_iter)
    phase = phase.cpu().numpy()[0]
    assert weighted_phase_diff(phase, fx[f"gl_phase_{n_iter}"], mag) <= tol
    got, want = spectral_convergence(mag, phase, c["n_fft"], c["hop"]), spectral_convergence(mag, fx[f"gl_phase_{n_iter}"], c["n_fft"], c["hop"])
    assert abs(got - want) <= 2e-3, (got, want)
    # the synthesis output is istft(mag * exp(i phase))
    spec = (mag * (np.cos(phase) + 1j * np.sin(phase))).astype(np.complex64)
    assert istft_error(wav.cpu().numpy()[0], audio_ref.istft(spec, c["n_fft"], c["hop"]), c["n_fft"], c["hop"]) <= 1e-4


def test_convert_mel2wav_matches_reference(ap):
    fx = load_fixture("audio")
    fs, wav = ap.convert_mel2wav(fx["mel_db"].copy())
    assert fs == int(fx["fs"]) and wav.dtype == np.float64 and wav.shape == fx["wav"].shape
    rel = np.linalg.norm(wav - fx["wav"]) / np.linalg.norm(fx["wav"])
    assert rel <= 0.2, rel  # 32 chaotic iterations; see module docstring
    assert abs(np.abs(wav).max() - np.abs(fx["wav"]).max()) <= 0.05
    # the deterministic tail alone (clip / trim / normalise / Butterworth) must match tightly given the same input signal
    c = AUDIO_CASE
    ph = fx["gl_phase_32"]
    raw = audio_ref.istft((fx["mag"] * (np.cos(ph) + 1j * np.sin(ph))).astype(np.complex64), c["n_fft"], c["hop"])
    tail = ap.finalize(torch.from_numpy(raw)[None]).cpu().numpy()[0]
    assert np.abs(tail - fx["wav"]).max() <= 1e-5


def test_batched_rows_equal_single_rows(ap):
    fx = load_fixture("audio")
    mel = torch.from_numpy(fx["mel_db"])
    batch = torch.stack([mel, mel - 0.5, mel.flip(1)])
    together = ap.convert_mel2wav_batch(batch, n_iter=8)
    for i in range(3):
        alone = ap.convert_mel2wav_batch(batch[i:i + 1], n_iter=8)
        assert torch.equal(together[i], alone[0])


def test_wav_to_mel_matches_reference(ap):
    """feature extraction on the device (stft -> |.| -> mel filterbank -> dB)."""
    fx = load_fixture("audio")
    sig = torch.from_numpy(fx["signal"])
    mel = ap.wav_to_mel(torch.stack([sig, 0.25 * sig])).cpu().numpy()
    assert mel.shape == (2,) + fx["mel_db"].shape
    assert np.abs(mel[0] - fx["mel_db"]).max() <= 1e-4          # dB values are O(1..10); fp32 log of an fp32 GEMM
    want = audio_ref.wav_to_mel(0.25 * fx["signal"], fx["mel_basis"], AUDIO_CASE["n_fft"], AUDIO_CASE["hop"], AUDIO_CASE["log_func"], AUDIO_CASE["ref"])
    assert np.abs(mel[1] - want).max() <= 1e-4


def test_finalize_chunked_filter_matches_lfilter(ap):
    """clip / trim / peak-normalise / Butterworth low-pass (core/processors.py:91-95, utils/audio/base.py:164-169) on
    signals long enough for many filter chunks: the overlap-discard kernel equals scipy's sequential lfilter to float64
    rounding, for ragged batch sizes, and a row's result does not depend on its batch."""
    import scipy.signal

    rng = np.random.default_rng(5)
    B, n = 5, 30000 + 2 * ap.TRIM + 37            # ~30 chunks of 1024, ragged tail
    wav = (rng.standard_normal((B, n)) * 0.4).astype(np.float32)
    wav[1, 1000:1010] = 3.0                        # spurious samples > 1 are zeroed (core/processors.py:92)
    got = ap.finalize(torch.from_numpy(wav)).cpu().numpy()
    for b in range(B):
        y = wav[b].copy()
        y[np.abs(y) > 1] = 0
        y = y[ap.TRIM:-ap.TRIM]
        y = y / np.abs(y).max()
        want = scipy.signal.lfilter(ap._b, ap._a, y.astype(np.float64))
        assert np.abs(got[b] - want).max() <= 1e-13
    alone = ap.finalize(torch.from_numpy(wav[3:4])).cpu().numpy()
    assert np.array_equal(alone[0], got[3])


def test_finalize_slow_filter_falls_back_to_sequential(ap):
    """A filter whose impulse response does not die out within the warm-up cap runs on the sequential kernel."""
    import ctypes as C

    import scipy.signal

    from genvox_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(6)
    B, n, trim = 3, 5000, 10
    wav = (rng.standard_normal((B, n)) * 0.3).astype(np.float32)
    b = np.array([1.0, 0.0]); a = np.array([1.0, -0.9999])     # pole at 0.9999: |M^W| < 1e-18 needs W ~ 4e5
    y = torch.from_numpy(wav).cuda()
    out = torch.empty(B, n - 2 * trim, dtype=torch.float64, device="cuda")
    scratch = torch.empty(B, dtype=torch.int32, device="cuda")
    _lib.check(lib.gvx_wav_finalize(y.data_ptr(), B, n, trim, (C.c_double * 2)(*b), (C.c_double * 2)(*a), 1, out.data_ptr(),
                                    scratch.data_ptr(), torch.cuda.current_stream().cuda_stream))
    got = out.cpu().numpy()
    for r in range(B):
        x = wav[r].copy(); x[np.abs(x) > 1] = 0; x = x[trim:-trim]; x = x / np.abs(x).max()
        want = scipy.signal.lfilter(b, a, x.astype(np.float64))
        assert np.abs(got[r] - want).max() <= 1e-9 * np.abs(want).max()


@pytest.mark.parametrize("B,T", [(1, 1), (3, 13), (2, 29), (5, 100)])
def test_fused_griffin_lim_matches_rocfft_pipeline(ap, B, T, monkeypatch):
    """The fused in-LDS FFT iteration (n_fft 1024 / hop 256) against the rocFFT pipeline it replaces, on the same
    magnitudes: frame counts around the 13-block / 16-frame workgroup shape, first-iteration and momentum paths.
    Both are fp32; they differ only in the FFTs' internal rounding, which Griffin-Lim amplifies slowly."""
    rng = np.random.default_rng(B * 100 + T)
    mag = np.abs(rng.standard_normal((B, 513, T))).astype(np.float32) * 3.0
    mag[:, :, T // 2] *= 1e-3                      # a quiet frame
    for n_iter, tol in ((0, 1e-5), (1, 2e-5), (4, 2e-4)):
        monkeypatch.delenv("GVX_GL_ROCFFT", raising=False)
        ph_f, wav_f = ap.griffin_lim(torch.from_numpy(mag), n_iter=n_iter)
        monkeypatch.setenv("GVX_GL_ROCFFT", "1")
        ph_r, wav_r = ap.griffin_lim(torch.from_numpy(mag), n_iter=n_iter)
        wf, wr = wav_f.cpu().numpy(), wav_r.cpu().numpy()
        for b in range(B):   # edge samples are divided by a ~1e-10 window sum: compare weighted (see istft_error)
            assert istft_error(wf[b], wr[b], 1024, 256) <= tol * max(float(np.abs(wr[b]).max()), 1.0) * 10, (n_iter, "wav", b)
        assert weighted_phase_diff(ph_f.cpu().numpy(), ph_r.cpu().numpy(), mag) <= tol * 50, (n_iter, "phase")


def test_other_fft_size_uses_rocfft_pipeline_and_matches_oracle():
    """n_fft 512 / hop 128 has no fused kernels: the rocFFT pipeline serves it; two Griffin-Lim iterations and the
    final iSTFT against the oracle (utils/audio/base.py:147-162, :71-88)."""
    n_fft, hop, T = 512, 128, 37
    ap2 = AudioProcessor(AudioConfig(sampling_rate=22050, filter_length=n_fft, hop_length=hop, n_mels=80, mel_fmin=0.0,
                                     mel_fmax=8000.0, log_func="np.log", ref_level_db=1.0))
    rng = np.random.default_rng(9)
    mag = np.abs(rng.standard_normal((n_fft // 2 + 1, T))).astype(np.float32)
    phase, wav = ap2.griffin_lim(torch.from_numpy(mag)[None], n_iter=2)
    phase = phase.cpu().numpy()[0]
    want = audio_ref.griffin_lim(mag, n_fft, hop, n_iter=2)
    assert weighted_phase_diff(phase, want, mag) <= 1e-3
    spec = (mag * (np.cos(phase) + 1j * np.sin(phase))).astype(np.complex64)
    assert istft_error(wav.cpu().numpy()[0], audio_ref.istft(spec, n_fft, hop), n_fft, hop) <= 1e-4
