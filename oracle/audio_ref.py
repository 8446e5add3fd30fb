"""ORACLE — test infrastructure, never shipped on the product path.

NumPy restatement of the mel->wav half of the reference's audio library: Hann STFT /
inverse STFT without centring, Slaney mel filterbank and its pseudo-inverse, fast
Griffin-Lim, and the ``convert_mel2wav`` tail.  Vectorised over frames (the reference
loops per frame in Python) but the arithmetic, the dtypes at every step and the
order of the overlap-add accumulation are kept so results are bit-comparable with
the reference on the same NumPy build.  Pinned by ``tests/golden/audio_*.npz``, which
``tests/golden/make_fixtures.py`` produced from the imported reference.

Each function cites the reference lines it restates (utils/audio/base.py unless noted).
"""
from __future__ import annotations

import numpy as np
import scipy.signal


def hann_window(n_fft: int) -> np.ndarray:
    """Periodic Hann, float32 (base.py:64, :74)."""
    return scipy.signal.get_window("hann", n_fft, fftbins=True).astype(np.float32)


def frame_count(n_samples: int, n_fft: int, hop: int) -> int:
    """base.py:9 (no padding, no centring)."""
    return (n_samples - n_fft) // hop + 1


def stft(y: np.ndarray, n_fft: int, hop: int) -> np.ndarray:
    """base.py:58-69. float32 signal -> complex64 [1 + n_fft/2, n_frames]."""
    assert y.dtype == np.float32 and y.ndim == 1
    n = frame_count(y.shape[0], n_fft, hop)
    frames = np.lib.stride_tricks.as_strided(y, shape=(n, n_fft), strides=(y.strides[0] * hop, y.strides[0]))
    win = hann_window(n_fft)
    out = np.zeros((1 + n_fft // 2, n), dtype=np.complex64)
    out[:, :] = np.fft.rfft(win[None, :] * frames, n_fft, axis=1).T
    return out


def istft(spec: np.ndarray, n_fft: int, hop: int) -> np.ndarray:
    """base.py:71-88. Overlap-add of window * irfft(column), accumulated frame by frame
    into a float32 buffer (same rounding order as the reference), divided by the
    summed squared window where that exceeds float32 tiny."""
    n = spec.shape[1]
    y = np.zeros(n_fft + (n - 1) * hop, dtype=np.float32)
    win = hann_window(n_fft)
    seg = win[None, :] * np.fft.irfft(spec.T, n_fft, axis=1)  # float32 or float64, as the input dictates
    for t in range(n):
        y[t * hop: t * hop + n_fft] += seg[t]
    wsq = win ** 2
    norm = np.zeros(y.shape, dtype=np.float32)
    for t in range(n):
        norm[t * hop: t * hop + n_fft] += wsq
    nz = norm > np.finfo(np.float32).tiny
    y[nz] /= norm[nz]
    return y


def hz_to_mel(hz: float) -> float:
    """Slaney scale (base.py:90-102)."""
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return min_log_mel + np.log(hz / min_log_hz) / logstep if hz >= min_log_hz else hz / f_sp


def mel_to_hz(mel: float) -> float:
    """Slaney scale (base.py:104-115)."""
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return min_log_hz * np.exp(logstep * (mel - min_log_mel)) if mel >= min_log_mel else f_sp * mel


def mel_filter(fs: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """base.py:117-134: triangular filters, Slaney area normalisation, float32 [n_mels, 1+n_fft/2]."""
    n_bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0, fs / 2, n_bins)
    mel_f = np.array([mel_to_hz(m) for m in np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2)])
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, n_bins), dtype=np.float32)
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w


def inverse_mel_filter(mel_basis: np.ndarray) -> np.ndarray:
    """base.py:136-137."""
    return np.linalg.pinv(mel_basis)


def db_to_amplitude(db: np.ndarray, log_func: str = "np.log10", ref: float = 1.0, amin: float = 1e-5) -> np.ndarray:
    """base.py:38-52 with power=False, scale=1 (how core/processors.py:87 calls it).
    Unlike the reference this does not modify its argument in place."""
    x = np.array(db, copy=True)
    x /= 1.0  # in-place like the reference, so a float32 input stays float32 (NEP 50 would promote x + float64)
    if log_func == "np.log":
        x += np.log(np.maximum(amin, ref))
        return np.exp(x)
    x += np.log10(np.maximum(amin, ref))
    return np.power(10, x)


def amplitude_to_db(mag: np.ndarray, log_func: str = "np.log10", ref: float = 1.0, amin: float = 1e-5) -> np.ndarray:
    """base.py:24-36 with power=False, scale=1 (how core/processors.py:77 calls it)."""
    lf = np.log if log_func == "np.log" else np.log10
    db = lf(np.maximum(amin, np.abs(mag)))
    db -= lf(np.maximum(amin, ref))
    return db * 1.0


def wav_to_mel(signal: np.ndarray, mel_basis: np.ndarray, n_fft: int, hop: int, log_func: str, ref: float) -> np.ndarray:
    """AudioProcessor.convert_wav2mel without the file I/O (core/processors.py:70-79): stft -> |.| -> fft2mel (base.py:139-141) -> dB."""
    spec = stft(signal, n_fft, hop)
    return amplitude_to_db(np.matmul(mel_basis, np.abs(spec)), log_func=log_func, ref=ref)


def griffin_lim(mag: np.ndarray, n_fft: int, hop: int, momentum: float = 0.99, n_iter: int = 32,
                return_complex: bool = False) -> np.ndarray:
    """Fast Griffin-Lim (base.py:147-162): magnitude [bins, frames] -> phase [bins, frames].
    The first update promotes to complex128 because the 'previous' buffer starts as
    float64 zeros (base.py:153), exactly as in the reference."""
    angles = np.ones(mag.shape, dtype=np.complex64)
    eps = np.finfo(np.complex64).tiny
    prev = np.zeros(angles.shape)
    angles *= mag
    for _ in range(n_iter):
        rebuilt = stft(istft(angles, n_fft, hop), n_fft, hop)
        angles = rebuilt - (momentum / (1 + momentum)) * prev
        angles /= np.abs(angles) + eps
        angles *= mag
        prev = rebuilt
    return angles if return_complex else np.angle(angles)


def convert_mel2wav(mel_db: np.ndarray, mel_basis_inv: np.ndarray, fs: int, n_fft: int, hop: int,
                    log_func: str, ref: float, n_iter: int = 32):
    """AudioProcessor.convert_mel2wav (core/processors.py:81-96)."""
    mel_amp = db_to_amplitude(mel_db, log_func=log_func, ref=ref)
    mag = np.matmul(mel_basis_inv, mel_amp)  # mel2fft, base.py:143-145
    ang = griffin_lim(mag, n_fft, hop, n_iter=n_iter)
    spec = mag * (np.cos(ang) + 1j * np.sin(ang))  # base.py:54-56
    sig = istft(spec, n_fft, hop)
    sig[(sig > 1) | (sig < -1)] = 0
    sig = sig[500:-500]
    sig = (sig / max(np.abs(np.min(sig)), np.abs(np.max(sig)))).astype(np.float32)  # base.py:20-22
    b, a = scipy.signal.butter(6, 6000, fs=fs, btype="low", analog=False)  # base.py:164-169
    return fs, scipy.signal.lfilter(b, a, sig)
