"""Host-side mirror of the reference's ``AudioProcessor`` mel->wav path (core/processors.py:55-96) on the MI355X.

One-off host work stays on the host exactly as in the reference: the Slaney mel filterbank and its pseudo-inverse
(utils/audio/base.py:90-137; float64 intermediates, float32 result), the Hann window and the Butterworth
coefficients (scipy.signal.butter, base.py:164-166).  Everything per utterance - dB->amplitude, pseudo-inverse
projection, fast Griffin-Lim, inverse STFT, clip/trim/normalise/low-pass - runs batched on the GPU through the
C ABI (include/genvox_amd.h, vocoder section; rocFFT underneath).  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple, Union

import numpy as np
import scipy.signal
import torch

from . import _lib
from .configs import AudioConfig


def hz_to_mel(hz: float) -> float:
    """Slaney mel scale: linear below 1 kHz, logarithmic above (reference: utils/audio/base.py:90-102)."""
    lin_step, knee_hz = 200.0 / 3.0, 1000.0
    if hz < knee_hz:
        return hz / lin_step
    return knee_hz / lin_step + np.log(hz / knee_hz) * 27.0 / np.log(6.4)


def mel_to_hz(mel: float) -> float:
    """Inverse of :func:`hz_to_mel` (reference: utils/audio/base.py:104-115)."""
    lin_step, knee_hz = 200.0 / 3.0, 1000.0
    knee_mel = knee_hz / lin_step
    if mel < knee_mel:
        return lin_step * mel
    return knee_hz * np.exp((np.log(6.4) / 27.0) * (mel - knee_mel))


def get_mel_filter(fs: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """Triangular mel filterbank with Slaney area normalisation, float32 [n_mels, 1 + n_fft/2]
    (reference: utils/audio/base.py:117-134)."""
    bins = 1 + n_fft // 2
    bin_hz = np.linspace(0, fs / 2, bins)
    edges_hz = np.array([mel_to_hz(m) for m in np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2)])
    widths = np.diff(edges_hz)
    dist = np.subtract.outer(edges_hz, bin_hz)
    bank = np.zeros((n_mels, bins), dtype=np.float32)
    for i in range(n_mels):
        rising, falling = -dist[i] / widths[i], dist[i + 2] / widths[i + 1]
        bank[i] = np.maximum(0, np.minimum(rising, falling))
    bank *= (2.0 / (edges_hz[2:] - edges_hz[:-2]))[:, np.newaxis]
    return bank


def get_inverse_mel_filter(mel_basis: np.ndarray) -> np.ndarray:
    """Moore-Penrose pseudo-inverse [bins, n_mels] (reference: utils/audio/base.py:136-137)."""
    return np.linalg.pinv(mel_basis)


class AudioProcessor:
    TRIM = 500          # samples dropped at both ends (core/processors.py:93)
    LOWPASS_HZ = 6000   # utils/audio/base.py:168-169
    LOWPASS_ORDER = 6

    def __init__(self, config: AudioConfig, device: Union[str, torch.device] = "cuda:0"):
        self.config = config
        c = config
        self.mel_basis = get_mel_filter(fs=c.sampling_rate, n_fft=c.filter_length, n_mels=c.n_mels, fmin=c.mel_fmin, fmax=c.mel_fmax)
        self.inverse_mel_basis = get_inverse_mel_filter(mel_basis=self.mel_basis)
        self.window = scipy.signal.get_window("hann", c.filter_length, fftbins=True).astype(np.float32)
        self._b, self._a = scipy.signal.butter(self.LOWPASS_ORDER, self.LOWPASS_HZ, fs=c.sampling_rate, btype="low", analog=False)
        self.device = torch.device(device)
        self._plan: Optional[int] = None
        self._ws: Optional[torch.Tensor] = None
        self._dev_consts = None

    def __del__(self):
        try:
            if self._plan is not None:
                _lib.load().gvx_gl_plan_destroy(self._plan)
        except Exception:
            pass

    # ------------------------------------------------------------------ plumbing
    def _ensure(self):
        if self.device.type != "cuda":
            raise RuntimeError("genvox_amd.AudioProcessor's mel->wav path runs on an MI355X only (no CPU fallback)")
        lib = _lib.load()
        if self._plan is None:
            h = C.c_void_p()
            _lib.check(lib.gvx_gl_plan_create(self.config.filter_length, self.config.hop_length, C.byref(h)))
            self._plan = h.value
        if self._dev_consts is None:
            self._dev_consts = (torch.from_numpy(self.window).to(self.device),
                                torch.from_numpy(np.ascontiguousarray(self.inverse_mel_basis, dtype=np.float32)).to(self.device))
        return lib

    def _workspace(self, B: int, T: int) -> torch.Tensor:
        need = _lib.load().gvx_gl_workspace_bytes(self._plan, B, T, self.config.n_mels)
        if need == 0:
            raise _lib.GvxError("could not plan the vocoder workspace: " + _lib.load().gvx_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    # ------------------------------------------------------------------ device stages (batched, reference layouts)
    def stft(self, signal: torch.Tensor) -> torch.Tensor:
        """[B, n] float32 -> complex64 [B, bins, T] (reference stft, utils/audio/base.py:58-69, per row)."""
        lib = self._ensure()
        x = signal.to(self.device, torch.float32).contiguous()
        B, n = x.shape
        c = self.config
        T = (n - c.filter_length) // c.hop_length + 1
        out = torch.empty(B, c.filter_length // 2 + 1, T, 2, device=self.device)
        ws = self._workspace(B, T)
        _lib.check(lib.gvx_stft(self._plan, x.data_ptr(), self._dev_consts[0].data_ptr(), B, n, out.data_ptr(), ws.data_ptr(),
                                ws.numel(), self._stream()))
        return torch.view_as_complex(out)

    def istft(self, spec: torch.Tensor) -> torch.Tensor:
        """complex64 [B, bins, T] -> [B, n_fft + (T-1)*hop] (reference istft, utils/audio/base.py:71-88)."""
        lib = self._ensure()
        z = torch.view_as_real(spec.to(self.device, torch.complex64).contiguous()).contiguous()
        B, bins, T, _ = z.shape
        c = self.config
        out = torch.empty(B, c.filter_length + (T - 1) * c.hop_length, device=self.device)
        ws = self._workspace(B, T)
        _lib.check(lib.gvx_istft(self._plan, z.data_ptr(), self._dev_consts[0].data_ptr(), B, T, out.data_ptr(), ws.data_ptr(),
                                 ws.numel(), self._stream()))
        return out

    def mel_to_magnitude(self, mel_db: torch.Tensor) -> torch.Tensor:
        """db_to_amplitude + mel2fft: [B, n_mels, T] -> [B, bins, T]."""
        lib = self._ensure()
        x = mel_db.to(self.device, torch.float32).contiguous()
        B, M, T = x.shape
        c = self.config
        out = torch.empty(B, c.filter_length // 2 + 1, T, device=self.device)
        ws = self._workspace(B, T)
        _lib.check(lib.gvx_mel_to_magnitude(self._plan, x.data_ptr(), self._dev_consts[1].data_ptr(), B, M, T,
                                            0 if c.log_func == "np.log" else 1, float(c.ref_level_db), out.data_ptr(),
                                            ws.data_ptr(), ws.numel(), self._stream()))
        return out

    def griffin_lim(self, mag: torch.Tensor, n_iter: int = 32, momentum: float = 0.99, want_phase: bool = True,
                    want_wav: bool = True) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
        """[B, bins, T] magnitude -> (phase [B, bins, T], waveform [B, n]) (reference griffin_lim + final istft)."""
        lib = self._ensure()
        m = mag.to(self.device, torch.float32).contiguous()
        B, bins, T = m.shape
        c = self.config
        phase = torch.empty_like(m) if want_phase else None
        wav = torch.empty(B, c.filter_length + (T - 1) * c.hop_length, device=self.device) if want_wav else None
        ws = self._workspace(B, T)
        _lib.check(lib.gvx_griffin_lim(self._plan, m.data_ptr(), self._dev_consts[0].data_ptr(), B, T, n_iter, float(momentum),
                                       phase.data_ptr() if want_phase else None, wav.data_ptr() if want_wav else None,
                                       ws.data_ptr(), ws.numel(), self._stream()))
        return phase, wav

    def finalize(self, wav: torch.Tensor) -> torch.Tensor:
        """clip / trim 500 / peak-normalise / Butterworth low-pass -> float64 [B, n - 1000] (core/processors.py:91-95)."""
        lib = self._ensure()
        y = wav.to(self.device, torch.float32).contiguous()
        B, n = y.shape
        out = torch.empty(B, n - 2 * self.TRIM, dtype=torch.float64, device=self.device)
        scratch = torch.empty(B, dtype=torch.int32, device=self.device)
        nb = len(self._b)
        b = (C.c_double * nb)(*[float(v) for v in self._b])
        a = (C.c_double * nb)(*[float(v) for v in self._a])
        _lib.check(lib.gvx_wav_finalize(y.data_ptr(), B, n, self.TRIM, b, a, nb - 1, out.data_ptr(), scratch.data_ptr(), self._stream()))
        return out

    def wav_to_mel(self, signal: torch.Tensor) -> torch.Tensor:
        """[B, n] normalised float32 signals -> mel (dB) [B, n_mels, T] (reference convert_wav2mel chain, per row)."""
        lib = self._ensure()
        x = signal.to(self.device, torch.float32).contiguous()
        B, n = x.shape
        c = self.config
        T = (n - c.filter_length) // c.hop_length + 1
        if len(self._dev_consts) < 3:
            self._dev_consts = self._dev_consts + (torch.from_numpy(np.ascontiguousarray(self.mel_basis)).to(self.device),)
        out = torch.empty(B, c.n_mels, T, device=self.device)
        ws = self._workspace(B, T)
        _lib.check(lib.gvx_wav_to_mel(self._plan, x.data_ptr(), self._dev_consts[0].data_ptr(), self._dev_consts[2].data_ptr(), B, n,
                                      c.n_mels, 0 if c.log_func == "np.log" else 1, float(c.ref_level_db), out.data_ptr(),
                                      ws.data_ptr(), ws.numel(), self._stream()))
        return out

    def convert_wav2mel(self, input_path: str, output_path: str) -> None:
        """Reference signature (core/processors.py:70-79): read a wav file, write the mel (dB) as .npy."""
        import scipy.io.wavfile

        fs, sig = scipy.io.wavfile.read(input_path)
        assert fs == self.config.sampling_rate, f"wav file ({input_path}) sampling rate ({fs}) does not match with config ({self.config.sampling_rate})"
        if self.config.normalize:
            sig = (sig / max(np.abs(np.min(sig)), np.abs(np.max(sig)))).astype(np.float32)   # utils/audio/base.py:20-22
        mel = self.wav_to_mel(torch.from_numpy(np.ascontiguousarray(sig, dtype=np.float32))[None])
        np.save(output_path, mel[0].cpu().numpy())

    # ------------------------------------------------------------------ reference surface
    def convert_mel2wav_batch(self, mels: torch.Tensor, n_iter: int = 32) -> torch.Tensor:
        """[B, n_mels, T] mel (dB) -> float64 waveforms [B, n_fft + (T-1)*hop - 1000] on the device."""
        mag = self.mel_to_magnitude(mels)
        _, wav = self.griffin_lim(mag, n_iter=n_iter, want_phase=False)
        return self.finalize(wav)

    def convert_mel2wav(self, mel: Union[np.ndarray, str, torch.Tensor], n_iter: int = 32) -> Tuple[int, np.ndarray]:
        """Reference signature (core/processors.py:81-96): one mel [n_mels, T] (array or .npy path) -> (fs, float64 signal)."""
        if isinstance(mel, str):
            mel = np.load(mel)
        x = torch.as_tensor(mel, dtype=torch.float32)
        wav = self.convert_mel2wav_batch(x.unsqueeze(0), n_iter=n_iter)
        return self.config.sampling_rate, wav[0].cpu().numpy()
