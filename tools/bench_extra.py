#!/usr/bin/env python3
"""Secondary measurements for DESIGN.md (BASELINE.json configs 2-4 beyond the headline line of bench.py):
teacher-forced at B=64, autoregressive RTF (B=64, 1000 steps), Postnet-only MFMA rate, Griffin-Lim throughput."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from genvox_amd import weights as gw
from genvox_amd.audio import AudioProcessor
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2


def timed(fn, warm=1, reps=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    which = set(sys.argv[1:]) or {"tf64", "ar", "postnet", "gl"}
    mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, hop_length=256, log_func="np.log"), TextConfig(n_tokens=40)
    res = {}
    if which & {"tf64", "ar", "ar1", "postnet"}:
        model = Tacotron2(mc, ac, tc)
        model.load_state_dict(gw.generate_state_dict(mc, ac, tc, seed=0))
        model = model.to("cuda:0")
    if "tf64" in which:
        B, L, T = 64, 128, 800
        batch = {k: torch.from_numpy(v).cuda() for k, v in gw.synthetic_inputs(B, L, T, 40, 80, seed=3).items()}
        dt = timed(lambda: model.forward(batch), warm=2, reps=5)
        res["teacher_forced_b64"] = {"ms": round(dt * 1e3, 2), "mel_frames_per_s": round(B * T / dt)}
    if "ar" in which:
        B, L = 64, 128
        mc.gate_threshold = 1.0  # never fires: exactly max_decoder_steps = 1000 frames (BASELINE config 3)
        tok = torch.from_numpy(gw.synthetic_inputs(B, L, 8, 40, 80, seed=3)["token_padded"]).cuda()
        dt = timed(lambda: model.inference({"tokens": tok}), warm=1, reps=2)
        audio_s = mc.max_decoder_steps * ac.hop_length / ac.sampling_rate
        res["autoregressive_b64_1000steps"] = {"ms": round(dt * 1e3, 1), "us_per_step": round(dt / mc.max_decoder_steps * 1e6, 1),
                                               "rtf_per_utterance_stream": round(dt / audio_s, 5),
                                               "rtf_aggregate": round(dt / (audio_s * B), 6),
                                               "mel_frames_per_s": round(B * mc.max_decoder_steps / dt)}
        mc.gate_threshold = 0.5
    if "ar1" in which:   # the reference's own autoregressive shape: one utterance
        mc.gate_threshold = 1.0
        tok = torch.from_numpy(gw.synthetic_inputs(1, 128, 8, 40, 80, seed=3)["token_padded"]).cuda()
        dt = timed(lambda: model.inference({"tokens": tok}), warm=1, reps=2)
        audio_s = mc.max_decoder_steps * ac.hop_length / ac.sampling_rate
        res["autoregressive_b1_1000steps"] = {"ms": round(dt * 1e3, 1), "us_per_step": round(dt / mc.max_decoder_steps * 1e6, 1),
                                              "rtf": round(dt / audio_s, 5)}
    if "postnet" in which:
        B, T = 256, 800
        mel = torch.randn(64, 80, T, device="cuda")
        dt = timed(lambda: model.postnet_residual(mel), warm=1, reps=3) * (B / 64)
        flops = 8.68e6 * B * T
        res["postnet_b256x800"] = {"ms": round(dt * 1e3, 2), "tflops": round(flops / dt / 1e12, 1), "frac_of_157TF_fp32_mfma": round(flops / dt / 157.3e12, 3),
                                   "note": "4 calls of 64 rows (C-ABI batch limit)"}
    if "gl" in which:
        ap = AudioProcessor(ac)
        B, T, it = 256, 800, 60
        mel = torch.randn(B, 80, T, device="cuda") * 1.5 - 4.0
        mag = ap.mel_to_magnitude(mel)
        dt = timed(lambda: ap.griffin_lim(mag, n_iter=it, want_phase=False), warm=1, reps=2)
        res["griffin_lim_b256x800_60it"] = {"ms": round(dt * 1e3, 1), "frames_per_s": round(B * T / dt),
                                            "GBs_vs_min_fused_traffic": round(20516 * B * T * it / dt / 1e9),
                                            "utterances_per_s": round(B / dt, 1)}
        dt2 = timed(lambda: ap.convert_mel2wav_batch(mel, n_iter=32), warm=1, reps=2)
        res["convert_mel2wav_b256x800_32it"] = {"ms": round(dt2 * 1e3, 1), "utterances_per_s": round(B / dt2, 1)}
        sig = torch.rand(B, 1024 + (T - 1) * 256, device="cuda") * 2 - 1
        dt3 = timed(lambda: ap.wav_to_mel(sig), warm=1, reps=3)
        os.environ["GVX_GL_ROCFFT"] = "1"
        dt4 = timed(lambda: ap.wav_to_mel(sig), warm=1, reps=3)
        del os.environ["GVX_GL_ROCFFT"]
        res["wav_to_mel_b256x800"] = {"ms": round(dt3 * 1e3, 2), "frames_per_s": round(B * T / dt3), "ms_rocfft_pipeline": round(dt4 * 1e3, 2)}
    if "cpu" in which:
        # CPU baselines for configs 3 and 4 (the oracle = CPU restatement of the reference, on this box's host cores):
        # bounded samples, reported beside the GPU figures above; bench.py carries the one for config 2.
        import numpy as np

        from oracle import audio_ref, tacotron2_ref

        n_thr = max(1, min(16, os.cpu_count() or 1, len(os.sched_getaffinity(0))))
        torch.set_num_threads(n_thr)
        sd = gw.generate_state_dict(mc, ac, tc, seed=0)
        steps, L = 40, 128
        tok = torch.from_numpy(gw.synthetic_inputs(1, L, 8, 40, 80, seed=3)["token_padded"])
        masks = torch.from_numpy(gw.prenet_keep_masks(steps, mc.prenet_dim, seed=11)).reshape(2, steps, mc.prenet_dim)
        t0 = time.perf_counter()
        tacotron2_ref.tacotron2_inference(sd, tok, masks, gate_threshold=1.0, max_decoder_steps=steps)
        dt = time.perf_counter() - t0
        res["cpu_autoregressive_b1"] = {"us_per_step": round(dt / steps * 1e6, 1), "rtf_per_utterance_stream": round(dt / steps / (ac.hop_length / ac.sampling_rate), 4),
                                        "cores": n_thr, "sample": f"oracle Tacotron2.inference, batch 1 (the reference's only autoregressive mode), {steps} steps incl. encoder"}
        T, it = 200, 8
        mag = np.abs(np.random.default_rng(0).standard_normal((513, T))).astype(np.float32)
        t0 = time.perf_counter()
        audio_ref.griffin_lim(mag, 1024, 256, n_iter=it)
        dt = time.perf_counter() - t0
        per_frame_it = dt / (T * it)
        res["cpu_griffin_lim"] = {"frames_per_s_at_60it": round(1.0 / (per_frame_it * 60)), "cores": 1,
                                  "sample": f"oracle griffin_lim (numpy, per-frame loops like the reference), {T} frames x {it} iterations, scaled to 60"}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
