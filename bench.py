#!/usr/bin/env python3
"""Headline benchmark: teacher-forced Tacotron2 forward, mel-frames/s (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]

A "step" is one pass of the hot path (encoder -> Prenet -> T decoder steps -> projection -> Postnet ->
padding mask, Prenet keep masks drawn on device) over one synthetic batch that is already resident in
HBM.  Default workload = BASELINE.json configs[1]: batch 32 x 800 mel frames x 80 bins, 128 tokens, fp32,
random-init default-size weights (seed 0).  With N > 1 (launched by torch.distributed.run, one process per
GPU) rank 0 packs the weights and broadcasts the packed blob over RCCL; every rank then runs its own
independent batch (weak scaling, no data-path collective).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the decoder LSTM step launch):
algorithmic bytes per launch / its average duration, measured with HIP events on the launch stream
around 64 back-to-back replays of a mid-sequence step launch in an instrumented pass right after the timed region.  `cpu_baseline` times the oracle (the CPU
restatement of the reference) on the host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def algorithmic_bytes_lstm_launch(mc, B, L):
    """Bytes one decoder LSTM-step launch must move: both recurrent matrices and biases once, the
    x rows, cell state read+write, new hidden states, attention-query partial slabs, and - since the launch also
    produces the next attention step's location features - the two weight rows read and L*a features written per row."""
    P, E, A, D, a = mc.prenet_dim, mc.encoder_embedding_dim, mc.attention_rnn_dim, mc.decoder_rnn_dim, mc.attention_dim
    weights = 4 * A * (P + E + A) + 4 * D * (A + E + D) + 4 * A + 4 * D
    per_row = (P + E + A) + (A + E + D) + 2 * (A + D) + (A + D)
    slabs = (A // 8) * a
    loc = 2 * L + L * a
    return 4 * (weights + B * (per_row + slabs + loc))


def algorithmic_bytes_decoder_step(mc, B, L):
    """SURVEY.md section 8d figure for one whole decoder step (weights once + per-row traffic)."""
    return 4 * (18103953 + B * (645 * L + 9553))


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="rows per GPU (BASELINE configs[1]: 32)")
    ap.add_argument("--frames", type=int, default=800)
    ap.add_argument("--tokens", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = "RANK" in os.environ   # launched by torch.distributed.run (any N, also N = 1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm

    from genvox_amd import weights as gw
    from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
    from genvox_amd.dist import broadcast_packed_weights
    from genvox_amd.tacotron2 import Tacotron2

    mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, hop_length=256, log_func="np.log"), TextConfig(n_tokens=40)
    B, T, L = args.batch, args.frames, args.tokens
    model = Tacotron2(mc, ac, tc)
    if rank == 0:
        model.load_state_dict(gw.generate_state_dict(mc, ac, tc, seed=0))
    model = model.to(dev)
    broadcast_packed_weights(model, src=0)  # one RCCL broadcast of the packed blob (no-op for a single process)

    inp = gw.synthetic_inputs(B, L, T, tc.n_tokens, ac.n_mels, seed=3 + rank)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    torch.manual_seed(11 + rank)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    log(f"rank {rank}: weights bound, inputs resident; warmup {args.warmup}")
    for _ in range(args.warmup):
        model.forward(batch)
    sync_all()
    log("timed region")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = model.forward(batch)
    sync_all()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)   # the slowest rank defines the job's time
        elapsed = float(t.item())
    assert torch.isfinite(out["mel_outputs_postnet"]).all()
    log(f"timed region done: {elapsed:.3f} s for {args.steps} steps")

    # ---- instrumented pass (not part of the timed region): per-launch and per-stage device times
    roofline, stages = None, None
    if rank == 0:
        model.enable_stage_timing(True)       # pass 1: HIP events between the stages of one eager forward
        model.forward(batch)
        torch.cuda.synchronize()
        st, launches = model.stage_times_ms()
        model.enable_stage_timing(False)
        model.enable_kernel_timing(True)      # pass 2: the step's launches replayed back to back between events
        model.forward(batch)
        torch.cuda.synchronize()
        kt = model.kernel_times_ms()
        model.enable_kernel_timing(False)
        alg = algorithmic_bytes_lstm_launch(mc, B, L)
        achieved = alg / (kt["decoder_lstm_step"] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(REPO, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("batch") == B:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "decoder_lstm_step_kernel", "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "algorithmic_bytes_per_launch": alg,
                    "avg_launch_us": round(kt["decoder_lstm_step"] * 1e3, 2),
                    "attention_launch_us": round(kt["attention_step"] * 1e3, 2),
                    "decoder_step_GBs_survey_8d": round(algorithmic_bytes_decoder_step(mc, B, L) * T / (st["decoder_loop"] * 1e-3) / 1e9, 1)}
        stages = {k: round(v, 3) for k, v in st.items()}

    # ---- CPU baseline (rank 0, N = 1 only): the oracle on the host cores, bounded sample of the same workload
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import tacotron2_ref

        # the GPU box gives one GPU a 16-core CPU share; more threads than that only oversubscribe
        n_thr = max(1, min(int(os.environ.get("GVX_CPU_THREADS", "16")), os.cpu_count() or 1, len(os.sched_getaffinity(0))))
        torch.set_num_threads(n_thr)
        sd = gw.generate_state_dict(mc, ac, tc, seed=0)

        def run_cpu(Tc):
            sub = gw.synthetic_inputs(B, L, Tc, tc.n_tokens, ac.n_mels, seed=3)
            cb = {k: torch.from_numpy(v) for k, v in sub.items()}
            masks = torch.from_numpy(gw.prenet_keep_masks((Tc + 1) * B, mc.prenet_dim))
            c0 = time.perf_counter()
            tacotron2_ref.tacotron2_forward(sd, cb, masks, mask_padding=True)
            return time.perf_counter() - c0

        log(f"cpu baseline: oracle on {n_thr} threads")
        Tc = min(T, 16)
        probe_s = run_cpu(Tc)  # short probe sizes the sample to ~10-20 s of CPU work
        log(f"cpu probe: {Tc} frames in {probe_s:.2f} s")
        Tc = int(max(Tc, min(T, Tc * 15.0 / max(probe_s, 1e-3))))
        cpu_s = run_cpu(Tc)
        log(f"cpu sample: {Tc} frames in {cpu_s:.2f} s")
        cpu = {"value": round(B * Tc / cpu_s, 1), "unit": "mel-frames/s", "cores": n_thr, "kind": "port",
               "sample": f"oracle teacher-forced forward, batch {B} x {Tc} frames x {L} tokens (first {Tc} of {T} frames), "
                         f"{cpu_s:.1f} s, torch {torch.__version__} CPU"}

    if rank == 0:
        frames = world * B * T * args.steps
        line = {
            "metric": "mel-frames/sec (teacher-forced batch)", "value": round(frames / elapsed, 1), "unit": "mel-frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Tacotron2 teacher-forced forward, batch {B} x {T} mel frames x 80 bins, {L} tokens, per GPU "
                                   f"(BASELINE configs[1]); random-init default dims, seed 0",
                       "global_batch": world * B, "frames": T, "tokens": L, "parallelism": f"dp{world} (independent batches, one RCCL weight broadcast)"},
            "roofline": roofline, "cpu_baseline": cpu, "stage_ms": stages,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
