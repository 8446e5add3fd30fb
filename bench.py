#!/usr/bin/env python3
"""Headline benchmark: teacher-forced Tacotron2 forward, mel-frames/s, plus the rest of BASELINE.json's metric
(autoregressive RTF, batch 64 x 800, Postnet MFMA rate, Griffin-Lim) as extra keys of the same JSON line.

    python bench.py [--gpus N --steps K --warmup W]

A "step" is one pass of the hot path (encoder -> Prenet -> T decoder steps -> projection -> Postnet -> padding mask,
Prenet keep masks drawn on device) over one synthetic batch that is already resident in HBM.  Headline workload =
BASELINE.json configs[1]: batch 32 x 800 mel frames x 80 bins, 128 tokens, fp32, random-init default-size weights
(seed 0).

Multi-GPU (weak scaling, B rows per GPU, no data-path collective): one process per GPU.  Either the caller launches
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (RANK / WORLD_SIZE in the environment), or
`python bench.py --gpus N` alone: it then starts that launcher itself as a child process - before this process has
touched the GPU - and relays the child's output (rank 0's JSON line) and exit code.  Rank 0 packs the weights and
broadcasts the packed blob over RCCL; every rank then runs its own independent batch.

Rank 0 prints ONE JSON line.
  roofline      dominant kernel: `decoder_resident_kernel`, ONE launch for all T decoder steps (default layer sizes, B <= 32,
                L <= 256).  Unit = one decoder step for all B rows; `achieved` = SURVEY.md section 8d's algorithmic bytes per step
                x T / the launch's duration, measured live with a HIP event pair on the launch stream around the kernel in an
                instrumented pass right after the timed region (third of three warm passes); `decoder_step_us` next to it is the
                loop's time per step from HIP events recorded INSIDE the timed region.  The kernel keeps the weights on chip, so
                `traffic` is null: HBM counters cannot be collected for it (`rocprofv3 --pmc` serialises kernels, the resident
                decoder and attention kernels wait for each other) - see DESIGN.md section 6.  Shapes without the resident
                kernel report the weight-streaming step launch instead (64 back-to-back replays of a mid-sequence step).
  cpu_baseline  the oracle (CPU restatement of the reference) on the host cores, bounded sample of the same workload.
                Every timed forward is `forward(batch, strict=False)`: the mirror's default (strict=True) synchronises after each
                call to look at the hand-off status; here nothing may synchronise inside the timed region, and the status is
                checked right after it instead (next entry).
  validated     the run checks what it times: device status words after the timed region and after every leg (a hand-off
                time-out or a bad token id fails the run), and the last timed forward must equal an un-timed run with the
                same Prenet-mask seed bit for bit.
  extra         (N = 1) the other BASELINE configurations, each timed like the headline (warm-up, then K runs bracketed by
                synchronize) with its own roofline fraction and CPU-oracle figure: c1_b1x568 (configs[0]), tf_b64x800,
                ar_b64_1000, ar_b1_1000, ar_b1_1000_L190 (a long sentence), postnet_b256x800, train_step_b32x200, gl_60it_b256x800.
                (N > 1) ar_b64_1000_per_rank: every rank decodes 64 rows autoregressively at the same time.
"""
import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
# (kernel, batch rows, tokens) -> HBM bytes per launch measured with PMC counters (profiles/r03_pmc_sq_pa.json: 1.06 x algorithmic)
MEASURED_HBM_BYTES_PER_LAUNCH = {("decoder_lstm_step_pa_kernel", 32, 128): 74524540}
MEASURED_HBM_SOURCE = {("decoder_lstm_step_pa_kernel", 32, 128): "profiles/r03_pmc_sq_pa.json"}
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak (same guide): v_mfma_f32_32x32x2_f32 at 64 FLOP/clk/SIMD


def algorithmic_bytes_lstm_launch(mc, B, L, persistent=False):
    """Bytes one decoder LSTM-step launch must move: both recurrent matrices and biases once, the
    x rows, cell state read+write, new hidden states, attention-query partial slabs, and - when the launch also
    produces the next attention step's location features - the two weight rows read and L*a features written per row.
    persistent: the launch beside the persistent attention kernel (96 slabs, no location features)."""
    P, E, A, D, a = mc.prenet_dim, mc.encoder_embedding_dim, mc.attention_rnn_dim, mc.decoder_rnn_dim, mc.attention_dim
    weights = 4 * A * (P + E + A) + 4 * D * (A + E + D) + 4 * A + 4 * D
    per_row = (P + E + A) + (A + E + D) + 2 * (A + D) + (A + D)
    slabs = ((128 if B > 32 else (96 if L <= 128 else 64)) if persistent else A // 8) * a   # launch layouts of skinny.hip
    loc = 0 if persistent else 2 * L + L * a
    if persistent:   # the Prenet columns of the attention LSTM are applied by one GEMM before the loop: the launch reads their
        weights -= 4 * A * P          # product (4A floats per row) instead of the 4A x P weights and the P inputs per row
        per_row += 4 * A - P
    return 4 * (weights + B * (per_row + slabs + loc))


def flops_lstm_launch(mc, B):
    P, E, A, D = mc.prenet_dim, mc.encoder_embedding_dim, mc.attention_rnn_dim, mc.decoder_rnn_dim
    return 2 * B * (4 * A * (P + E + A) + 4 * D * (A + E + D))


def algorithmic_bytes_decoder_step(mc, B, L):
    """SURVEY.md section 8d figure for one whole decoder step (weights once + per-row traffic)."""
    return 4 * (18103953 + B * (645 * L + 9553))


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_threads():
    # the GPU box gives one GPU a 16-core CPU share; more threads than that only oversubscribe the cgroup
    return max(1, min(int(os.environ.get("GVX_CPU_THREADS", "16")), os.cpu_count() or 1, len(os.sched_getaffinity(0))))


def visible_gpu_count():
    """GPUs this process may use, WITHOUT loading the HIP runtime (the parent of the self-launch must never touch it: on
    this pool a process that initialised HIP must not start another program).  Counts the KFD topology nodes that have SIMDs
    (CPU nodes have none), capped by the first *_VISIBLE_DEVICES list that is set."""
    import glob

    n = 0
    for prop in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(prop) as f:
                for line in f:
                    k, _, v = line.partition(" ")
                    if k == "simd_count" and int(v) > 0:
                        n += 1
        except (OSError, ValueError):
            pass
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
            break
    return n


def self_launch(args):
    """--gpus N without a launcher: start torch.distributed.run as a CHILD (this process has not touched the GPU and never
    will: on this pool a process that initialised HIP must not exec another program) and relay its exit code."""
    have = visible_gpu_count()
    if have < (1 if args.rehearse_one_gpu else args.gpus):
        raise SystemExit(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible on this node")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("launching: " + " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def timed(torch, fn, warm, reps):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="rows per GPU (BASELINE configs[1]: 32)")
    ap.add_argument("--frames", type=int, default=800)
    ap.add_argument("--tokens", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary configurations (extra key)")
    ap.add_argument("--only-extra", default="", help="comma list of extra configurations to run (default: all)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N > 1 ranks share cuda:0 over the gloo backend: exercises the launcher / barrier / broadcast / JSON "
                         "flow on a 1-GPU box; the line is marked \"rehearsal\" and is not a scaling measurement")
    args = ap.parse_args()

    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}")
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = "RANK" in os.environ   # launched by torch.distributed.run (any N, also N = 1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo")
        else:
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
    broadcast_packed_weights(model, src=0)  # one RCCL broadcast of the packed blob (no collective for a single process)
    if args.rehearse_one_gpu and world > 1:
        # several PROCESSES on one GPU: a resident loop needs every CU (two of them would wait for each other's workgroups until
        # their time-outs; the turn-taking of resident loops is per process), so the rehearsal runs the launch-per-step kernels
        from genvox_amd import _lib
        model._ensure_packed()
        _lib.check(_lib.load().gvx_model_set_resident_kernels(model._handle, 0))
        log(f"rank {rank}: rehearsal on a shared GPU: resident kernels off (launch-per-step loops)")

    inp = gw.synthetic_inputs(B, L, T, tc.n_tokens, ac.n_mels, seed=3 + rank)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    torch.manual_seed(11 + rank)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    log(f"rank {rank}: weights bound, inputs resident; warmup {args.warmup}")
    model.enable_stage_timing(True)   # 6 HIP event records per forward, on the launch stream, inside the timed region
    # Reference run for the validation of the timed region's outputs: the Prenet masks of a forward are drawn on the device
    # from a seed taken from torch's CPU generator, so the same manual_seed in front of the LAST timed forward must give
    # the same outputs bit for bit (fixed reduction orders everywhere); the oracle pins this path in tests/, not here.
    VALID_SEED = 1234567 + rank
    torch.manual_seed(VALID_SEED)
    ref = {k: v.clone() for k, v in model.forward(batch, strict=False).items()}
    model.check_status()
    for _ in range(max(args.warmup, 0)):
        model.forward(batch, strict=False)
    sync_all()
    log("timed region")
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == args.steps - 1:
            torch.manual_seed(VALID_SEED)
        out = model.forward(batch, strict=False)
    sync_all()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)   # the slowest rank defines the job's time
        elapsed = float(t.item())
    model.check_status()   # raises if a hand-off of the resident-attention loop timed out (outputs would be NaN) or a token id was bad
    assert torch.isfinite(out["mel_outputs_postnet"]).all()
    validated = all(torch.equal(out[k], ref[k]) for k in ref)
    if not validated:
        raise SystemExit("bench.py: the last timed forward does not reproduce the un-timed reference run bit for bit")
    del ref
    log(f"timed region done: {elapsed:.3f} s for {args.steps} steps; outputs of the last timed forward validated")

    def lstm_roofline(m, b, batch_b, loop_ms):
        """Roofline of the decoder loop's dominant kernel at batch b (instrumented pass, see module docstring)."""
        m.enable_kernel_timing(True)
        for _ in range(3):   # (the third pass is the one read: the first one of a mode pays its one-off set-up)
            m.forward(batch_b, strict=False)
            torch.cuda.synchronize()
        kt = m.kernel_times_ms()
        m.enable_kernel_timing(False)
        persistent = kt["attention_step"] == 0.0   # the attention ran as one kernel beside the loop (B <= 32, default sizes)
        sec = kt["decoder_lstm_step"] * 1e-3       # per step: launch duration, or the resident kernel's duration / T
        tfl = flops_lstm_launch(mc, b) / sec / 1e12
        step_bytes = algorithmic_bytes_decoder_step(mc, b, L)
        common = {"mfma_tflops": round(tfl, 1), "mfma_frac_of_157TF": round(tfl / MFMA_F32_PEAK_TFLOPS, 4),
                  "decoder_step_us": round(loop_ms * 1e3 / T, 2),
                  "decoder_step_GBs_survey_8d": round(step_bytes * T / (loop_ms * 1e-3) / 1e9, 1),
                  # the whole step against the HBM peak (SURVEY 8d bytes per step / measured step time): the figure to compare
                  # across rounds
                  "decoder_step_frac": round(step_bytes * T / (loop_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if m.loop_kind(b, L) == 2:
            # ONE weight-stationary launch for the whole loop: unit = decoder step, T units per launch.  SURVEY 8d's algorithmic
            # bytes (weights once per step + per-row traffic) / the launch's duration: the kernel keeps the weights on chip, so it
            # beats what streaming them could reach (6.3 TB/s copy ceiling) and its real HBM traffic (`traffic`, PMC) is far BELOW the
            # algorithmic bytes; what bounds it now is the fp32 matrix pipe + the step's hand-off chain (mfma_frac_of_157TF)
            alg = step_bytes * T
            gbs = alg / (sec * T) / 1e9
            kname = "decoder_resident_kernel"
            return {"bound": "hbm", "kernel": kname, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "traffic": MEASURED_HBM_BYTES_PER_LAUNCH.get((kname, b, L, T)),
                    "traffic_source": MEASURED_HBM_SOURCE.get((kname, b, L, T)),
                    "algorithmic_bytes_per_launch": alg, "units_per_launch": T, "avg_launch_us": round(sec * T * 1e6, 1),
                    "avg_step_us_in_kernel": round(sec * 1e6, 2),
                    "attention": "attn_persistent_kernel (one launch per decoder loop, 32 CUs)", "attention_launch_us": None, **common}
        alg = algorithmic_bytes_lstm_launch(mc, b, L, persistent)
        gbs = alg / sec / 1e9
        kname = ("decoder_lstm_step_pa64_kernel" if b > 32 else ("decoder_lstm_step_pa_kernel" if L <= 128 else "decoder_lstm_step_pa192_kernel")) \
            if persistent else f"decoder_lstm_step_kernel<{2 if b > 32 else 1}>"
        return {"bound": "hbm", "kernel": kname,
                "achieved": round(gbs, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                # HBM bytes per launch from separate rocprofv3 --pmc passes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE; counter
                # collection serialises kernels, so it cannot run inside this command): the committed measurement of the same
                # kernel at this shape, not something read at bench time
                "traffic": MEASURED_HBM_BYTES_PER_LAUNCH.get((kname, b, L)),
                "traffic_source": MEASURED_HBM_SOURCE.get((kname, b, L)),
                "algorithmic_bytes_per_launch": alg, "avg_launch_us": round(kt["decoder_lstm_step"] * 1e3, 2),
                "attention": "attn_persistent_kernel (one launch per decoder loop, 32 CUs)" if persistent else "attn_step_kernel per step",
                "attention_launch_us": None if persistent else round(kt["attention_step"] * 1e3, 2), **common}

    roofline, stages = None, None
    if rank == 0:
        st, launches = model.stage_times_ms()   # events of the last forward of the timed region
        model.enable_stage_timing(False)
        stages = {k: round(v, 3) for k, v in st.items()}
        roofline = lstm_roofline(model, B, batch, st["decoder_loop"])

    n_thr = cpu_threads()
    cpu_model = host_cpu_model()

    # ---- CPU baseline (rank 0, N = 1 only): the oracle on the host cores, bounded sample of the same workload
    cpu = None
    sd_cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import tacotron2_ref

        torch.set_num_threads(n_thr)
        sd_cpu = gw.generate_state_dict(mc, ac, tc, seed=0)

        def run_cpu(Tc):
            sub = gw.synthetic_inputs(B, L, Tc, tc.n_tokens, ac.n_mels, seed=3)
            cb = {k: torch.from_numpy(v) for k, v in sub.items()}
            masks = torch.from_numpy(gw.prenet_keep_masks((Tc + 1) * B, mc.prenet_dim))
            c0 = time.perf_counter()
            tacotron2_ref.tacotron2_forward(sd_cpu, cb, masks, mask_padding=True)
            return time.perf_counter() - c0

        log(f"cpu baseline: oracle on {n_thr} threads ({cpu_model})")
        Tc = min(T, 16)
        probe_s = run_cpu(Tc)  # short probe sizes the sample to ~10 s of CPU work
        log(f"cpu probe: {Tc} frames in {probe_s:.2f} s")
        Tc = int(max(Tc, min(T, Tc * 10.0 / max(probe_s, 1e-3))))
        cpu_s = run_cpu(Tc)
        log(f"cpu sample: {Tc} frames in {cpu_s:.2f} s")
        cpu = {"value": round(B * Tc / cpu_s, 1), "unit": "mel-frames/s", "cores": n_thr, "kind": "port", "host_cpu": cpu_model,
               "sample": f"oracle teacher-forced forward, batch {B} x {Tc} frames x {L} tokens (first {Tc} of {T} frames), "
                         f"{cpu_s:.1f} s, torch {torch.__version__} CPU"}

    audio_s_per_frame = ac.hop_length / ac.sampling_rate

    def ar_leg(bb, reps=3, tokens=None):
        """Autoregressive decode of bb rows x max_decoder_steps (the gate never fires): seconds per batch."""
        mc.gate_threshold = 1.0   # never fires: exactly max_decoder_steps frames (BASELINE configs[2])
        tok = torch.from_numpy(gw.synthetic_inputs(bb, tokens or L, 8, tc.n_tokens, ac.n_mels, seed=3 + rank)["token_padded"]).to(dev)
        with contextlib.redirect_stdout(sys.stderr):   # "Warning! Reached max decoder steps" must not land next to the JSON line
            dt = timed(torch, lambda: model.inference({"tokens": tok}), 2, reps)
        mc.gate_threshold = 0.5
        model.check_status()
        return dt, tok

    def ar_entry(dt, bb, n_streams, tokens=None):
        S = mc.max_decoder_steps
        kind = model.ar_loop_kind(min(bb, 32), tokens or L)
        execution = {2: "two resident kernels per 32-row chunk for the whole decode (weight-stationary LSTM / projection / Prenet kernel beside the "
                        "attention rows; the kernels end the loop themselves)" + (", chunks one after the other" if bb > 32 else ""),
                     1: "launches per step beside the resident attention kernel, host stop polling",
                     0: "launches per step, host stop polling" + (", 32-row chunks on two streams" if bb > 32 else "")}[kind]
        return {"us_per_step": round(dt / S * 1e6, 2), "ms_per_utterance_batch": round(dt * 1e3, 2),
                "rtf_per_stream": round(dt / (S * audio_s_per_frame), 6), "rtf_aggregate": round(dt / (S * audio_s_per_frame * n_streams), 7),
                "mel_frames_per_s": round(n_streams * S / dt, 1), "includes": "encoder + 1000 decoder steps + Postnet", "execution": execution}

    # ---- N > 1: the second line SURVEY.md section 8d asks of the sharded configuration - autoregressive decode, 64 rows per
    # rank, all ranks at once (barrier on both sides, slowest rank defines the time)
    extra = None
    if world > 1 and not args.no_extra:
        sync_all()
        dt, _ = ar_leg(64)
        sync_all()
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            extra = {"ar_b64_1000_per_rank": {**ar_entry(float(t.item()), 64, 64 * world), "ranks": world,
                                              "note": "64 rows per rank decoded concurrently on every rank; rtf_aggregate / mel_frames_per_s are whole-job figures"}}

    # ---- the other BASELINE configurations (rank 0, N = 1 only)
    if rank == 0 and world == 1 and not args.no_extra:
        want = set(filter(None, args.only_extra.split(","))) or {"c1_b1x568", "tf_b64x800", "ar_b64_1000", "ar_b1_1000", "ar_b1_1000_L190", "postnet_b256x800",
                                                                  "train_step_b32x200", "gl_60it_b256x800"}
        with_cpu = not args.no_cpu_baseline
        extra = {"host_cpu": cpu_model, "cpu_threads": n_thr}
        if with_cpu:
            from oracle import audio_ref, tacotron2_ref
            if sd_cpu is None:
                sd_cpu = gw.generate_state_dict(mc, ac, tc, seed=0)
            torch.set_num_threads(n_thr)
        if "c1_b1x568" in want:   # BASELINE configs[0]: one LJSpeech-sized utterance, teacher-forced
            log("extra: teacher-forced single utterance, 1 x 568 frames, 100 tokens")
            L1, T1 = 100, 568
            c1 = {k: torch.from_numpy(v).to(dev) for k, v in gw.synthetic_inputs(1, L1, T1, tc.n_tokens, ac.n_mels, seed=3).items()}
            dt = timed(torch, lambda: model.forward(c1, strict=False), 2, args.steps)
            model.check_status()
            e = {"ms_per_utterance": round(dt * 1e3, 3), "mel_frames_per_s": round(T1 / dt, 1),
                 "rtf": round(dt / (T1 * audio_s_per_frame), 6), "steps": args.steps,
                 "workload": f"Tacotron2.forward, batch 1 x {T1} frames, {L1} tokens (BASELINE configs[0], an average LJSpeech clip)"}
            if with_cpu:
                cb = {k: v.cpu() for k, v in c1.items()}
                masks = torch.from_numpy(gw.prenet_keep_masks(T1 + 1, mc.prenet_dim))
                c0 = time.perf_counter()
                tacotron2_ref.tacotron2_forward(sd_cpu, cb, masks, mask_padding=True)
                cs = time.perf_counter() - c0
                e["cpu_baseline"] = {"ms_per_utterance": round(cs * 1e3, 1), "mel_frames_per_s": round(T1 / cs, 1), "cores": n_thr, "kind": "port",
                                     "sample": f"oracle teacher-forced forward of the whole utterance (1 x {T1} frames), torch {torch.__version__} CPU"}
            extra["c1_b1x568"] = e
            del c1
        if "tf_b64x800" in want:   # the north_star's target batch
            log("extra: teacher-forced batch 64 x 800")
            b64 = {k: torch.from_numpy(v).to(dev) for k, v in gw.synthetic_inputs(64, L, T, tc.n_tokens, ac.n_mels, seed=3).items()}
            model.enable_stage_timing(True)
            dt = timed(torch, lambda: model.forward(b64, strict=False), 2, args.steps)
            model.check_status()
            lanes = getattr(model, "last_forward_lanes", False)
            st64, _ = model.stage_times_ms(lane=0 if lanes else None)
            model.enable_stage_timing(False)
            seq32 = getattr(model, "last_forward_sequential32", False)
            if lanes:
                execution = "2 chunks x 32 rows, concurrently on 2 HIP streams, one C-ABI call each (launch per attention step)"
            elif seq32:
                execution = ("2 chunks x 32 rows, one after the other on one stream, each beside the resident attention kernel; the Postnet "
                             "once over all 64 rows behind them (stage_ms: the second chunk's call, which ends behind the projection)")
            else:
                execution = ("ONE call: 64-row loop beside the resident attention kernel (64 CUs), 384-workgroup launches with two batch "
                             "tiles each - one pass over the recurrent weights per step for all 64 rows")
            e64 = {"mel_frames_per_s": round(64 * T / dt, 1), "ms_per_step": round(dt * 1e3, 3), "steps": args.steps, "execution": execution,
                   "stage_ms": {k: round(v, 3) for k, v in st64.items()},
                   "decoder_step_us_per_64_rows": round(st64["decoder_loop"] * 1e3 / T * (2 if seq32 else 1), 2)}
            if not lanes and not seq32:
                e64["roofline"] = lstm_roofline(model, 64, b64, st64["decoder_loop"])
            extra["tf_b64x800"] = e64
            del b64
        for key, bb in (("ar_b64_1000", 64), ("ar_b1_1000", 1)):
            if key not in want:
                continue
            log(f"extra: autoregressive batch {bb} x {mc.max_decoder_steps} steps")
            dt, tok = ar_leg(bb)
            e = ar_entry(dt, bb, bb)
            if with_cpu and bb == 1:
                steps_c = 40
                masks = torch.from_numpy(gw.prenet_keep_masks(steps_c, mc.prenet_dim, seed=11)).reshape(2, steps_c, mc.prenet_dim)
                c0 = time.perf_counter()
                tacotron2_ref.tacotron2_inference(sd_cpu, tok.cpu()[:1], masks, 1.0, steps_c)
                cs = time.perf_counter() - c0
                e["cpu_baseline"] = {"us_per_step": round(cs / steps_c * 1e6, 1), "rtf_per_stream": round(cs / (steps_c * audio_s_per_frame), 4),
                                     "cores": n_thr, "kind": "port",
                                     "sample": f"oracle Tacotron2.inference, batch 1 (the reference's only autoregressive mode), {steps_c} steps incl. encoder + Postnet"}
            extra[key] = e
        if "ar_b1_1000_L190" in want:   # a long sentence (the reference's autoregressive mode is batch 1; LJSpeech rows reach ~190 characters)
            log("extra: autoregressive batch 1 x 1000 steps, 190 tokens")
            dt, _ = ar_leg(1, tokens=190)
            extra["ar_b1_1000_L190"] = ar_entry(dt, 1, 1, tokens=190)
        if "postnet_b256x800" in want:
            log("extra: Postnet batch 256 x 800 (one call)")
            mel = torch.randn(256, ac.n_mels, T, device=dev)
            dt = timed(torch, lambda: model.postnet_residual(mel), 1, 3)
            model.check_status()
            flops = 8.68e6 * 256 * T   # 8.68 MFLOP per frame
            e = {"ms": round(dt * 1e3, 3), "tflops": round(flops / dt / 1e12, 1), "mfma_frac_of_157TF": round(flops / dt / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                 "frames_per_s": round(256 * T / dt, 1), "note": "whole call incl. the two layout transposes and the residual"}
            if with_cpu:
                rows = mel[:4].cpu()
                c0 = time.perf_counter()
                tacotron2_ref.postnet(sd_cpu, rows)
                cs = time.perf_counter() - c0
                e["cpu_baseline"] = {"frames_per_s": round(4 * T / cs, 1), "cores": n_thr, "kind": "port", "sample": f"oracle Postnet on 4 x {T} frames"}
            extra["postnet_b256x800"] = e
            del mel
        if "train_step_b32x200" in want:   # the reference's train_step (forward, loss, backward, clip, Adam)
            log("extra: training step, batch 32 x 200 frames")
            Tt = 200
            tb = {k: torch.from_numpy(v).to(dev) for k, v in gw.synthetic_inputs(B, L, Tt, tc.n_tokens, ac.n_mels, seed=3).items()}
            sd_keep = {k: v.detach().clone() for k, v in model.state_dict().items()}   # the legs after this one use the original weights
            opt = model.get_optimizer()
            dt = timed(torch, lambda: model.train_step(tb, model.get_criterion(), opt), 2, 3)   # (the second step builds the device re-packing map)
            model.check_status()
            e = {"s_per_step": round(dt, 4), "mel_frames_per_s": round(B * Tt / dt, 1), "loss_after": round(model.loss_items["loss"], 4),
                 "grad_norm": round(model.grad_norm_val, 4),
                 "note": "explicit backward (no autograd): both recurrences as single C-ABI calls (3 launches per decoder step, 1 per encoder step), weights re-packed on the device; pinned to the reference's own train_step"}
            # forward 48.5 MFLOP per mel frame, the backward twice that (data and weight gradients)
            e["tflops"] = round(3 * 48.5e6 * B * Tt / dt / 1e12, 2)
            e["mfma_frac_of_157TF"] = round(e["tflops"] / 157.3, 4)
            if with_cpu:
                from oracle import train_ref
                Bc, Lc, Tc2 = 8, 64, 96
                cb = {k: torch.from_numpy(v) for k, v in gw.synthetic_inputs(Bc, Lc, Tc2, tc.n_tokens, ac.n_mels, seed=3).items()}
                gen = torch.Generator().manual_seed(1)
                bern = lambda shape, p: (torch.rand(shape, generator=gen) >= p).to(torch.uint8)
                E_, A_, D_, P_, C_, n_ = mc.encoder_embedding_dim, mc.attention_rnn_dim, mc.decoder_rnn_dim, mc.prenet_dim, mc.postnet_embedding_dim, mc.postnet_n_convolutions
                cm = {"encoder": [bern((Bc, E_, Lc), 0.5) for _ in range(mc.encoder_n_convolutions)], "prenet": bern((2, Tc2 + 1, Bc, P_), 0.5),
                      "attention_rnn": bern((Tc2, Bc, A_), 0.1), "decoder_rnn": bern((Tc2, Bc, D_), 0.1),
                      "postnet": [bern((Bc, C_ if i < n_ - 1 else ac.n_mels, Tc2), 0.5) for i in range(n_)]}
                c0 = time.perf_counter()
                co, ct = train_ref.train_forward(sd_cpu, cb, cm, mc)
                train_ref.train_backward(sd_cpu, cb, cm, mc, co, ct)
                cs = time.perf_counter() - c0
                e["cpu_baseline"] = {"mel_frames_per_s": round(Bc * Tc2 / cs, 1), "cores": n_thr, "kind": "port",
                                     "sample": f"oracle train_forward + train_backward (explicit formulas, torch CPU), batch {Bc} x {Tc2} frames x {Lc} tokens, {cs:.1f} s"}
            # the same step at the C2 size (32 x 800 frames): what one batch of the headline configuration costs to train on
            tb8 = {k: torch.from_numpy(v).to(dev) for k, v in gw.synthetic_inputs(B, L, T, tc.n_tokens, ac.n_mels, seed=3).items()}
            dt8 = timed(torch, lambda: model.train_step(tb8, model.get_criterion(), opt), 1, 2)
            model.check_status()
            e["c2_size_b32x800"] = {"s_per_step": round(dt8, 4), "mel_frames_per_s": round(B * T / dt8, 1)}
            del tb8
            extra["train_step_b32x200"] = e
            model.load_state_dict(sd_keep)
            model.eval()
            del tb
        if "gl_60it_b256x800" in want:
            from genvox_amd.audio import AudioProcessor

            log("extra: Griffin-Lim 60 iterations, batch 256 x 800")
            apx = AudioProcessor(ac, device=dev)
            mel = torch.randn(256, ac.n_mels, T, device=dev) * 1.5 - 4.0
            mag = apx.mel_to_magnitude(mel)
            dt = timed(torch, lambda: apx.griffin_lim(mag, n_iter=60, want_phase=False), 1, 3)
            gbs = 20516 * 256 * T * 60 / dt / 1e9   # 20 516 B per frame-iteration (minimal fused traffic)
            e = {"ms": round(dt * 1e3, 2), "frames_per_s": round(256 * T / dt, 1), "utterances_per_s": round(256 / dt, 1),
                 "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                              "algorithmic_bytes_per_frame_iteration": 20516,
                              # what gl_iteration_kernel itself moves (the spectrum stays on chip): 12 500 B per frame-iteration
                              "moved_bytes_per_frame_iteration": 12500, "moved_gbs": round(12500 * 256 * T * 60 / dt / 1e9, 1)}}
            if with_cpu:
                import numpy as np

                Tc, itc = 200, 6
                magc = np.abs(np.random.default_rng(0).standard_normal((513, Tc))).astype(np.float32)
                c0 = time.perf_counter()
                audio_ref.griffin_lim(magc, 1024, 256, n_iter=itc)
                cs = time.perf_counter() - c0
                e["cpu_baseline"] = {"frames_per_s_at_60it": round(Tc * itc / cs / 60, 1), "cores": 1, "kind": "port",
                                     "sample": f"oracle griffin_lim (NumPy, one utterance like the reference), {Tc} frames x {itc} iterations, scaled to 60"}
            extra["gl_60it_b256x800"] = e

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
            "validated": "status words clear after the timed region; last timed forward bit-equal to an un-timed run with the same mask seed",
            "extra": extra,
        }
        if args.rehearse_one_gpu:
            line["rehearsal"] = f"{world} ranks sharing ONE GPU over gloo: flow check only, not a scaling measurement"
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
