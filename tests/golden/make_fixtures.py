#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference itself.

Runs ONLY in the build container, where the reference checkout is mounted read-only at
/root/reference; nothing on the GPU box or in the product path reads it.  The reference
is imported unchanged.  Four modules it imports at package level but never uses on the
model/audio path are absent from this image (yt_dlp, inflect, wandb, g2p_en) and are
registered as empty stand-ins before the import.

What is written (inputs + expected outputs only, no reference code):
  tf_*.npz     teacher-forced Tacotron2.forward: batch, bit-packed Prenet keep masks,
               the four outputs and intermediate taps
  ar_*.npz     autoregressive Tacotron2.inference (batch 1): tokens, per-step masks, outputs
  audio.npz    mel basis / pseudo-inverse, STFT / iSTFT, Griffin-Lim phases, convert_mel2wav
  collate.npz  TextMelCollateFn on a ragged list (ties included): inputs and the collated batch
  text.json    TextProcessor.tokenize / generate_token_map / tokens_to_indices on digit-free sentences
  ref_exp/     config.yaml and checkpoint_3.pt WRITTEN BY the reference (reduced dims, the reference's own init),
               expected.npz = the reference's inference outputs from those files

Weights are NOT stored: they are regenerated from genvox_amd.weights (seed, dims) and
loaded INTO the reference with load_state_dict.

Usage:  python tests/golden/make_fixtures.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"

for _name in ("yt_dlp", "inflect", "wandb", "g2p_en"):
    sys.modules.setdefault(_name, types.ModuleType(_name))
sys.modules["inflect"].engine = lambda *a, **k: None
sys.modules["g2p_en"].G2p = lambda *a, **k: None
sys.path.insert(0, REFERENCE)
sys.path.insert(1, REPO)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import configs as ref_configs  # noqa: E402  (reference)
from configs.models import Tacotron2Config as RefTacotron2Config  # noqa: E402  (reference)
from models.tts.tacotron2 import Tacotron2 as RefTacotron2, Tacotron2Loss as RefTacotron2Loss  # noqa: E402  (reference)
from core.processors import AudioProcessor as RefAudioProcessor  # noqa: E402  (reference)
from utils.audio import base as ref_audio  # noqa: E402  (reference)

from genvox_amd import weights as gw  # noqa: E402
from tests.golden.cases import AR_CASES, AUDIO_CASE, TF_CASES, TRAIN_CASE, case_configs  # noqa: E402

torch.set_num_threads(8)


def build_reference_model(case):
    mc, ac, tc = case_configs(case)
    ref_tc = ref_configs.TextConfig()
    ref_tc.n_tokens = tc.n_tokens
    ref_ac = ref_configs.AudioConfig(**{k: v for k, v in ac.to_dict().items()})
    ref_mc = RefTacotron2Config(**{k: v for k, v in mc.to_dict().items()})
    model = RefTacotron2(ref_mc, ref_ac, ref_tc).eval()
    sd = gw.generate_state_dict(mc, ac, tc, seed=case["weight_seed"], peaky_attention=case.get("peaky", False))
    model.load_state_dict(sd, strict=True)
    return model, (mc, ac, tc)


def draw_keep_masks(seed, shape, n_pairs=1):
    """The reference consumes torch's CPU RNG only in the two Prenet dropouts (eval mode),
    so after manual_seed(s) its masks equal these draws, in this order."""
    torch.manual_seed(seed)
    out = []
    for _ in range(n_pairs):
        pair = [(F.dropout(torch.ones(*shape), 0.5, True) > 0).to(torch.uint8) for _ in range(2)]
        out.append(torch.stack(pair))
    return out


def make_tf_case(name, case):
    model, (mc, ac, tc) = build_reference_model(case)
    B, L, T = case["B"], case["L"], case["T"]
    inp = gw.synthetic_inputs(B, L, T, tc.n_tokens, ac.n_mels, seed=case["input_seed"],
                              token_lengths=case["token_lengths"], mel_lengths=case["mel_lengths"])
    batch = {k: torch.from_numpy(v) for k, v in inp.items()}
    masks = draw_keep_masks(case["mask_seed"], (T + 1, B, mc.prenet_dim))[0]  # [2, T+1, B, P]

    taps = {"ctx": []}
    hooks = [
        model.encoder.register_forward_hook(lambda m, i, o: taps.__setitem__("encoder_outputs", o.detach().clone())),
        model.decoder.prenet.register_forward_hook(lambda m, i, o: taps.__setitem__("prenet_outputs", o.detach().clone())),
        model.decoder.attention_layer.memory_layer.register_forward_hook(
            lambda m, i, o: taps.__setitem__("processed_memory", o.detach().clone())),
        model.decoder.attention_layer.register_forward_hook(lambda m, i, o: taps["ctx"].append(o[0].detach().clone())),
    ]
    torch.manual_seed(case["mask_seed"])
    with torch.no_grad():
        out = model.forward({k: v.clone() for k, v in batch.items()})
    for h in hooks:
        h.remove()
    # the masks above really are what the reference drew: its prenet output must be reproduced exactly
    x = torch.cat((torch.zeros(1, B, ac.n_mels), batch["mel_padded"].permute(2, 0, 1)), 0)
    w0 = model.decoder.prenet.layers[0].linear_layer.weight
    w1 = model.decoder.prenet.layers[1].linear_layer.weight
    chk = torch.relu(torch.relu(x @ w0.t()) * masks[0] * 2 @ w1.t()) * masks[1] * 2
    assert torch.equal(chk, taps["prenet_outputs"]), "mask extraction does not reproduce the reference's Prenet"

    # the reference's criterion on its own outputs (eval_step, models/tts/tacotron2.py:524-529, :598-615)
    with torch.no_grad():
        loss = RefTacotron2Loss({k: v.clone() for k, v in batch.items()}, out)
    np.savez_compressed(
        os.path.join(HERE, f"{name}.npz"),
        **{k: v for k, v in inp.items()},
        loss=np.float32(loss["loss"].item()), mel_loss=np.float32(loss["mel_loss"].item()), gate_loss=np.float32(loss["gate_loss"].item()),
        keep_masks_packed=np.packbits(masks.numpy().reshape(2, -1), axis=1),
        mel_outputs=out["mel_outputs"].numpy(), mel_outputs_postnet=out["mel_outputs_postnet"].numpy(),
        gate_outputs=out["gate_outputs"].numpy(), alignments=out["alignments"].numpy(),
        encoder_outputs=taps["encoder_outputs"].numpy(), processed_memory=taps["processed_memory"].numpy(),
        prenet_outputs=taps["prenet_outputs"].numpy(), attention_contexts=torch.stack(taps["ctx"], 1).numpy(),
    )
    a = out["alignments"]
    print(f"{name}: mel_post |max| {out['mel_outputs_postnet'].abs().max():.3f}  max alignment {a.max():.3f}")


def make_train_case(case):
    """Reference in .train() mode: forward, Tacotron2Loss, loss.backward() (what train_step does before clipping,
    models/tts/tacotron2.py:515-520).  Every dropout the reference draws goes through F.dropout; it is replaced for the
    duration of the call by an equivalent that draws an explicit Bernoulli keep mask, so that the masks are inputs of the
    fixture (call order: encoder convolutions, Prenet x 2, then per decoder step attention-LSTM / decoder-LSTM output,
    Postnet convolutions) and the gradient that reaches each dropout output can be read back."""
    model, (mc, ac, tc) = build_reference_model(case)
    model.train()
    B, L, T = case["B"], case["L"], case["T"]
    inp = gw.synthetic_inputs(B, L, T, tc.n_tokens, ac.n_mels, seed=case["input_seed"],
                              token_lengths=case["token_lengths"], mel_lengths=case["mel_lengths"])
    batch = {k: torch.from_numpy(v) for k, v in inp.items()}
    drawn = []   # (keep mask uint8, output tensor with retained grad)
    real_dropout = F.dropout

    def recording_dropout(x, p=0.5, training=True, inplace=False):
        if not training:
            return x
        keep = torch.bernoulli(torch.full_like(x, 1.0 - p))
        out = x * keep / (1.0 - p)
        if out.requires_grad:
            out.retain_grad()
        drawn.append((keep.to(torch.uint8), out))
        return out

    optimizer = model.get_optimizer()["optimizer"]   # torch.optim.Adam(lr, weight_decay of the config): models/tts/tacotron2.py:506-513
    optimizer.zero_grad()
    taps = {}
    hook = model.decoder.register_forward_hook(lambda m, i, o: taps.__setitem__("dec_mel", o[0].detach().clone()))
    torch.manual_seed(case["mask_seed"])
    F.dropout = recording_dropout
    try:
        out = model.forward({k: v.clone() for k, v in batch.items()})
        hook.remove()
        dec_mel = out["mel_outputs"]
        dec_mel.retain_grad()
        loss = RefTacotron2Loss({k: v.clone() for k, v in batch.items()}, out)
        loss["loss"].backward()
    finally:
        F.dropout = real_dropout
    ne, npn = mc.encoder_n_convolutions, mc.postnet_n_convolutions
    assert len(drawn) == ne + 2 + 2 * T + npn, len(drawn)
    enc, pre, steps, post = drawn[:ne], drawn[ne:ne + 2], drawn[ne + 2:ne + 2 + 2 * T], drawn[ne + 2 + 2 * T:]
    pack = lambda ms: np.packbits(np.stack([m.numpy().reshape(-1) for m in ms]), axis=1)
    grads = {("grad." + k): v.grad.numpy().copy() for k, v in model.named_parameters()}   # before clipping
    # the rest of train_step (models/tts/tacotron2.py:521-522): clip_grad_norm_, Adam.step
    grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), model.model_config.grad_clip_thresh).item()
    optimizer.step()
    after = {("after." + k): v.detach().numpy().copy() for k, v in model.named_parameters()}
    bn_stats = {("state." + k): v.detach().numpy() for k, v in model.state_dict().items() if "running_" in k}
    np.savez_compressed(
        os.path.join(HERE, "train_small.npz"),
        **{k: v for k, v in inp.items()},
        enc_keep_packed=pack([m for m, _ in enc]),               # [n_enc][B, E, L]
        prenet_keep_packed=pack([m for m, _ in pre]),            # [2][T+1, B, P]
        att_keep_packed=pack([m for m, _ in steps[0::2]]),       # [T][B, A]
        dec_keep_packed=pack([m for m, _ in steps[1::2]]),       # [T][B, D]
        post_keep_packed=pack([m.reshape(-1)[: m.numel()] for m, _ in post[:-1]]) if npn > 1 else np.zeros((0, 0), np.uint8),
        post_last_keep_packed=pack([post[-1][0]]),               # [B, n_mels, T] (the last layer has n_mels channels)
        mel_outputs=out["mel_outputs"].detach().numpy(), mel_outputs_postnet=out["mel_outputs_postnet"].detach().numpy(),
        gate_outputs=out["gate_outputs"].detach().numpy(), alignments=out["alignments"].detach().numpy(),
        loss=np.float32(loss["loss"].item()), mel_loss=np.float32(loss["mel_loss"].item()), gate_loss=np.float32(loss["gate_loss"].item()),
        decoder_mel_unmasked=taps["dec_mel"].numpy(),            # the Postnet's input: the decoder's mel output BEFORE the padding mask (:463, :466-473)
        grad_mel_outputs=dec_mel.grad.numpy(),                   # d loss / d (decoder mel output), all paths
        grad_gate_outputs=np.zeros(0, np.float32),
        grad_enc_conv_out=enc[-1][1].grad.numpy(),               # d loss / d (output of the encoder's convolution stack) [B, E, L]
        grad_post_out=post[-1][1].grad.numpy(),                  # d loss / d (Postnet output before the residual add)
        grad_norm=np.float32(grad_norm), **grads, **bn_stats, **after,
    )
    print(f"train_small: loss {loss['loss'].item():.5f}  dropouts {len(drawn)}  |grad mel| max {dec_mel.grad.abs().max():.4e}")


def make_ar_case(name, case):
    model, (mc, ac, tc) = build_reference_model(case)
    L, steps = case["L"], case["max_decoder_steps"]
    tokens = (gw.hashed_uniform(case["input_seed"], "ar_tokens", L) * tc.n_tokens).astype(np.int64)[None, :]
    pairs = draw_keep_masks(case["mask_seed"], (1, mc.prenet_dim), n_pairs=steps)
    masks = torch.stack(pairs, dim=1).squeeze(2)  # [2, steps, P]

    def run(threshold):
        model.decoder.gate_threshold = threshold
        model.decoder.max_decoder_steps = steps
        torch.manual_seed(case["mask_seed"])
        return model.inference({"tokens": torch.from_numpy(tokens).int()})

    out = run(1.0)  # never fires: exactly `steps` frames
    assert out["mel_outputs"].shape[2] == steps
    threshold = 1.0
    if case["gate_fires"]:
        # choose a threshold the sigmoid(gate) track first crosses at a step well inside the run
        sig = torch.sigmoid(out["gate_outputs"][0]).numpy()
        lo, hi = steps // 4, (3 * steps) // 4
        k = lo + int(np.argmax(sig[lo:hi]))
        prior = float(sig[:k].max())
        if sig[k] > prior:
            threshold = float((prior + sig[k]) / 2)
        else:  # pick the last record-setting step instead
            rec = [i for i in range(1, steps) if sig[i] > sig[:i].max()]
            k = rec[-1]
            threshold = float((sig[:k].max() + sig[k]) / 2)
        out = run(threshold)
        assert out["mel_outputs"].shape[2] == k + 1, (out["mel_outputs"].shape, k)
    np.savez_compressed(
        os.path.join(HERE, f"{name}.npz"),
        tokens=tokens, keep_masks_packed=np.packbits(masks.numpy().reshape(2, -1), axis=1),
        gate_threshold=np.float64(threshold), max_decoder_steps=np.int64(steps),
        mel_outputs=out["mel_outputs"].numpy(), mel_outputs_postnet=out["mel_outputs_postnet"].numpy(),
        gate_outputs=out["gate_outputs"].numpy(), alignments=out["alignments"].numpy(),
    )
    print(f"{name}: frames {out['mel_outputs'].shape[2]} threshold {threshold:.6f}")


def make_audio_case():
    c = AUDIO_CASE
    ref_ac = ref_configs.AudioConfig(sampling_rate=c["fs"], filter_length=c["n_fft"], hop_length=c["hop"],
                                     n_mels=c["n_mels"], mel_fmin=c["fmin"], mel_fmax=c["fmax"],
                                     log_func=c["log_func"], ref_level_db=c["ref"])
    ap = RefAudioProcessor(ref_ac)
    n_fft, hop, T = c["n_fft"], c["hop"], c["frames"]
    n = n_fft + (T - 1) * hop
    t = np.arange(n) / c["fs"]
    u = gw.hashed_uniform(c["seed"], "audio_noise", n)
    sig = (0.45 * np.sin(2 * np.pi * (220 + 900 * t) * t) + 0.25 * np.sin(2 * np.pi * 1330 * t)
           + 0.05 * (2 * u - 1)).astype(np.float32)
    spec = ref_audio.stft(sig, n_fft, hop)
    back = ref_audio.istft(spec, n_fft, hop)
    mel_amp = ref_audio.fft2mel(np.abs(spec), ap.mel_basis)
    mel_db = ref_audio.amplitude_to_db(mel_amp, log_func=c["log_func"], ref=c["ref"], power=False, scale=1)
    mag = ref_audio.mel2fft(ref_audio.db_to_amplitude(mel_db.copy(), log_func=c["log_func"], ref=c["ref"],
                                                      power=False, scale=1), ap.inverse_mel_basis)
    gl = {f"gl_phase_{k}": ref_audio.griffin_lim(mag, n_fft, hop, n_iter=k) for k in (1, 2, 32)}
    fs, wav = ap.convert_mel2wav(mel_db.copy())
    np.savez_compressed(
        os.path.join(HERE, "audio.npz"), signal=sig, stft_real=spec.real, stft_imag=spec.imag, istft=back,
        mel_basis=ap.mel_basis, inverse_mel_basis=ap.inverse_mel_basis, mel_db=mel_db.astype(np.float32),
        mag=mag.astype(np.float32), wav=wav, fs=np.int64(fs), **gl)
    print(f"audio: frames {T} wav {wav.shape} {wav.dtype} neg-mag-frac {(mag < 0).mean():.4f}")


HOST_SENTENCES = [   # digit-free: number spelling goes through the third-party `inflect`, which this image lacks
    "Dr. Smith met Mrs. Jones & Co. at St. Mary's (the old one) - twice!",
    "  Tabs\tand   runs of   spaces,\nnew lines; UPPER case?  ",
    "Lt. Col. Brown [ret.] said: \"Ft. Worth ~ isn't far\" <quote> #tag @home ten% _under_ a+b=c | x/y",
    "the quick brown fox jumps over the lazy dog",
    "Gen. Hon. Rev. Sgt. Capt. Esq. Ltd. Jr. Maj. Drs. Mr.",
]


def make_host_fixtures():
    """Host rows of SURVEY.md section 8: collate layout (a16), text front-end (f1), config + checkpoint files (f2) -
    all produced by the reference's own code."""
    import json

    from core.processors import TextProcessor as RefTextProcessor  # noqa: E402  (reference)
    from models.tts import TextMelCollateFn as RefCollate  # noqa: E402  (reference)

    # ---- a16: TextMelCollateFn on a ragged list with equal token counts (tie order) and equal frame counts
    rng = np.random.default_rng(31)
    tok_lens = [9, 21, 15, 21, 3, 15, 15, 1]
    mel_lens = [14, 30, 22, 30, 5, 9, 41, 1]
    items = [{"tokens": torch.from_numpy(rng.integers(0, 30, size=n).astype(np.int32)),
              "features": torch.from_numpy(rng.standard_normal((12, t)).astype(np.float32))} for n, t in zip(tok_lens, mel_lens)]
    out = RefCollate()(items)
    np.savez_compressed(
        os.path.join(HERE, "collate.npz"), tok_lens=np.array(tok_lens), mel_lens=np.array(mel_lens),
        tokens_cat=np.concatenate([x["tokens"].numpy() for x in items]),
        feats_cat=np.concatenate([x["features"].numpy() for x in items], axis=1),
        **{k: v.numpy() for k, v in out.items()})
    print("collate:", {k: tuple(v.shape) for k, v in out.items()}, "order of token_lengths", out["token_lengths"].tolist())

    # ---- f1: TextProcessor with base_cleaners on digit-free sentences
    ref_tc = ref_configs.TextConfig(language="english", cleaners=["base_cleaners"])
    tp = RefTextProcessor(ref_tc)
    toks = [tp.tokenize(t) for t in HOST_SENTENCES]
    token_map = tp.generate_token_map()
    idx = [tp.tokens_to_indices(t) for t in toks]
    with open(os.path.join(HERE, "text.json"), "w") as f:
        json.dump({"sentences": HOST_SENTENCES, "cleaned": ["".join(t) for t in toks], "token_map": token_map, "indices": idx,
                   "n_tokens": ref_tc.n_tokens}, f, indent=1, ensure_ascii=False)
    print("text: n_tokens", ref_tc.n_tokens, "|", "".join(toks[0]))

    # ---- f2: exp/config.yaml written by BaseConfig.write_configs_to_file (as the trainer does, core/trainer/__init__.py:73-82)
    # and checkpoint_<it>.pt written as CheckpointManager.save_model does (get_checkpoint_statedicts + iteration + torch.save),
    # for a reduced-dims model with the REFERENCE's own (seeded) initialisation; then the reference's outputs from those files.
    from tests.golden.cases import SMALL
    exp = os.path.join(HERE, "ref_exp")
    os.makedirs(exp, exist_ok=True)
    ref_mc = RefTacotron2Config(**SMALL["model"], max_decoder_steps=12, gate_threshold=1.0)
    ref_ac = ref_configs.AudioConfig(filter_length=1024, hop_length=256, n_mels=SMALL["n_mels"], log_func="np.log")
    ref_trainer = ref_configs.TrainerConfig(project_name="fixture", experiment_id="fixture", use_wandb=False, run_eval=False)
    torch.manual_seed(1234)
    model = RefTacotron2(ref_mc, ref_ac, ref_tc).eval()
    # non-trivial BatchNorm statistics, as after training
    g = torch.Generator().manual_seed(5)
    for name, buf in model.named_buffers():
        if name.endswith("running_mean"):
            buf.copy_(torch.randn(buf.shape, generator=g) * 0.1)
        elif name.endswith("running_var"):
            buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
    ref_configs.BaseConfig.write_configs_to_file(
        path=os.path.join(exp, "config.yaml"),
        configs={"model_config": model.model_config, "trainer_config": ref_trainer, "audio_config": ref_ac, "text_config": ref_tc})
    model_dict = model.get_checkpoint_statedicts(optimizer=None)
    model_dict["iteration"] = 3
    torch.save(model_dict, os.path.join(exp, "checkpoint_3.pt"))
    text = HOST_SENTENCES[3]
    tokens = torch.IntTensor(tp.tokens_to_indices(tp.tokenize(text))).unsqueeze(0)
    steps = ref_mc.max_decoder_steps
    pairs = draw_keep_masks(77, (1, ref_mc.prenet_dim), n_pairs=steps)
    masks = torch.stack(pairs, dim=1).squeeze(2)
    torch.manual_seed(77)
    out = model.inference({"tokens": tokens})
    np.savez_compressed(os.path.join(exp, "expected.npz"), tokens=tokens.numpy(), text=np.array(text),
                        keep_masks_packed=np.packbits(masks.numpy().reshape(2, -1), axis=1),
                        **{k: v.numpy() for k, v in out.items()})
    print("ref_exp: frames", out["mel_outputs"].shape[2], "files", sorted(os.listdir(exp)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "train":
        make_train_case(TRAIN_CASE)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "host":
        make_host_fixtures()
        sys.exit(0)
    for name, case in TF_CASES.items():
        make_tf_case(name, case)
    for name, case in AR_CASES.items():
        make_ar_case(name, case)
    make_audio_case()
    make_host_fixtures()
