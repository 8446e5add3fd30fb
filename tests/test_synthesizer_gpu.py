"""GPU: the drop-in surface end to end - config.yaml + checkpoint .pt in the reference's formats -> Synthesizer.tts -> waveform;
the collate mirror feeding Tacotron2.forward."""
import numpy as np
import pytest
import torch

from genvox_amd import weights as gw
from genvox_amd.collate import TextMelCollateFn
from genvox_amd.configs import AudioConfig, BaseConfig, Tacotron2Config, TextConfig
from genvox_amd.synthesizer import Synthesizer
from genvox_amd.tacotron2 import Tacotron2
from genvox_amd.text import TextProcessor

pytestmark = pytest.mark.gpu


def test_synthesizer_from_reference_style_checkpoint(tmp_path):
    text = "Dr. Smith paid $3.50 on the 21st, in 1999."
    tcfg = TextConfig(cleaners=["base_cleaners"])
    tp = TextProcessor(tcfg)
    tp.tokenize(text + " abcdefghijklmnopqrstuvwxyz")
    tp.generate_token_map()
    mc = Tacotron2Config(max_decoder_steps=40, gate_threshold=1.0)
    ac = AudioConfig(filter_length=1024, hop_length=256, log_func="np.log")
    cfg, ckpt = str(tmp_path / "config.yaml"), str(tmp_path / "checkpoint_1.pt")
    BaseConfig.write_configs_to_file(cfg, {"model_config": mc, "audio_config": ac, "text_config": tcfg, "trainer_config": None})
    sd = gw.generate_state_dict(mc, ac, tcfg, seed=4)
    torch.save({"model_statedict": sd, "iteration": 1}, ckpt)   # the reference's checkpoint dict (tacotron2.py:574-579)
    syn = Synthesizer(tts_model_class=Tacotron2, tts_config_path=cfg, tts_checkpoint_path=ckpt, use_cuda=True)
    out = syn.tts(text)
    T = 40
    assert set(out) == {"mel_outputs", "mel_outputs_postnet", "gate_outputs", "alignments", "waveform", "sampling_rate"}
    assert out["mel_outputs_postnet"].shape == (80, T) and out["gate_outputs"].shape == (T,)
    assert out["alignments"].shape[0] == T and out["sampling_rate"] == 22050
    assert out["waveform"].dtype == np.float64 and out["waveform"].shape == (1024 + (T - 1) * 256 - 1000,)
    assert np.isfinite(out["waveform"]).all() and np.abs(out["waveform"]).max() > 0
    # same seed -> same Prenet masks -> identical mels (the masks are drawn from a seed taken from torch's RNG)
    torch.manual_seed(7)
    a = syn.tts(text)["mel_outputs_postnet"]
    torch.manual_seed(7)
    b = syn.tts(text)["mel_outputs_postnet"]
    assert np.array_equal(a, b)


def test_collate_feeds_forward():
    mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, log_func="np.log"), TextConfig(n_tokens=30)
    m = Tacotron2(mc, ac, tc).to("cuda:0")
    g = torch.Generator().manual_seed(0)
    items = [{"tokens": torch.randint(0, 30, (n,), generator=g, dtype=torch.int32), "features": torch.randn(80, t, generator=g)}
             for n, t in ((9, 14), (21, 30), (15, 22))]
    batch = TextMelCollateFn()(items)
    assert batch["token_lengths"].tolist() == [21, 15, 9] and batch["mel_lengths"].tolist() == [30, 22, 14]
    assert batch["gate_padded"][2, 13] == 1 and batch["gate_padded"][2, 12] == 0
    out = m.forward(m.prepare_batch(batch, "cuda:0"))
    assert out["mel_outputs_postnet"].shape == (3, 80, 30) and torch.isfinite(out["mel_outputs_postnet"]).all()
    assert torch.all(out["gate_outputs"][2, 14:] == 1e3)


def test_c_abi_rejects_bad_arguments_loudly():
    """Error behaviour of the boundary: status code + message, no crash, no silent fallback."""
    import ctypes as C

    from genvox_amd import _lib

    lib = _lib.load()
    mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, log_func="np.log"), TextConfig(n_tokens=12)
    m = Tacotron2(mc, ac, tc).to("cuda:0")
    # token id outside the embedding table: flagged, not an out-of-bounds read (row is embedded as zeros)
    bad = {"token_padded": torch.full((2, 8), 99, dtype=torch.long), "token_lengths": torch.tensor([8, 8]),
           "mel_padded": torch.zeros(2, 80, 4), "gate_padded": torch.zeros(2, 4), "mel_lengths": torch.tensor([4, 4])}
    out = m.forward(bad)
    assert torch.isfinite(out["mel_outputs"]).all()
    with pytest.raises(IndexError):
        m.check_status()
    m.forward({**bad, "token_padded": torch.zeros(2, 8, dtype=torch.long)})
    m.check_status()   # the flag is per call
    h = m._handle
    ws = m._get_workspace(2, 8, 4)
    s = torch.cuda.current_stream().cuda_stream
    x = torch.zeros(16, device="cuda")
    # batch over the per-call limit, workspace too small, misaligned workspace, null outputs
    assert lib.gvx_encoder_forward(h, x.data_ptr(), None, 65, 8, x.data_ptr(), ws.data_ptr(), ws.numel(), s) == -2
    assert b"batch" in lib.gvx_last_error()
    assert lib.gvx_encoder_forward(h, x.data_ptr(), None, 2, 8, x.data_ptr(), ws.data_ptr(), 1024, s) == -5
    assert b"workspace too small" in lib.gvx_last_error()
    assert lib.gvx_encoder_forward(h, x.data_ptr(), None, 2, 8, x.data_ptr(), ws.data_ptr() + 4, ws.numel() - 4, s) == -5
    assert lib.gvx_postnet_forward(h, None, None, 2, 4, None, ws.data_ptr(), ws.numel(), s) == -1
    assert lib.gvx_postnet_forward(h, x.data_ptr(), None, 2, 4, x.data_ptr(), ws.data_ptr(), 256, s) == -5   # Postnet plan too small
    assert lib.gvx_mask_padding(None, None, None, None, 2, 80, 4, s) == -1
    # the recurrent entry points take at most 64 rows per call; the host mirror chunks larger batches itself (below)
    assert lib.gvx_decoder_autoregressive(h, x.data_ptr(), None, 65, 8, 4, 0.5, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(),
                                          x.data_ptr(), None, ws.data_ptr(), ws.numel(), s) == -2
    # a teacher-forced batch over the limit is split transparently and matches per-chunk calls
    B, L, T = 70, 6, 3
    g = torch.Generator().manual_seed(1)
    big = {"token_padded": torch.randint(0, 12, (B, L), generator=g), "token_lengths": torch.full((B,), L),
           "mel_padded": torch.randn(B, 80, T, generator=g), "gate_padded": torch.zeros(B, T), "mel_lengths": torch.full((B,), T)}
    masks = torch.from_numpy(gw.prenet_keep_masks((T + 1) * B, mc.prenet_dim)).reshape(2, T + 1, B, mc.prenet_dim)
    whole = m.forward({**big, "prenet_keep_masks": masks})
    part = m.forward({**{k: v[64:] for k, v in big.items()}, "prenet_keep_masks": masks[:, :, 64:].contiguous()})
    assert whole["mel_outputs"].shape == (B, 80, T)
    # (rows 64.. run as a chunk of a two-stream batch in `whole` and as a batch of their own in `part`: the latter takes the
    # persistent-attention loop, whose fp32 sums are ordered differently - equal to rounding, not bit for bit)
    assert (whole["mel_outputs_postnet"][64:] - part["mel_outputs_postnet"]).abs().max() <= 1e-5


@pytest.mark.parametrize("name", ["tf_small", "tf_full"])
def test_eval_step_criterion_matches_reference(name):
    """Tacotron2.eval_step + Tacotron2Loss (models/tts/tacotron2.py:524-529, :598-615): the HIP reduction on the
    reference's own outputs reproduces the reference criterion's values (fixtures), and eval_step fills
    loss_items_eval from the GPU forward like the reference does."""
    from genvox_amd.tacotron2 import Tacotron2Loss
    from oracle import tacotron2_ref
    from tests.golden.cases import TF_CASES, case_configs
    from tests.helpers import case_state_dict, load_fixture, tf_batch, unpack_masks

    case, fx = TF_CASES[name], load_fixture(name)
    batch = {"mel_padded": torch.from_numpy(fx["mel_padded"]), "gate_padded": torch.from_numpy(fx["gate_padded"])}
    ref_out = {k: torch.from_numpy(fx[k]).cuda() for k in ("mel_outputs", "mel_outputs_postnet", "gate_outputs")}
    got = Tacotron2Loss(batch, ref_out)
    for k in ("loss", "mel_loss", "gate_loss"):
        assert got[k].shape == () and got[k].dtype == torch.float32
        assert abs(float(got[k]) - float(fx[k])) <= 2e-6 * max(1.0, abs(float(fx[k]))), k
    again = Tacotron2Loss(batch, ref_out)
    assert all(torch.equal(got[k], again[k]) for k in got)   # fixed reduction order

    mc, ac, tc = case_configs(case)
    m = Tacotron2(mc, ac, tc)
    m.load_state_dict(case_state_dict(name))
    m = m.to("cuda:0")
    masks = unpack_masks(fx["keep_masks_packed"], (2, (case["T"] + 1) * case["B"], mc.prenet_dim))
    full = {**tf_batch(fx), "prenet_keep_masks": masks}
    out = m.eval_step(full, m.get_criterion(), None)
    want = tacotron2_ref.tacotron2_loss({k: v for k, v in full.items()}, {k: v.cpu() for k, v in out.items()})
    for k in ("loss", "mel_loss", "gate_loss"):
        assert abs(m.loss_items_eval[k + "_eval"] - float(want[k])) <= 2e-6 * max(1.0, abs(float(want[k])))
        assert abs(m.loss_items_eval[k + "_eval"] - float(fx[k])) <= 1e-4 * max(1.0, abs(float(fx[k])))   # forward parity carried through
    assert m.get_eval_priority() == m.loss_items_eval["loss_eval"]
    with pytest.raises(ValueError):
        Tacotron2Loss({"mel_padded": batch["mel_padded"][:, :, :-1], "gate_padded": batch["gate_padded"]}, ref_out)


def test_synthesizer_on_files_written_by_the_reference(monkeypatch):
    """tests/golden/ref_exp/ holds config.yaml and checkpoint_3.pt exactly as GenVox writes them
    (its own initialisation, reduced dims) and the reference's inference outputs from them.  Synthesizer loads both files;
    the model reproduces the reference's mels for the same text and Prenet masks, and tts() runs end to end."""
    import os

    from tests.helpers import GOLDEN, TOL, max_abs_diff, unpack_masks

    exp = os.path.join(GOLDEN, "ref_exp")
    syn = Synthesizer(tts_model_class=Tacotron2, tts_config_path=os.path.join(exp, "config.yaml"),
                      tts_checkpoint_path=os.path.join(exp, "checkpoint_3.pt"), use_cuda=True)
    with np.load(os.path.join(exp, "expected.npz")) as z:
        fx = {k: z[k] for k in z.files}
    text = str(fx["text"])
    tokens = syn.text_processor.tokens_to_indices(syn.text_processor.tokenize(text))
    assert tokens == fx["tokens"][0].tolist()
    mc = syn.tts_model.model_config
    steps = mc.max_decoder_steps
    masks = unpack_masks(fx["keep_masks_packed"], (2, steps, mc.prenet_dim)).reshape(2, steps, 1, mc.prenet_dim)
    out = syn.tts_model.inference({"tokens": torch.IntTensor(tokens).unsqueeze(0), "prenet_keep_masks": masks})
    for k in ("mel_outputs", "mel_outputs_postnet", "gate_outputs", "alignments"):
        assert out[k].shape == fx[k].shape, k
        assert max_abs_diff(out[k], fx[k]) <= TOL, k
    res = syn.tts(text)
    assert res["mel_outputs_postnet"].shape == (mc_n_mels(syn), steps) and np.isfinite(res["waveform"]).all()


def mc_n_mels(syn):
    return syn.tts_model.audio_config.n_mels
