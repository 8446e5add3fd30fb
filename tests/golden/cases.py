"""Definitions of the golden cases: dims, seeds and shapes.  Shared by the generator
(make_fixtures.py, build container only) and by the tests (any machine)."""
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig

FULL = dict(model={}, n_mels=80, n_tokens=40)
# reduced dims, odd kernel sizes different from the defaults, to exercise generality
SMALL = dict(
    model=dict(symbols_embedding_dim=32, encoder_embedding_dim=32, encoder_kernel_size=3, encoder_n_convolutions=2,
               decoder_rnn_dim=64, attention_rnn_dim=48, prenet_dim=24, attention_dim=16,
               attention_location_n_filters=8, attention_location_kernel_size=7,
               postnet_embedding_dim=40, postnet_kernel_size=5, postnet_n_convolutions=3),
    n_mels=24, n_tokens=17)

TF_CASES = {
    "tf_full": dict(dims=FULL, weight_seed=0, input_seed=3, mask_seed=11, B=3, L=24, T=40,
                    token_lengths=[24, 17, 9], mel_lengths=[40, 31, 22]),
    "tf_full_peaky": dict(dims=FULL, weight_seed=1, input_seed=4, mask_seed=12, peaky=True, B=4, L=40, T=48,
                          token_lengths=[40, 33, 33, 5], mel_lengths=[30, 48, 41, 17]),
    "tf_small": dict(dims=SMALL, weight_seed=2, input_seed=5, mask_seed=13, peaky=True, B=5, L=13, T=11,
                     token_lengths=[13, 13, 8, 2, 1], mel_lengths=[11, 3, 9, 11, 1]),
}

AR_CASES = {
    "ar_full_fixed": dict(dims=FULL, weight_seed=0, input_seed=6, mask_seed=14, L=21, max_decoder_steps=36,
                          gate_fires=False),
    "ar_full_gate": dict(dims=FULL, weight_seed=1, input_seed=7, mask_seed=15, peaky=True, L=33, max_decoder_steps=40,
                         gate_fires=True),
    "ar_small_gate": dict(dims=SMALL, weight_seed=2, input_seed=8, mask_seed=16, peaky=True, L=9, max_decoder_steps=24,
                          gate_fires=True),
}

# training mode (BatchNorm batch statistics, encoder / LSTM-output / Postnet dropouts on): forward outputs, loss and the
# gradients of the reference's loss.backward() (models/tts/tacotron2.py:515-522)
TRAIN_CASE = dict(dims=SMALL, weight_seed=3, input_seed=9, mask_seed=17, peaky=True, B=4, L=9, T=10,
                  token_lengths=[9, 7, 4, 2], mel_lengths=[10, 6, 10, 3])

AUDIO_CASE = dict(fs=22050, n_fft=1024, hop=256, n_mels=80, fmin=0.0, fmax=8000.0, log_func="np.log", ref=1.0,
                  frames=40, seed=21)


def case_configs(case):
    d = case["dims"]
    mc = Tacotron2Config(**d["model"])
    ac = AudioConfig(filter_length=1024, hop_length=256, n_mels=d["n_mels"], log_func="np.log")
    tc = TextConfig(n_tokens=d["n_tokens"])
    return mc, ac, tc
