"""Seeded, platform-independent Tacotron2 weights keyed like the reference's state_dict.

The benchmark and the parity tests need random-init weights of the reference
architecture that are bit-identical here and on the GPU box without shipping a
112 MB blob and without torch's RNG (whose streams may differ between builds).
Every tensor element is a pure function of (seed, key name, flat index) through
a splitmix64 counter hash, so the oracle, the HIP path and the imported reference
(tests/golden/make_fixtures.py loads these into it with ``load_state_dict``) all
see the same numbers.

Key names and shapes follow the reference's ``state_dict`` (SURVEY.md section 8 row a1;
reference: models/tts/tacotron2.py:417-448, models/generic.py:5-54).  Magnitudes
follow the reference's initialisers (Xavier-uniform with the per-layer gain,
models/generic.py:15-18/:48-51; embedding bound models/tts/tacotron2.py:420-423;
torch's LSTM default U(-1/sqrt(H), 1/sqrt(H))) so activations have realistic
ranges; BatchNorm statistics are deliberately non-trivial so BN folding is exercised.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

_MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for ch in text.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def hashed_uniform(seed: int, key: str, n: int) -> np.ndarray:
    """n float64 values in [0, 1), a pure function of (seed, key, index)."""
    base = np.uint64((_fnv1a64(key) ^ (seed * 0xD6E8FEB86659FD93)) & 0xFFFFFFFFFFFFFFFF)
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        bits = _splitmix64(_splitmix64(idx + base) ^ base)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def uniform_tensor(seed: int, key: str, shape: Tuple[int, ...], lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = hashed_uniform(seed, key, n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


_GAIN = {"linear": 1.0, "sigmoid": 1.0, "tanh": 5.0 / 3.0, "relu": math.sqrt(2.0)}


def _xavier_bound(shape: Tuple[int, ...], gain: str) -> float:
    receptive = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    fan_in, fan_out = shape[1] * receptive, shape[0] * receptive
    return _GAIN[gain] * math.sqrt(6.0 / (fan_in + fan_out))


def state_dict_spec(model_config, audio_config, text_config) -> "OrderedDict[str, Tuple[Tuple[int, ...], str, object]]":
    """key -> (shape, kind, arg) in the reference's state_dict order.

    kind: 'xavier' (arg = gain), 'uniform' (arg = bound), 'bn_w', 'bn_b', 'bn_mean', 'bn_var', 'count'.
    """
    mc, ac, tc = model_config, audio_config, text_config
    E = mc.encoder_embedding_dim
    assert mc.symbols_embedding_dim == E, "symbols_embedding_dim must equal encoder_embedding_dim (reference: tacotron2.py:210/:233)"
    assert E % 2 == 0
    H = E // 2
    M, P, A, D, a = ac.n_mels, mc.prenet_dim, mc.attention_rnn_dim, mc.decoder_rnn_dim, mc.attention_dim
    F, kl = mc.attention_location_n_filters, mc.attention_location_kernel_size
    ke, pk, PD = mc.encoder_kernel_size, mc.postnet_kernel_size, mc.postnet_embedding_dim
    spec: "OrderedDict[str, Tuple[Tuple[int, ...], str, object]]" = OrderedDict()

    def bn(prefix: str, c: int):
        spec[prefix + ".weight"] = ((c,), "bn_w", None)
        spec[prefix + ".bias"] = ((c,), "bn_b", None)
        spec[prefix + ".running_mean"] = ((c,), "bn_mean", None)
        spec[prefix + ".running_var"] = ((c,), "bn_var", None)
        spec[prefix + ".num_batches_tracked"] = ((), "count", None)

    n_tok = tc.n_tokens
    spec["embedding.weight"] = ((n_tok, E), "uniform", math.sqrt(3.0) * math.sqrt(2.0 / (n_tok + E)))
    for i in range(mc.encoder_n_convolutions):
        p = f"encoder.convolutions.{i}"
        spec[p + ".0.conv.weight"] = ((E, E, ke), "xavier", "relu")
        spec[p + ".0.conv.bias"] = ((E,), "uniform", 1.0 / math.sqrt(E * ke))
        bn(p + ".1", E)
    for sfx in ("", "_reverse"):
        b = 1.0 / math.sqrt(H)
        spec["encoder.lstm.weight_ih_l0" + sfx] = ((4 * H, E), "uniform", b)
        spec["encoder.lstm.weight_hh_l0" + sfx] = ((4 * H, H), "uniform", b)
        spec["encoder.lstm.bias_ih_l0" + sfx] = ((4 * H,), "uniform", b)
        spec["encoder.lstm.bias_hh_l0" + sfx] = ((4 * H,), "uniform", b)
    spec["decoder.prenet.layers.0.linear_layer.weight"] = ((P, M), "xavier", "linear")
    spec["decoder.prenet.layers.1.linear_layer.weight"] = ((P, P), "xavier", "linear")
    b = 1.0 / math.sqrt(A)
    spec["decoder.attention_rnn.weight_ih"] = ((4 * A, P + E), "uniform", b)
    spec["decoder.attention_rnn.weight_hh"] = ((4 * A, A), "uniform", b)
    spec["decoder.attention_rnn.bias_ih"] = ((4 * A,), "uniform", b)
    spec["decoder.attention_rnn.bias_hh"] = ((4 * A,), "uniform", b)
    spec["decoder.attention_layer.query_layer.linear_layer.weight"] = ((a, A), "xavier", "tanh")
    spec["decoder.attention_layer.memory_layer.linear_layer.weight"] = ((a, E), "xavier", "tanh")
    spec["decoder.attention_layer.v.linear_layer.weight"] = ((1, a), "xavier", "linear")
    spec["decoder.attention_layer.location_layer.location_conv.conv.weight"] = ((F, 2, kl), "xavier", "linear")
    spec["decoder.attention_layer.location_layer.location_dense.linear_layer.weight"] = ((a, F), "xavier", "tanh")
    b = 1.0 / math.sqrt(D)
    spec["decoder.decoder_rnn.weight_ih"] = ((4 * D, A + E), "uniform", b)
    spec["decoder.decoder_rnn.weight_hh"] = ((4 * D, D), "uniform", b)
    spec["decoder.decoder_rnn.bias_ih"] = ((4 * D,), "uniform", b)
    spec["decoder.decoder_rnn.bias_hh"] = ((4 * D,), "uniform", b)
    spec["decoder.linear_projection.linear_layer.weight"] = ((M, D + E), "xavier", "linear")
    spec["decoder.linear_projection.linear_layer.bias"] = ((M,), "uniform", 1.0 / math.sqrt(D + E))
    spec["decoder.gate_layer.linear_layer.weight"] = ((1, D + E), "xavier", "sigmoid")
    spec["decoder.gate_layer.linear_layer.bias"] = ((1,), "uniform", 1.0 / math.sqrt(D + E))
    n_post = mc.postnet_n_convolutions
    for i in range(n_post):
        cin = M if i == 0 else PD
        cout = M if i == n_post - 1 else PD
        p = f"postnet.convolutions.{i}"
        spec[p + ".0.conv.weight"] = ((cout, cin, pk), "xavier", "linear" if i == n_post - 1 else "tanh")
        spec[p + ".0.conv.bias"] = ((cout,), "uniform", 1.0 / math.sqrt(cin * pk))
        bn(p + ".1", cout)
    return spec


def generate_state_dict_numpy(model_config, audio_config, text_config, seed: int = 0,
                              peaky_attention: bool = False) -> "OrderedDict[str, np.ndarray]":
    """Random-init weights as numpy arrays (fp32; BN counters int64).

    ``peaky_attention`` scales the attention energy path (v, query, location
    weights) so the softmax is sharp; random init alone gives nearly flat
    alignments that barely exercise masking (SURVEY.md section 7 hard parts).
    """
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, (shape, kind, arg) in state_dict_spec(model_config, audio_config, text_config).items():
        if kind == "xavier":
            b = _xavier_bound(shape, arg)
            t = uniform_tensor(seed, key, shape, -b, b)
        elif kind == "uniform":
            t = uniform_tensor(seed, key, shape, -arg, arg)
        elif kind == "bn_w":
            t = uniform_tensor(seed, key, shape, 0.5, 1.5)
        elif kind in ("bn_b", "bn_mean"):
            t = uniform_tensor(seed, key, shape, -0.3, 0.3)
        elif kind == "bn_var":
            t = uniform_tensor(seed, key, shape, 0.5, 1.5)
        elif kind == "count":
            t = np.array(1000, dtype=np.int64)
        else:  # pragma: no cover
            raise ValueError(kind)
        out[key] = t
    if peaky_attention:
        att = "decoder.attention_layer."
        out[att + "v.linear_layer.weight"] = out[att + "v.linear_layer.weight"] * np.float32(6.0)
        out[att + "query_layer.linear_layer.weight"] = out[att + "query_layer.linear_layer.weight"] * np.float32(2.0)
        out[att + "location_layer.location_conv.conv.weight"] = \
            out[att + "location_layer.location_conv.conv.weight"] * np.float32(4.0)
        out[att + "memory_layer.linear_layer.weight"] = out[att + "memory_layer.linear_layer.weight"] * np.float32(2.0)
    return out


def generate_state_dict(model_config, audio_config, text_config, seed: int = 0, peaky_attention: bool = False):
    """Same as :func:`generate_state_dict_numpy` but as torch CPU tensors (``load_state_dict`` ready)."""
    import torch

    sd = generate_state_dict_numpy(model_config, audio_config, text_config, seed, peaky_attention)
    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v)).reshape(v.shape)) for k, v in sd.items())


def synthetic_inputs(B: int, L: int, T: int, n_tokens: int, n_mels: int, seed: int = 3,
                     token_lengths=None, mel_lengths=None) -> Dict[str, np.ndarray]:
    """Synthetic teacher-forcing batch in the reference's collate layout
    (reference: models/tts/__init__.py:29-62): tokens uniform in [0, n_tokens), mels ~ N(0,1) (Box-Muller on the
    hashed uniforms), zero padding past each row's length, token lengths descending."""
    tok = (hashed_uniform(seed, "tokens", B * L) * n_tokens).astype(np.int64).reshape(B, L)
    u1 = hashed_uniform(seed, "mel_u1", B * n_mels * T)
    u2 = hashed_uniform(seed, "mel_u2", B * n_mels * T)
    mel = (np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32).reshape(B, n_mels, T)
    tl = np.full(B, L, np.int64) if token_lengths is None else np.asarray(token_lengths, np.int64)
    ml = np.full(B, T, np.int64) if mel_lengths is None else np.asarray(mel_lengths, np.int64)
    assert tl.shape == (B,) and ml.shape == (B,) and tl.max() == L and ml.max() == T
    assert np.all(np.diff(tl) <= 0), "token lengths must be descending (collate sort order)"
    gate = np.zeros((B, T), np.float32)
    for b in range(B):
        tok[b, tl[b]:] = 0
        mel[b, :, ml[b]:] = 0.0
        gate[b, ml[b] - 1:] = 1.0
    return {"token_padded": tok, "token_lengths": tl, "mel_padded": mel, "gate_padded": gate, "mel_lengths": ml}


def prenet_keep_masks(n_rows: int, prenet_dim: int, seed: int = 11) -> np.ndarray:
    """Two Bernoulli(0.5) keep masks as uint8 {0,1}, shape [2, n_rows, prenet_dim].

    The Prenet's dropout is part of the function (reference: models/tts/tacotron2.py:143,
    always training=True), so parity tests pass the masks explicitly."""
    u = hashed_uniform(seed, "prenet_masks", 2 * n_rows * prenet_dim)
    return (u < 0.5).astype(np.uint8).reshape(2, n_rows, prenet_dim)
