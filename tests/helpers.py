"""Shared helpers for the parity tests (fixture loading, weight sets, mask unpacking)."""
import functools
import os

import numpy as np
import torch

from genvox_amd import weights as gw
from tests.golden.cases import AR_CASES, AUDIO_CASE, TF_CASES, case_configs

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-3  # BASELINE.json north_star: mels within 1e-3 (fp32) of the reference


def load_fixture(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@functools.lru_cache(maxsize=4)
def _state_dict_cached(dims_key, weight_seed, peaky):
    case = {**TF_CASES, **AR_CASES}[dims_key]
    mc, ac, tc = case_configs(case)
    return gw.generate_state_dict(mc, ac, tc, seed=weight_seed, peaky_attention=peaky)


def case_state_dict(name):
    case = {**TF_CASES, **AR_CASES}[name]
    return _state_dict_cached(name, case["weight_seed"], case.get("peaky", False))


def unpack_masks(packed, shape):
    n = int(np.prod(shape[1:]))
    bits = np.unpackbits(packed, axis=1)[:, :n]
    return torch.from_numpy(np.ascontiguousarray(bits.reshape(shape)))


def tf_batch(fx):
    return {k: torch.from_numpy(fx[k]) for k in ("token_padded", "token_lengths", "mel_padded", "gate_padded", "mel_lengths")}


def max_abs_diff(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0
