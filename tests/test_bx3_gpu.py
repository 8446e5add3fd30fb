"""GPU: the opt-in bf16x3 GEMM (csrc/gemm_bx3.hip, GVX_GEMM_BX3=1 - read once per process, hence a child process): the
teacher-forced fixtures, the Postnet and the encoder through it against the reference's numbers at the usual 1e-3, and bit
reproducibility of repeated single-stream calls.  (Why it is not the default: see the header of gemm_bx3.hip.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import torch
from genvox_amd.tacotron2 import Tacotron2
from tests.golden.cases import TF_CASES, case_configs
from tests.helpers import TOL, case_state_dict, load_fixture, max_abs_diff, tf_batch, unpack_masks
KEYS = ("alignments", "gate_outputs", "mel_outputs", "mel_outputs_postnet")
worst = 0.0
for name in ("tf_full", "tf_small"):
    case, fx = TF_CASES[name], load_fixture(name)
    mc, ac, tc = case_configs(case)
    m = Tacotron2(mc, ac, tc); m.load_state_dict(case_state_dict(name)); m = m.to("cuda:0").eval()
    masks = unpack_masks(fx["keep_masks_packed"], (2, (case["T"] + 1) * case["B"], mc.prenet_dim))
    batch = {**tf_batch(fx), "prenet_keep_masks": masks}
    a, b = m.forward(batch), m.forward(batch)
    for k in KEYS:
        assert torch.equal(a[k], b[k]), (name, k)
        d = max_abs_diff(a[k], fx[k]); worst = max(worst, d)
        assert d <= TOL, (name, k, d)
    m.check_status()
print("worst abs diff", worst)
'''


def test_bf16x3_gemm_opt_in_matches_the_reference_fixtures():
    env = dict(os.environ, GVX_GEMM_BX3="1", PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "worst abs diff" in r.stdout
