#!/usr/bin/env python3
"""Developer diagnostic (GPU box): per-stage max|diff| of the HIP path against the golden fixtures and the oracle."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from genvox_amd.tacotron2 import Tacotron2
from oracle import tacotron2_ref
from tests.golden.cases import AR_CASES, TF_CASES, case_configs
from tests.helpers import case_state_dict, load_fixture, max_abs_diff, tf_batch, unpack_masks


def build(name, case):
    mc, ac, tc = case_configs(case)
    m = Tacotron2(mc, ac, tc)
    m.load_state_dict(case_state_dict(name))
    return m.to("cuda:0"), (mc, ac, tc)


def main():
    only = sys.argv[1:] or None
    for name, case in TF_CASES.items():
        if only and name not in only:
            continue
        fx = load_fixture(name)
        m, (mc, ac, tc) = build(name, case)
        B, T = case["B"], case["T"]
        masks = unpack_masks(fx["keep_masks_packed"], (2, (T + 1) * B, mc.prenet_dim))
        mem = m.encode(torch.from_numpy(fx["token_padded"]), torch.from_numpy(fx["token_lengths"]))
        torch.cuda.synchronize()
        print(f"[{name}] encoder_outputs diff {max_abs_diff(mem, fx['encoder_outputs']):.3e}", flush=True)
        post = m.postnet_residual(torch.from_numpy(fx["mel_outputs"]))
        sd = case_state_dict(name)
        ref_post = torch.from_numpy(fx["mel_outputs"]) + tacotron2_ref.postnet(sd, torch.from_numpy(fx["mel_outputs"]))
        print(f"[{name}] postnet(ref mel) diff vs oracle {max_abs_diff(post, ref_post):.3e}", flush=True)
        batch = tf_batch(fx)
        batch["prenet_keep_masks"] = masks
        out = m.forward(batch)
        torch.cuda.synchronize()
        for k in ("alignments", "gate_outputs", "mel_outputs", "mel_outputs_postnet"):
            print(f"[{name}] {k:22s} diff vs reference fixture {max_abs_diff(out[k], fx[k]):.3e}", flush=True)
    for name, case in AR_CASES.items():
        if only and name not in only:
            continue
        fx = load_fixture(name)
        mc, ac, tc = case_configs(case)
        steps = int(fx["max_decoder_steps"])
        mc.max_decoder_steps = steps
        mc.gate_threshold = float(fx["gate_threshold"])
        m = Tacotron2(mc, ac, tc)
        m.load_state_dict(case_state_dict(name))
        m = m.to("cuda:0")
        masks = unpack_masks(fx["keep_masks_packed"], (2, steps, mc.prenet_dim)).reshape(2, steps, 1, mc.prenet_dim)
        out = m.inference({"tokens": torch.from_numpy(fx["tokens"]), "prenet_keep_masks": masks})
        torch.cuda.synchronize()
        print(f"[{name}] frames {out['mel_outputs'].shape[2]} (reference {fx['mel_outputs'].shape[2]})")
        if out["mel_outputs"].shape == fx["mel_outputs"].shape:
            for k in ("alignments", "gate_outputs", "mel_outputs", "mel_outputs_postnet"):
                print(f"[{name}] {k:22s} diff vs reference fixture {max_abs_diff(out[k], fx[k]):.3e}", flush=True)


if __name__ == "__main__":
    main()
