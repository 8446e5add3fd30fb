#!/usr/bin/env python3
"""Counter collection for the headline's dominant kernel, decoder_lstm_step_pa_kernel (the launch beside the resident
attention kernel).  rocprofv3 --pmc serialises kernels, so the loop itself cannot run as it does in production (its
hand-offs need both kernels at once: they time out - quickly here, GVX_HANDOFF_SPIN_LIMIT - and the call's outputs are
NaN by design).  What IS meaningful under the counters: the 64 back-to-back replays of a mid-sequence launch that the
kernel-timing pass issues after the loop (gvx_api.hip, decoder_tf_impl: the context counter already stands at its final
value, nothing waits) - the same launches bench.py's roofline figure times.  tools/pmc_pa_summary.py keeps only those.

    rocprofv3 --kernel-trace --output-format csv --pmc <counters> -d gpurun_out/pmc_pa/<set> -- python3 tools/pa_pmc.py [B]
"""
import os
import sys

os.environ.setdefault("GVX_HANDOFF_SPIN_LIMIT", "2000")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from genvox_amd import weights as gw
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T, L = 64, 128
mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, log_func="np.log"), TextConfig(n_tokens=40)
m = Tacotron2(mc, ac, tc)
m.load_state_dict(gw.generate_state_dict(mc, ac, tc, 0))
m = m.to("cuda:0")
batch = {k: torch.from_numpy(v).cuda() for k, v in gw.synthetic_inputs(B, L, T, 40, 80, seed=3).items()}
m.enable_kernel_timing(True)
m.forward(batch)
torch.cuda.synchronize()
print(m.kernel_times_ms())
