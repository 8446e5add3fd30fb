"""The RCCL branch of the multi-GPU start-up on real hardware: a one-rank "nccl" process group
initialised IN this process, then the library's own `broadcast_packed_weights` / `gather_mels` on cuda:0 and a forward on the
received blob.  (The N > 1 protocol - size check, config mismatch on every rank, shard planning, gather - is covered by the
world-size-2 gloo test in tests/test_dist_cpu.py; an N > 1 RCCL run needs more than the one GPU a test box has.)"""
import socket

import pytest
import torch
import torch.distributed as dist

from genvox_amd import dist as gdist
from genvox_amd import weights as gw
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import Tacotron2

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_one_rank_broadcast_and_gather_on_the_gpu():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", world_size=1, rank=0)
    try:
        assert dist.get_backend() == "nccl"
        mc, ac, tc = Tacotron2Config(), AudioConfig(filter_length=1024, log_func="np.log"), TextConfig(n_tokens=40)
        m = Tacotron2(mc, ac, tc)
        m.load_state_dict(gw.generate_state_dict(mc, ac, tc, seed=0))
        m = m.to("cuda:0")
        before = m.packed_blob().clone()
        gdist.broadcast_packed_weights(m, src=0)      # size broadcast, all-reduce of the agreement flag, blob broadcast: all RCCL
        assert torch.equal(m.packed_blob(), before)
        inp = gw.synthetic_inputs(4, 24, 16, tc.n_tokens, ac.n_mels, seed=1)
        batch = gdist.shard_batch({k: torch.from_numpy(v) for k, v in inp.items()}, 0, 1)
        out = m.forward(batch)
        m.check_status()
        parts = gdist.gather_mels(out["mel_outputs_postnet"], batch["mel_lengths"].to("cuda:0"), 16)   # all_gather over RCCL
        assert len(parts) == 1 and torch.equal(parts[0], out["mel_outputs_postnet"])
        t = torch.ones(3, device="cuda:0")
        dist.all_reduce(t)
        assert torch.equal(t, torch.ones(3, device="cuda:0"))
    finally:
        dist.destroy_process_group()
