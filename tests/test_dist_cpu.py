"""CPU: the N > 1 host path with world_size 2 on gloo (sharding plan, blob-broadcast protocol, gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from genvox_amd import dist as gdist


def test_shard_plans():
    assert [gdist.shard_rows(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    lens = [5, 40, 12, 33, 7, 21, 9]
    plan = gdist.plan_shards(lens, 3)
    flat = [i for p in plan for i in p]
    assert sorted(flat) == list(range(len(lens)))
    ordered = [lens[i] for i in flat]
    assert ordered == sorted(lens, reverse=True)  # contiguous runs of the length-sorted order
    assert max(len(p) for p in plan) - min(len(p) for p in plan) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the product's weight broadcast (genvox_amd.dist.broadcast_packed_weights) on CPU-resident models: rank 0 holds the
        # "checkpoint", rank 1 a different random init; afterwards rank 1 must hold rank 0's packed blob bit for bit
        from genvox_amd import weights as gw
        from genvox_amd.tacotron2 import Tacotron2
        from tests.golden.cases import TF_CASES, case_configs

        mc, ac, tc = case_configs(TF_CASES["tf_small"])
        model = Tacotron2(mc, ac, tc)
        sd0 = gw.generate_state_dict(mc, ac, tc, seed=2)
        model.load_state_dict(sd0 if rank == 0 else gw.generate_state_dict(mc, ac, tc, seed=99))
        gdist.broadcast_packed_weights(model, src=0)
        want = Tacotron2(mc, ac, tc)
        want.load_state_dict(sd0)
        want_blob = want.pack_weights_host()
        mine = model.packed_blob() if rank == 0 else model._blob
        ok = bool(torch.equal(mine, want_blob)) and mine.numel() == model.blob_numel()
        # ranks built from different configs must all fail loudly instead of binding a misaligned blob
        other = Tacotron2(*case_configs(TF_CASES["tf_full" if rank == 1 else "tf_small"]))
        try:
            gdist.broadcast_packed_weights(other, src=0)
            ok = False
        except RuntimeError as e:
            ok &= "configs differ" in str(e)
        B, M, T = 5, 8, 12
        batch = {
            "token_padded": torch.arange(B * 6).reshape(B, 6), "token_lengths": torch.tensor([6, 5, 4, 3, 2]),
            "mel_padded": torch.arange(B * M * T, dtype=torch.float32).reshape(B, M, T),
            "gate_padded": torch.zeros(B, T), "mel_lengths": torch.tensor([12, 7, 9, 4, 11]),
        }
        mine = gdist.shard_batch(batch, rank, world)
        lo, hi = gdist.shard_rows(B, rank, world)
        ok &= mine["token_padded"].shape == (hi - lo, int(batch["token_lengths"][lo:hi].max()))
        ok &= mine["mel_padded"].shape[2] == int(batch["mel_lengths"][lo:hi].max())
        parts = gdist.gather_mels(mine["mel_padded"], mine["mel_lengths"], T)
        whole = torch.cat(parts, dim=0)
        ok &= whole.shape == (B, M, T)
        for b in range(B):
            n = int(batch["mel_lengths"][b])
            tmax_r = [int(batch["mel_lengths"][slice(*gdist.shard_rows(B, r, world))].max()) for r in range(world)]
            r_of_b = 0 if b < gdist.shard_rows(B, 0, world)[1] else 1
            ok &= bool(torch.equal(whole[b, :, : tmax_r[r_of_b]], batch["mel_padded"][b, :, : tmax_r[r_of_b]]))
        results[rank] = ok
    finally:
        dist.destroy_process_group()


def test_world2_gloo_broadcast_shard_gather():
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), results), nprocs=world, join=True)
    assert dict(results) == {0: True, 1: True}
