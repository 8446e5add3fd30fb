"""Multi-GPU plumbing for batched inference: one process per GPU, utterances sharded across ranks.

The forward has no exchange step: utterances never interact (BatchNorm runs on running statistics), so the batch
dimension partitions freely and the only collective is ONE broadcast of the packed weight blob (~113 MB fp32) from
the rank that loaded the checkpoint, over RCCL/xGMI (``torch.distributed`` backend "nccl"; "gloo" works for CPU
rehearsal of the sharding logic).  The reference is single-device.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist


def _world_info() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def broadcast_packed_weights(model, src: int = 0) -> None:
    """Make every rank's model use rank ``src``'s weights: src packs, ONE broadcast of the packed blob, the others bind it.

    Works on the model's device: "nccl" (= RCCL) for CUDA models; with "gloo" and CPU-resident models the same protocol
    runs without a GPU (the blob is then only kept, see ``Tacotron2.bind_packed_blob``)."""
    if not (dist.is_available() and dist.is_initialized()):
        if model._device().type == "cuda":
            model._ensure_packed()
        return
    rank, _world = _world_info()
    if rank == src:
        blob = model.packed_blob()
    else:
        blob = torch.empty(model.blob_numel(), dtype=torch.float32, device=model._device())
    sizes = torch.tensor([blob.numel()], dtype=torch.int64, device=blob.device)
    dist.broadcast(sizes, src=src)   # a rank built from different configs must fail here, not read a misaligned blob
    ok = torch.tensor([int(int(sizes.item()) == blob.numel())], dtype=torch.int64, device=blob.device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)   # every rank learns of a mismatch, so nobody is left waiting in the broadcast
    if int(ok.item()) != 1:
        raise RuntimeError(f"rank {rank}: packed blob has {blob.numel()} floats, rank {src} broadcasts {int(sizes.item())}: "
                           f"model configs differ between ranks")
    dist.broadcast(blob, src=src)
    if rank != src:
        model.bind_packed_blob(blob)


def shard_rows(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous row range [lo, hi) of rank; sizes differ by at most one."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def plan_shards(token_lengths: Sequence[int], world: int) -> List[List[int]]:
    """Assign utterances to ranks: sort by token length (descending, the collate order of the reference,
    models/tts/__init__.py:32) and deal contiguous runs, so every rank's padding stays tight."""
    order = sorted(range(len(token_lengths)), key=lambda i: -int(token_lengths[i]))
    return [order[slice(*shard_rows(len(order), r, world))] for r in range(world)]


def shard_batch(batch: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    """This rank's rows of a collated batch (rows are already sorted by token length), trimmed to its own max lengths."""
    n = batch["token_padded"].shape[0]
    lo, hi = shard_rows(n, rank, world)
    out = {k: v[lo:hi] for k, v in batch.items()}
    if hi > lo:
        out["token_padded"] = out["token_padded"][:, : int(out["token_lengths"].max())]
        tmax = int(out["mel_lengths"].max())
        out["mel_padded"] = out["mel_padded"][:, :, :tmax]
        if "gate_padded" in out:
            out["gate_padded"] = out["gate_padded"][:, :tmax]
    return out


def gather_mels(mel: torch.Tensor, mel_lengths: torch.Tensor, t_max: int) -> List[torch.Tensor]:
    """Optional all-gather of per-rank mel shards [b_r, M, t_r] (padded to t_max) onto every rank."""
    rank, world = _world_info()
    pad = torch.zeros(mel.shape[0], mel.shape[1], t_max, dtype=mel.dtype, device=mel.device)
    pad[:, :, : mel.shape[2]] = mel
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return [pad]   # (an initialised one-rank group still runs the collectives: the RCCL path is then exercised end to end)
    sizes = [torch.zeros(1, dtype=torch.long, device=mel.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([mel.shape[0]], device=mel.device))
    bmax = int(max(int(s.item()) for s in sizes))
    buf = torch.zeros(bmax, mel.shape[1], t_max, dtype=mel.dtype, device=mel.device)
    buf[: mel.shape[0]] = pad
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    return [o[: int(s.item())] for o, s in zip(outs, sizes)]
