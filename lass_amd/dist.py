"""Clip-level sharding of the evaluation set + the single exchange step.

Clips are independent (eval-mode BN), so each rank separates a contiguous block of the clip list with replicated
weights and no data-path collective; one all-gather of the per-clip (sdr, sdri, sisdr) triples ends the run.  With
torch.distributed backend "nccl" that all-gather is RCCL over xGMI; the same code runs on "gloo" for CPU tests."""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block partition: rank r takes [r*n/W, (r+1)*n/W) (sizes differ by at most one)."""
    return (rank * n) // world_size, ((rank + 1) * n) // world_size


def gather_rows(local: np.ndarray, n_total: int, device=None) -> np.ndarray:
    """All-gather per-clip rows (n_local, k) f64 from every rank into (n_total, k), in clip order.

    Ragged shards are padded with NaN to the largest shard so ONE fixed-size all_gather suffices."""
    rank, ws = world()
    local = np.asarray(local, dtype=np.float64).reshape(-1, local.shape[-1] if local.ndim > 1 else 1)
    if ws == 1 and not (dist.is_available() and dist.is_initialized()):
        assert local.shape[0] == n_total
        return local
    k = local.shape[1]
    cap = max(shard_range(n_total, r, ws)[1] - shard_range(n_total, r, ws)[0] for r in range(ws))
    buf = torch.full((cap, k), float("nan"), dtype=torch.float64)
    buf[:local.shape[0]] = torch.from_numpy(local)
    if dist.get_backend() == "nccl":
        buf = buf.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    out = [torch.empty_like(buf) for _ in range(ws)]
    dist.all_gather(out, buf)
    rows = []
    for r in range(ws):
        lo, hi = shard_range(n_total, r, ws)
        rows.append(out[r][:hi - lo].cpu().numpy())
    return np.concatenate(rows, axis=0)
