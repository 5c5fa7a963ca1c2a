"""Clip sharding and the result gather for one-process-per-GPU runs.

Clips are independent units (the reference decodes windows on independent threads,
src/main.rs:890-919), so the path shards with no data-path collective; the only exchange is one
all-gather of fixed-stride int32 result records:

    record[i] = [clip_id, n_tokens, tokens[0 .. stride-3]]        (int32, stride = 2 + max tokens)
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def shard_clip_ids(rank: int, world: int, clips_per_rank: int) -> List[int]:
    """Weak scaling: rank r owns clip ids r*C .. (r+1)*C-1."""
    return list(range(rank * clips_per_rank, (rank + 1) * clips_per_rank))


def pack_records(clip_ids: Sequence[int], tokens: Sequence[np.ndarray], max_tokens: int) -> np.ndarray:
    rec = np.zeros((len(clip_ids), 2 + max_tokens), np.int32)
    for i, (cid, tk) in enumerate(zip(clip_ids, tokens)):
        if len(tk) > max_tokens:
            raise ValueError("token row longer than the record stride")
        rec[i, 0] = cid
        rec[i, 1] = len(tk)
        rec[i, 2:2 + len(tk)] = np.asarray(tk, np.int32)
    return rec


def unpack_records(rec: np.ndarray) -> List[Tuple[int, np.ndarray]]:
    out = [(int(r[0]), r[2:2 + int(r[1])].astype(np.int64)) for r in rec]
    out.sort(key=lambda t: t[0])
    return out


def gather_records(dist, rec: np.ndarray, device: str = "cpu") -> np.ndarray:
    """all_gather of equally-shaped records over the initialised process group
    (backend "nccl" = RCCL over xGMI on ROCm, or "gloo" on CPU)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(rec))
    if device != "cpu":
        t = t.to(device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return np.concatenate([o.cpu().numpy() for o in out], axis=0)


def max_over_ranks(dist, value: float, device: str = "cpu") -> float:
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
