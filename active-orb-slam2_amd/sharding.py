"""Frame sharding across ranks (one process per GPU) and the single exchange step of the path:
gathering the per-frame keypoint/descriptor slots back to rank 0.

Extraction is embarrassingly parallel over frames, so there is no data-path collective inside the
timed extraction; the only communication is one fixed-size gather per step (RCCL over xGMI on the
GPU box: torch.distributed backend "nccl"; "gloo" in the CPU tests).
"""
from __future__ import annotations

import numpy as np

SLOT_HEADER = 16  # int32 n, 3 x int32 pad


def shard_range(n_frames: int, rank: int, world: int):
    """Contiguous block partition; earlier ranks take the remainder."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def slot_bytes(cap: int) -> int:
    """{int32 n; pad; KeyPoint[cap] (28 B each); uint8 desc[cap][32]} rounded to 16 B."""
    return (SLOT_HEADER + cap * (28 + 32) + 15) // 16 * 16


def pack_slots(results, cap: int) -> np.ndarray:
    """results: list of (kps[KP_DTYPE], desc[n,32]) -> uint8 [len(results), slot_bytes(cap)]"""
    out = np.zeros((len(results), slot_bytes(cap)), np.uint8)
    for i, (k, d) in enumerate(results):
        n = len(k)
        if n > cap:
            raise ValueError(f"frame has {n} keypoints, slot capacity {cap}")
        out[i, :4] = np.frombuffer(np.int32(n).tobytes(), np.uint8)
        out[i, SLOT_HEADER:SLOT_HEADER + 28 * n] = np.frombuffer(k.tobytes(), np.uint8)
        o = SLOT_HEADER + 28 * cap
        out[i, o:o + 32 * n] = np.ascontiguousarray(d, np.uint8).reshape(-1)
    return out


def unpack_slots(buf: np.ndarray, cap: int, kp_dtype):
    res = []
    for row in buf:
        n = int(np.frombuffer(row[:4].tobytes(), np.int32)[0])
        k = np.frombuffer(row[SLOT_HEADER:SLOT_HEADER + 28 * n].tobytes(), kp_dtype).copy()
        o = SLOT_HEADER + 28 * cap
        d = row[o:o + 32 * n].reshape(n, 32).copy()
        res.append((k, d))
    return res


def gather_slots(local_slots, frames_per_rank_max: int, rank: int, world: int, device="cpu"):
    """All ranks contribute a [frames_per_rank_max, slot] uint8 tensor (padded); rank 0 receives the
    concatenation.  Returns the gathered tensor list on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist

    t = torch.zeros((frames_per_rank_max, local_slots.shape[1]), dtype=torch.uint8, device=device)
    if len(local_slots):
        t[: len(local_slots)] = torch.from_numpy(local_slots).to(device)
    if world == 1:
        return [t]
    bufs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, bufs, dst=0)
    return bufs
