"""N>1 path on CPU: world_size-2 gloo run of the frame sharding + slot gather used by bench.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_result(frame, cap, kp_dtype):
    rng = np.random.default_rng(frame)
    n = int(rng.integers(0, cap + 1))
    k = np.zeros(n, kp_dtype)
    k["x"] = rng.random(n).astype(np.float32) * 640
    k["octave"] = frame
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    return k, d


def _worker(rank, world, port, n_frames, cap, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as g
    pkg = g.load_package()
    sh = pkg.sharding
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lo, hi = sh.shard_range(n_frames, rank, world)
    local = sh.pack_slots([_fake_result(f, cap, pkg.capi.KP_DTYPE) for f in range(lo, hi)], cap)
    fmax = -(-n_frames // world)
    bufs = sh.gather_slots(local, fmax, rank, world)
    ok = True
    if rank == 0:
        got = []
        for r in range(world):
            a, b = sh.shard_range(n_frames, r, world)
            got += sh.unpack_slots(bufs[r].numpy()[: b - a], cap, pkg.capi.KP_DTYPE)
        for f, (k, d) in enumerate(got):
            ek, ed = _fake_result(f, cap, pkg.capi.KP_DTYPE)
            ok &= k.tobytes() == ek.tobytes() and (d == ed).all()
        ok &= len(got) == n_frames
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok &= t.item() == float(world)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


@pytest.mark.parametrize("n_frames", [8, 5])
def test_world2_gather(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, 50, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == {0: True, 1: True}


def test_shard_range_partitions(pkg):
    sh = pkg.sharding
    for n in (0, 1, 7, 8, 256):
        for w in (1, 2, 3, 8):
            r = [sh.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    assert sh.slot_bytes(1200) % 16 == 0
