"""N>1 path on CPU: world_size-2 and world_size-8 gloo runs of the frame sharding + slot gather used by bench.py."""
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


def _run_world(world, n_frames, cap=50):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, cap, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    return res


@pytest.mark.parametrize("n_frames", [8, 5])
def test_world2_gather(n_frames):
    assert _run_world(2, n_frames) == {0: True, 1: True}


@pytest.mark.parametrize("n_frames", [5, 8, 13])
def test_world8_gather_with_ragged_frame_counts(n_frames):
    """BASELINE configs[4] at its full rank count (8 processes, gloo): 8 frames = one per rank; 5 frames leave three ranks with an
    empty shard (their padded, zero-count slot still takes part in the collective); 13 frames give five ranks two frames and three
    ranks one.  Rank 0 reassembles every frame's keypoints and descriptors bit for bit in frame order, and the max-over-ranks
    reduction bench.py times with sees every rank."""
    assert _run_world(8, n_frames) == {r: True for r in range(8)}


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


_HOST_PHASE_PROBS = {}   # filled by the parent before it forks the "ranks"


def _host_phase_worker(rank, threads, barrier, q):
    sys.path.insert(0, ROOT)
    import time
    import __graft_entry__ as g
    pkg = g.load_package()
    probs = _HOST_PHASE_PROBS[rank]
    pkg.capi.lba_host_phase(probs, threads)   # (first touch of the buffers)
    barrier.wait()
    t0 = time.perf_counter()
    b, s = pkg.capi.lba_host_phase(probs, threads)
    q.put((rank, b, s, (time.perf_counter() - t0) * 1e3))


def test_lba_host_phase_of_eight_ranks_shares_the_cores(pkg):
    """8 ranks' LocalBA handles build their windows' index structures at the same time.  Every rank caps its
    worker pool at cores / ranks (lba.hip: lba_host_threads; bench.py passes cores / (ranks x handles)), so 8 processes running the
    host part of a batch concurrently must not take much longer than one process alone with the same number of threads (an
    oversubscribed node -- 8 x 2 x 32 threads on the cores of one rank -- would)."""
    ranks, n_win = 8, 6
    threads = max(1, (os.cpu_count() or 1) // ranks)
    mix = pkg.synth.lba_window_mix(0, n_win)
    for m in mix:
        m["n_points"] = 1200 + m["n_points"] // 8   # (small windows: the generator, not the phase under test, is the slow part here)
    probs = [pkg.synth.synth_lba_problem(**kw) for kw in mix]
    for r in range(ranks):
        _HOST_PHASE_PROBS[r] = probs[r % n_win:] + probs[:r % n_win]
    ctx = mp.get_context("fork")

    def run(n_proc):
        barrier, q = ctx.Barrier(n_proc), ctx.Queue()
        ps = [ctx.Process(target=_host_phase_worker, args=(r, threads, barrier, q)) for r in range(n_proc)]
        [p.start() for p in ps]
        out = [q.get(timeout=600) for _ in ps]
        [p.join() for p in ps]
        return max(o[3] for o in out)
    alone = min(run(1) for _ in range(3))
    together = min(run(ranks) for _ in range(3))
    print(f"LocalBA host phase, {n_win} windows, {threads} thread(s) per rank: alone {alone:.1f} ms, {ranks} ranks at once {together:.1f} ms "
          f"({os.cpu_count()} cores)")
    # every core busy with one rank's thread: a generous bound that an oversubscribed pool (8 x 32 threads time-sliced on these
    # cores) would still break
    assert together < 4.0 * alone + 10.0, (alone, together)
