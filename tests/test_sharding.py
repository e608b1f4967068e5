"""Multi-rank host logic on CPU: pose sharding, screen-bin ownership, and the packed-bin
all_gather with world_size 2 and 3 over the gloo backend."""
import os
import socket

import numpy as np
import pytest

import wgpu_renderer_amd  # noqa: F401
from wgpu_renderer_amd import sharding


def test_pose_shards_partition_the_batch():
    for n, world in ((512, 8), (512, 3), (5, 8), (0, 2)):
        seen = []
        for r in range(world):
            seen += list(sharding.pose_shard(n, r, world))
        assert seen == list(range(n))
        sizes = [len(sharding.pose_shard(n, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


def test_bins_are_dealt_round_robin_and_cover_the_frame():
    w, h = 3840, 2160
    bx, by = sharding.bin_grid(w, h)
    assert (bx, by) == (60, 34)
    for world in (1, 2, 4, 8, 7):
        allb = sorted(b for r in range(world) for b in sharding.owned_bins(w, h, r, world))
        assert allb == list(range(bx * by))
        assert all(len(sharding.owned_bins(w, h, r, world)) <= sharding.bins_per_rank(w, h, world) for r in range(world))


@pytest.mark.parametrize("w,h,world", [(1920, 1080, 3), (333, 207, 2), (64, 64, 4)])
def test_pack_scatter_round_trip_single_process(w, h, world):
    rng = np.random.default_rng(0)
    frame = rng.integers(0, 2 ** 32, (h, w), dtype=np.uint32)
    out = np.zeros_like(frame)
    for r in range(world):
        sharding.scatter_bins_host(out, sharding.pack_bins_host(frame, r, world), r, world)
    assert np.array_equal(out, frame)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank "renders" the same deterministic frame but keeps only its own bins
        rng = np.random.default_rng(42)
        truth = rng.integers(0, 2 ** 31, (h, w), dtype=np.int64).astype(np.uint32)
        mine = np.zeros_like(truth)
        sharding.scatter_bins_host(mine, sharding.pack_bins_host(truth, rank, world), rank, world)  # foreign bins stay 0
        packed = torch.from_numpy(sharding.pack_bins_host(mine, rank, world).astype(np.int32))
        gathered = sharding.gather_packed(packed, world).numpy().astype(np.uint32)
        frame = np.zeros_like(truth)
        for r in range(world):
            sharding.scatter_bins_host(frame, gathered[r], r, world)
        ok = bool(np.array_equal(frame, truth))
        # frame-parallel: disjoint pose shards, slowest rank's time is what gets reported
        poses = list(sharding.pose_shard(17, rank, world))
        counts = [None] * world
        dist.all_gather_object(counts, poses)
        ok &= sorted(p for c in counts for p in c) == list(range(17))
        slow = sharding.max_over_ranks(1.0 + rank)
        ok &= slow == float(world)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_bin_gather_and_pose_split(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 200, 150, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]


# ------------------------------------------------------------------ wr_gather_layout / wr_gather_bins_via
def test_gather_layout_matches_the_round_robin_deal():
    """The C ABI's layout function (what wr_gather_bins_via addresses the root's receive area
    with) against the Python restatement, ragged sizes and worlds that do not divide the bins."""
    from wgpu_renderer_amd import abi
    for (w, h) in ((3840, 2160), (1920, 1080), (333, 207), (64, 64), (1, 1)):
        for world in (1, 2, 3, 4, 7, 8):
            slots = {abi.gather_layout(w, h, world, r)[1] for r in range(world)}
            assert slots == {sharding.bins_per_rank(w, h, world)}
            for r in range(world):
                assert abi.gather_layout(w, h, world, r)[0] == len(sharding.owned_bins(w, h, r, world))
    assert abi.gather_layout(64, 64, 4, 9) == (0, 1)  # a rank outside the world owns nothing


def _gather_worker(rank, world, port, w, h, root, q):
    """One rank of a gather to `root` with gloo send / recv standing in for the transport: every
    rank packs its bins, the root receives each peer's bins at the slot wr_gather_layout names
    and writes them into its frame, exactly the steps of wr_gather_bins_via."""
    import torch
    import torch.distributed as dist
    from wgpu_renderer_amd import abi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(7)
        truth = rng.integers(0, 2 ** 31, (h, w), dtype=np.int64).astype(np.uint32)
        mine = np.zeros_like(truth)
        sharding.scatter_bins_host(mine, sharding.pack_bins_host(truth, rank, world), rank, world)  # only this rank's bins
        owned, slot = abi.gather_layout(w, h, world, rank)
        ok = True
        if rank != root:
            packed = sharding.pack_bins_host(mine, rank, world)[:owned]
            if owned:
                dist.send(torch.from_numpy(packed.astype(np.int32).copy()), dst=root)
        else:
            recv_area = np.zeros((world * slot, sharding.BIN, sharding.BIN), np.uint32)
            for r in range(world):
                if r == root:
                    continue
                owned_r = abi.gather_layout(w, h, world, r)[0]
                if owned_r:
                    buf = torch.zeros((owned_r, sharding.BIN, sharding.BIN), dtype=torch.int32)
                    dist.recv(buf, src=r)
                    recv_area[r * slot: r * slot + owned_r] = buf.numpy().astype(np.uint32)
            frame = mine.copy()  # the root's own bins are in place
            bx, by = sharding.bin_grid(w, h)
            for b in range(bx * by):  # k_scatter_gathered: bin b is rank b % world's (b // world)-th
                r, j = b % world, b // world
                if r == root:
                    continue
                oy, ox = (b // bx) * sharding.BIN, (b % bx) * sharding.BIN
                th, tw = min(sharding.BIN, h - oy), min(sharding.BIN, w - ox)
                frame[oy:oy + th, ox:ox + tw] = recv_area[r * slot + j, :th, :tw]
            ok = bool(np.array_equal(frame, truth))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,root", [(2, 0), (3, 2)])
def test_gloo_gather_to_root_with_the_c_layout(world, root):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, 333, 207, root, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]


# ---------------------------------------------------------------- bench.py: starting N ranks
def _bench(*argv, env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], capture_output=True, text=True, env=e, timeout=120)


def test_bench_starts_its_own_ranks_when_asked_for_more_than_one_gpu():
    """`python bench.py --gpus N` without WORLD_SIZE is the parent of N ranks: the command it
    would start is torch.distributed.run with one rank per GPU on 127.0.0.1, with the caller's
    arguments passed on; with fewer devices than ranks it refuses (here: none) before starting
    anything."""
    import json
    p = _bench("--gpus", "4", "--steps", "7", "--warmup", "2", "--print-launch")
    assert p.returncode == 0, p.stderr
    cmd = json.loads(p.stdout)["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    script = cmd.index(next(c for c in cmd if c.endswith("bench.py")))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    p = _bench("--gpus", "2", "--steps", "3")
    assert p.returncode == 2 and "not starting" in p.stderr and p.stdout == ""


def _ranks_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import bench
    r = bench.Ranks()
    try:
        r.barrier()
        got = (r.rank, r.world, r.max_time(1.0 + rank), r.from_root(b"id-%d" % rank if rank == 0 else None))

        # a scale workload that fails on one rank only: every rank reports it and goes on
        def fails_on_rank_one():
            if rank == 1:
                raise RuntimeError("no such device")
            return {"value": 1.0}
        got += (bench.guarded(r, fails_on_rank_one), bench.guarded(r, lambda: {"value": 2.0 + r.max_time(rank)}))
        q.put(got)
    finally:
        r.close()


def test_bench_ranks_talk_over_gloo():
    """The control plane of a multi-rank bench run (barrier, slowest rank's time, the
    communicator id from rank 0) with world size 2 on CPU."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ranks_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [g[:4] for g in got] == [(0, 2, 2.0, b"id-0"), (1, 2, 2.0, b"id-0")]
    assert got[0][4] == {"error": "failed on another rank"} and got[1][4] == {"error": "RuntimeError: no such device"}
    assert got[0][5] == got[1][5] == {"value": 3.0}
