"""world_size-2 gloo test of the N>1 path: block sharding + the single all-gather, incl. a ragged tail."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from talkshow_amd.parallel import gather_sequences, generate_sharded, shard_range


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 32, 1024, 1025):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(n, r, world) for r in range(world)]
            flat = [i for a, b in blocks for i in range(a, b)]
            assert flat == list(range(n))


def _fake_generate(mfcc, ids, clip_index0):
    # a pure per-clip function standing in for the HIP path: depends on the clip's data and its GLOBAL index only
    n = mfcc.shape[0]
    idx = torch.arange(clip_index0, clip_index0 + n, dtype=torch.float32).view(n, 1, 1)
    poses = mfcc.mean(-1, keepdim=True).repeat(1, 1, 3) + idx + ids.view(n, 1, 1).float()
    return None, poses


def _worker(rank, world, port, n, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    mfcc = torch.randn(n, 8, 4, generator=g)
    ids = torch.arange(n) % 4
    local, (a, b) = generate_sharded(_fake_generate, mfcc, ids, batch=3)
    allp = gather_sequences(local, n_total=n)
    if rank == 0:
        torch.save(allp, out)
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [8, 7, 1])   # 1 clip on 2 ranks: rank 1 holds an empty shard
def test_two_ranks_equal_one(tmp_path, n):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "all.pt")
    mp.spawn(_worker, args=(2, port, n, out), nprocs=2, join=True)
    got = torch.load(out)
    g = torch.Generator().manual_seed(0)
    mfcc = torch.randn(n, 8, 4, generator=g)
    _, ref = _fake_generate(mfcc, torch.arange(n) % 4, 0)
    assert got.shape == ref.shape
    np.testing.assert_array_equal(got.numpy(), ref.numpy())      # same answer for 1 and 2 ranks


def _bench_worker(rank, world, port, steps, out):
    """bench.py's N > 1 leg without the GPU: Engine.plan groups the queued batches into passes, every pass's rows are kept, and
    ONE all-gather at the end carries all of them (rows are a function of the GLOBAL clip number only)."""
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, G = 4, 3
    eng = bench.Engine.__new__(bench.Engine)                     # plan() / all_rows() only: no device objects
    eng.G = G
    eng.outputs, k = [], 0
    for size in eng.plan(steps):                                   # what run_steps does, with a stand-in for run_group
        clip0 = (rank * steps + k) * B
        ids = torch.arange(clip0, clip0 + size * B, dtype=torch.float32).view(-1, 1, 1)
        eng.outputs.append(ids * torch.ones(1, 5, 3))
        k += size
    allp = gather_sequences(eng.all_rows())
    if rank == 0:
        torch.save(allp, out)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_bench_exchange_carries_every_step(tmp_path, world):
    steps, B = 7, 4                                                # plan(7) with G = 3: passes of 3, 3, 1 batches
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "all.pt")
    mp.spawn(_bench_worker, args=(world, port, steps, out), nprocs=world, join=True)
    got = torch.load(out)
    assert got.shape == (world * steps * B, 5, 3)
    want = torch.arange(world * steps * B, dtype=torch.float32).view(-1, 1, 1) * torch.ones(1, 5, 3)
    np.testing.assert_array_equal(got.numpy(), want.numpy())       # clip k sits at row k whatever the world size
