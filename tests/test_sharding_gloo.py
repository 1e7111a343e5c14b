"""world_size-2 rehearsal of the multi-GPU commitment on CPU (gloo): range split, all_gather of
the partial points (or all_to_all of bucket partials and per-rank slice reduction), combination
on rank 0.  The per-rank MSM itself is stood in by the oracle
here (no GPU in this suite); on GPUs the same protocol runs with plonk_hip_kzg_commit_dev."""
import os
import random
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import capi as c
from oracle import pyref as o
from plonk_amd import sharding


def test_shard_ranges_partition_exactly():
    for n, world in ((16, 1), (16, 2), (17, 4), (1 << 24, 8), (5, 8)):
        spans = [sharding.shard_range(n, world, r) for r in range(world)]
        assert spans[0][0] == 0
        for (f0, c0), (f1, _) in zip(spans, spans[1:]):
            assert f0 + c0 == f1
        assert spans[-1][0] + spans[-1][1] == n
        assert max(cnt for _, cnt in spans) - min(cnt for _, cnt in spans) <= 1


def _worker(rank, world, port, n, seed, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rnd = random.Random(seed)
    tau = c.fr_to_mont([rnd.randrange(1, o.R)])[0]
    srs = c.kzg_setup(tau, n - 1)
    scalars = c.fr_to_mont([rnd.randrange(o.R) for _ in range(n)])
    first, count = sharding.shard_range(n, world, rank)
    partial = c.msm_naive(srs[first:first + count], scalars[first:first + count])
    local = torch.from_numpy(partial.view(np.int64).copy())
    gathered = sharding.gather_partials(local)
    if rank == 0:
        total = c.g1_identity()
        for r in range(world):
            total = c.g1_add(total, gathered[r].numpy().view(np.uint64))
        full = c.msm_naive(srs, scalars)
        ok = c.g1_compress(total.reshape(1, 18)) == c.g1_compress(full.reshape(1, 18))
        kat = c.g1_compress(c.g1_mul(c.g1_generator(), c.fr_eval(scalars, tau)).reshape(1, 18))
        with open(out_path, "w") as fh:
            fh.write("ok" if ok and kat == c.g1_compress(total.reshape(1, 18)) else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 37), (3, 20)])
def test_gathered_partials_sum_to_the_commitment(tmp_path, world, n):
    c.build()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(world, port, n, 77, str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"


def _bucket_worker(rank, world, port, nbuckets, seed, out_path):
    """Bucket-partial exchange with 144-byte Jacobian points standing in for the device's bucket
    format: rank r holds a full array of bucket partials; after the all_to_all it reduces slice r."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gen = c.g1_generator()

    def partials_of(r):
        rnd = random.Random(seed + r)
        ks = [rnd.randrange(o.R) if rnd.random() < 0.8 else 0 for _ in range(nbuckets)]  # some empty buckets
        return ks, np.stack([c.g1_mul(gen, c.fr_to_mont([k])[0]) for k in ks])

    _, mine = partials_of(rank)
    send = torch.from_numpy(mine.view(np.uint8).reshape(-1).copy())
    recv = torch.empty_like(send)
    sharding.exchange_bucket_slices(send, recv)
    first, count = sharding.bucket_slice(nbuckets, world, rank)
    parts = recv.numpy().view(np.uint64).reshape(world, count, 18)
    share = c.g1_identity()
    for b in range(count):
        bucket = c.g1_identity()
        for src in range(world):
            bucket = c.g1_add(bucket, parts[src, b])
        share = c.g1_add(share, c.g1_mul(bucket, c.fr_to_mont([first + b + 1])[0]))
    gathered = sharding.gather_partials(torch.from_numpy(share.view(np.int64).copy()))
    if rank == 0:
        total = c.g1_identity()
        for r in range(world):
            total = c.g1_add(total, gathered[r].numpy().view(np.uint64))
        k = 0
        for r in range(world):
            ks, _ = partials_of(r)
            k += sum((b + 1) * kb for b, kb in enumerate(ks))
        want = c.g1_mul(gen, c.fr_to_mont([k % o.R])[0])
        ok = c.g1_compress(total.reshape(1, 18)) == c.g1_compress(want.reshape(1, 18))
        with open(out_path, "w") as fh:
            fh.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nbuckets", [(2, 16), (4, 16)])
def test_bucket_partial_exchange_sums_to_the_commitment(tmp_path, world, nbuckets):
    c.build()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "result.txt"
    mp.spawn(_bucket_worker, args=(world, port, nbuckets, 91, str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"


def test_bucket_slices_need_a_power_of_two_split():
    assert sharding.bucket_slice(1 << 19, 8, 3) == (3 << 16, 1 << 16)
    for nb, world in ((16, 3), (16, 8), (12, 4)):
        with pytest.raises(ValueError):
            sharding.bucket_slice(nb, world, 0)


def _task_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = sharding.TaskExchange()
    ntasks = 7
    full = (np.arange(ntasks * 22, dtype=np.uint64).reshape(ntasks, 22) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1 << 63)
    rows = np.zeros_like(full)
    for i in range(ntasks):
        if i % world == rank:
            rows[i] = full[i]
    got = ex.exchange(rows)
    ok = np.array_equal(got, full)
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    if rank == 0:
        with open(out_path, "w") as fh:
            fh.write("ok" if all(flags) else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_task_exchange_returns_every_rank_its_rows(tmp_path, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "result.txt"
    mp.spawn(_task_worker, args=(world, port, str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"


def _poly_worker(rank, world, port, batch, log_n, out_path):
    """A batch of transforms shared by polynomial: each rank transforms its own polynomials (the oracle
    stands in for plonk_hip_fr_ntt_dev), nothing is exchanged on the data path; the results are only
    collected here to check that the ranks' shares make up the whole batch exactly once."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1 << log_n
    rnd = random.Random(5)
    polys = [c.fr_to_mont([rnd.randrange(o.R) for _ in range(n)]) for _ in range(batch)]
    mine = sharding.polynomials_of_rank(batch, world, rank)
    done = {b: c.fr_ntt(polys[b]) for b in mine}
    everyone = [None] * world
    dist.all_gather_object(everyone, {b: v.tolist() for b, v in done.items()})
    if rank == 0:
        seen = {}
        ok = True
        for share in everyone:
            for b, v in share.items():
                ok = ok and b not in seen
                seen[b] = np.array(v, dtype=np.uint64)
        ok = ok and sorted(seen) == list(range(batch))
        # against an independent evaluation: value j of a transform is f(omega^j) (the O(n^2) definition)
        ok = ok and all(c.fr_from_mont(seen[b]) == o.dft_naive(c.fr_from_mont(polys[b]), n) for b in range(batch))
        with open(out_path, "w") as fh:
            fh.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,batch", [(2, 16), (3, 16), (2, 1)])
def test_batch_of_transforms_shared_by_polynomial(tmp_path, world, batch):
    c.build()
    assert sorted(b for r in range(world) for b in sharding.polynomials_of_rank(batch, world, r)) == list(range(batch))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "result.txt"
    mp.spawn(_poly_worker, args=(world, port, batch, 5, str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"
