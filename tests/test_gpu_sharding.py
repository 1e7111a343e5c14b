"""Two ranks sharing this box's GPU (gloo rendezvous, collectives staged through the host) run the
sharded commitment both ways -- all_gather of reduced partial points, all_to_all of bucket partials
-- through the C ABI; rank 0 checks both against the known answer [f(tau)]G from the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, log_n, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import plonk_amd
    from plonk_amd import sharding
    from oracle import capi as c

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    be = plonk_amd.Backend(device=0)
    n = 1 << log_n
    first, count = sharding.shard_range(n, world, rank)
    tau = c.fr_to_mont([0x5EED + (1 << 180)])[0]
    be.kzg_setup_range(tau, first, count)
    rng = np.random.default_rng(7)  # the same stream on every rank: rank 0 needs all of it
    sc = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    d_sc = be.to_device(sc[first:first + count])
    local_pt = torch.zeros(18, dtype=torch.int64, device=dev)
    gathered = torch.zeros(18 * world, dtype=torch.int64, device=dev)
    total_pt = torch.zeros(18, dtype=torch.int64, device=dev)
    nb, each = be.msm_bucket_layout()
    send = torch.empty(nb * each, dtype=torch.uint8, device=dev)
    recv = torch.empty(nb * each, dtype=torch.uint8, device=dev)
    results = []
    for mode in ("points", "buckets", "buckets"):  # twice: buffers are reused between steps
        total_pt.zero_()
        if mode == "points":
            sharding.commit_sharded(be, d_sc, count, local_pt, gathered, total_pt)
        else:
            sharding.commit_sharded_buckets(be, d_sc, count, send, recv, local_pt, gathered, total_pt)
        be.sync()
        results.append(total_pt.cpu().numpy().view(np.uint64).copy())
    if rank == 0:
        want = c.g1_compress(c.g1_mul(c.g1_generator(), c.fr_eval(sc, tau)).reshape(1, 18))
        ok = all(c.g1_compress(r.reshape(1, 18)) == want for r in results)
        with open(out_path, "w") as fh:
            fh.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()
    be.close()


@pytest.mark.parametrize("log_n", [12, 19])
def test_two_ranks_both_exchanges(tmp_path, log_n):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, port, log_n, str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"


def _prover_worker(rank, world, port, log_gates, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import plonk_amd
    from plonk_amd import prover as pv
    from plonk_amd import sharding

    torch.cuda.set_device(0)
    be = plonk_amd.Backend(device=0)
    circuit = pv.synthetic_circuit(log_gates, 3)
    p = pv.Prover(be, circuit.d)
    tau = 0xABCDEF0123456789
    p.setup(tau)
    keys = p.keys(circuit)
    alone = p.prove_resident(circuit, keys)
    p.kzg_log = []
    shared = p.prove_resident(circuit, keys, ranks=sharding.TaskExchange(device=torch.device("cuda", 0)))
    mine = p.audit_with_tau(tau)  # this rank's share of the 18 commitments and openings, plus T and v
    ok = shared == alone and p.check_identities((circuit.number_gates, circuit.pub_inputs), shared, keys)
    counts = [None] * world
    dist.all_gather_object(counts, (mine, bool(ok)))
    if rank == 0:
        total = sum(c for c, _ in counts) - 2 * (world - 1)  # every rank commits to T and v
        with open(out_path, "w") as fh:
            fh.write("ok" if all(good for _, good in counts) and total == 20 else "mismatch %r" % (counts,))
    dist.barrier()
    dist.destroy_process_group()
    be.close()


@pytest.mark.parametrize("world,log_gates", [(2, 6), (3, 10), (4, 7)])
def test_proof_shared_by_polynomial_between_ranks(tmp_path, world, log_gates):
    """The 18 independent commitments and openings of a proof dealt out to the ranks (sharing by
    polynomial), and with them the building of the polynomials they read (with four ranks the gate quotient
    is built by two of them and the running product / wiring quotient by the other two): the gathered proof
    equals the single-rank proof."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "result.txt"
    mp.spawn(_prover_worker, args=(world, port, log_gates, str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"
