"""Point-range sharding of one large commitment over the GPUs of a node.

    sum_i f_i [tau^i]G  =  sum over ranks g of  ( sum_{i in range_g} f_i [tau^i]G )

Rank g keeps SRS points [first_g, first_g + count_g) and the matching coefficients resident and
runs the whole local MSM; the only exchange is an all_gather of the 144-byte partial points
(raw bytes -- point addition is not a reduce-op, so never all_reduce), after which rank 0 adds
them (`commit_sharded`).  `commit_sharded_buckets` exchanges bucket partials instead: every rank
stops after its bucket sums, an all_to_all hands rank g everybody's partials of bucket slice g,
and rank g adds them up and reduces that slice, so the serial bucket reduction is spread over
the ranks at the price of moving the bucket arrays.  One process per GPU; `torch.distributed` supplies the collective (backend "nccl" is RCCL
over xGMI on ROCm; "gloo" on CPU tensors for tests)."""
import torch
import torch.distributed as dist

G1_LIMBS = 18


def polynomials_of_rank(batch, world, rank):
    """Sharing a batch of independent polynomials (transforms, quotients) by polynomial: polynomial b
    belongs to rank b mod world; no collective is needed on the data path."""
    return [b for b in range(batch) if b % world == rank]


# The library works on its own HIP stream, torch's collectives on torch's current stream: the two are
# ordered by events, the host does not wait in between.
_external = {}


def _library_stream(be):
    key = (be.device, be.stream)
    if key not in _external:
        _external[key] = torch.cuda.ExternalStream(be.stream, device=torch.device("cuda", be.device))
    return _external[key]


def library_then_torch(be):
    """What is queued on torch's current stream from here on runs after the library's queued work."""
    ev = torch.cuda.Event()
    ev.record(_library_stream(be))
    torch.cuda.current_stream().wait_event(ev)


def torch_then_library(be):
    """What the library queues from here on runs after the work already on torch's current stream."""
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    _library_stream(be).wait_event(ev)


def shard_range(n_total, world, rank):
    """Contiguous range of rank: (first, count); the first n_total % world ranks get one more."""
    base, extra = divmod(n_total, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def gather_partials(local_pt, gathered=None, group=None):
    """local_pt: int64 tensor (18,) holding this rank's partial point (Jacobian limbs).
    Returns an int64 tensor (world, 18), identical on every rank."""
    world = dist.get_world_size(group)
    if gathered is None:
        gathered = torch.empty(world * G1_LIMBS, dtype=local_pt.dtype, device=local_pt.device)
    if local_pt.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal on a box without RCCL peers: stage the 144-byte partials through the host
        staged = torch.empty(world * G1_LIMBS, dtype=local_pt.dtype)
        dist.all_gather_into_tensor(staged, local_pt.cpu().contiguous(), group=group)
        gathered.copy_(staged)
    else:
        dist.all_gather_into_tensor(gathered, local_pt.contiguous(), group=group)
    return gathered.view(world, G1_LIMBS)


def commit_sharded(be, d_scalars, n_local, local_pt, gathered, total_pt, group=None):
    """One sharded commitment step on a GPU rank: local MSM (library stream) -> all_gather of the
    partials -> sum on rank 0 into total_pt.  All tensors live on this rank's GPU."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1:
        # local_pt may still be read by the gather of the step before (it runs when the slowest rank arrives)
        torch_then_library(be)
    be.kzg_commit_dev(d_scalars, n_local, local_pt.data_ptr())
    if world == 1:
        return local_pt
    library_then_torch(be)  # the collective runs on torch's stream
    gather_partials(local_pt, gathered, group)
    if dist.get_rank(group) == 0:
        torch_then_library(be)
        be.g1_sum_dev(gathered.data_ptr(), world, total_pt.data_ptr())
    return total_pt


def bucket_slice(nbuckets, world, rank):
    """Bucket range of rank in the bucket-partial exchange: (first, count).  The slices must be
    equal powers of two of at least 4 buckets."""
    if world & (world - 1) or nbuckets % world or nbuckets // world < 4:
        raise ValueError("bucket exchange needs a power-of-two number of ranks and >= 4 buckets per rank")
    count = nbuckets // world
    return rank * count, count


def exchange_bucket_slices(send, recv, group=None):
    """send: uint8 tensor holding this rank's whole bucket array (world equal slices); recv: same
    size, receives slice `rank` of every rank's array, ordered by source rank."""
    if send.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal on a box without RCCL peers: stage the arrays through the host
        staged = torch.empty(send.numel(), dtype=send.dtype)
        dist.all_to_all_single(staged, send.cpu(), group=group)
        recv.copy_(staged)
    else:
        dist.all_to_all_single(recv, send, group=group)
    return recv


def commit_sharded_buckets(be, d_scalars, n_local, send, recv, local_pt, gathered, total_pt, group=None):
    """One sharded commitment step with the bucket-partial exchange: local bucket sums (library
    stream) -> all_to_all of bucket slices -> slice reduction -> all_gather of the slice shares
    -> sum on rank 0 into total_pt.  send/recv: uint8 GPU tensors of nbuckets * bytes_per_bucket
    (see Backend.msm_bucket_layout); every rank must hold equally many points."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        be.kzg_commit_dev(d_scalars, n_local, local_pt.data_ptr())
        return local_pt
    rank = dist.get_rank(group)
    nbuckets, _ = be.msm_bucket_layout()
    first, count = bucket_slice(nbuckets, world, rank)
    torch_then_library(be)  # send / local_pt may still be read by the exchanges of the step before
    be.msm_buckets_dev(d_scalars, n_local, send.data_ptr())
    library_then_torch(be)  # the collectives run on torch's stream
    exchange_bucket_slices(send, recv, group)
    torch_then_library(be)
    be.msm_reduce_bucket_slice_dev(recv.data_ptr(), world, first, count, local_pt.data_ptr())
    library_then_torch(be)
    gather_partials(local_pt, gathered, group)
    if rank == 0:
        torch_then_library(be)
        be.g1_sum_dev(gathered.data_ptr(), world, total_pt.data_ptr())
    return total_pt


class TaskExchange:
    """Sharing by polynomial: independent commitments / openings are dealt out to the ranks and every
    rank needs all results.  `exchange(rows, owner)` takes this rank's (tasks, k) uint64 array with the
    rows of its own tasks filled in and the owner rank of every task (default: task i to rank
    i mod world) and returns the array with every row filled in: an all_gather of the raw arrays
    (opaque limbs, never a reduction), rows picked by owner."""

    def __init__(self, device=None, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device

    def exchange(self, rows, owner=None):
        import numpy as np
        mine = torch.from_numpy(np.ascontiguousarray(rows).view(np.int64))
        staged = dist.get_backend(self.group) == "gloo" or self.device is None
        if not staged:
            mine = mine.to(self.device)
        everyone = torch.empty((self.world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(everyone.view(-1), mine.contiguous().view(-1), group=self.group)
        everyone = everyone.cpu().numpy().view(np.uint64)
        out = np.empty_like(rows)
        for i in range(rows.shape[0]):
            out[i] = everyone[owner[i] if owner is not None else i % self.world, i]
        return out
