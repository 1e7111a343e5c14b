"""Point-range sharding of one large commitment over the GPUs of a node.

    sum_i f_i [tau^i]G  =  sum over ranks g of  ( sum_{i in range_g} f_i [tau^i]G )

Rank g keeps SRS points [first_g, first_g + count_g) and the matching coefficients resident and
runs the whole local MSM; the only exchange is an all_gather of the 144-byte partial points
(raw bytes -- point addition is not a reduce-op, so never all_reduce), after which rank 0 adds
them.  One process per GPU; `torch.distributed` supplies the collective (backend "nccl" is RCCL
over xGMI on ROCm; "gloo" on CPU tensors for tests)."""
import torch
import torch.distributed as dist

G1_LIMBS = 18


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
    be.kzg_commit_dev(d_scalars, n_local, local_pt.data_ptr())
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_pt
    be.sync()  # the collective runs on torch's stream
    gather_partials(local_pt, gathered, group)
    torch.cuda.current_stream().synchronize()
    if dist.get_rank(group) == 0:
        be.g1_sum_dev(gathered.data_ptr(), world, total_pt.data_ptr())
    return total_pt
