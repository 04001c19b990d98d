"""Sharding of the block stream over the GPUs of one node (one process per GPU, torch.distributed).

Blocks are independent samples (the 4-pixel context is baked into each 68x68 input at cut time,
Inference_QBD.py:120-129), so the path shards with no data-path collective.  The only exchange is the final
gather of the split-flag records (1344 B per block: hor[256] | ver[256] | qt[64] | dire[768]) to rank 0, which owns
the PartitionMat writer.  The reference's counterpart is nn.DataParallel's scatter/gather through GPU 0
(Inference_QBD.py:223-224); here it is one RCCL gather per (component, QP) pass ("nccl" backend = RCCL on ROCm;
"gloo" is used on CPU-only hosts and in the CPU tests).
"""
import os

import numpy as np

RECORD = 1344  # bytes per block


def env_world():
    """(rank, world, local_rank) from the torchrun environment; (0, 1, 0) when launched plainly."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")))


def shard_bounds(n, rank, world):
    """Contiguous, nearly equal split of n blocks: the first n % world ranks take one extra block."""
    if world <= 0 or not (0 <= rank < world) or n < 0:
        raise ValueError("shard_bounds: bad arguments")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_counts(n, world):
    return [shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0] for r in range(world)]


def pack_records(hor, ver, q8, d8):
    """per-block arrays -> u8[n, 1344]"""
    n = hor.shape[0]
    rec = np.empty((n, RECORD), np.uint8)
    rec[:, :256] = hor.reshape(n, 256)
    rec[:, 256:512] = ver.reshape(n, 256)
    rec[:, 512:576] = q8.reshape(n, 64)
    rec[:, 576:] = d8.reshape(n, 768).view(np.uint8)
    return rec


def unpack_records(rec):
    n = rec.shape[0]
    rec = np.ascontiguousarray(rec)
    return (rec[:, :256].reshape(n, 16, 16).copy(), rec[:, 256:512].reshape(n, 16, 16).copy(),
            rec[:, 512:576].reshape(n, 8, 8).copy(), rec[:, 576:].copy().view(np.int8).reshape(n, 3, 16, 16))


def init_process_group(device=None):
    """Join the torchrun rendezvous (no-op for world 1).  Returns (rank, world, local_rank)."""
    rank, world, local = env_world()
    if world > 1:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = os.environ.get("PMP_DIST_BACKEND") or ("nccl" if (device is not None and torch.cuda.is_available()) else "gloo")
            kw = {"device_id": device} if backend == "nccl" else {}
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def all_reduce_sum(arr, device=None):
    """Element-wise sum of an int64 numpy array over all ranks (every rank gets the result): the size table of the sharded
    emission (emit.py).  RCCL moves a device tensor; gloo a CPU tensor."""
    import torch
    import torch.distributed as dist
    a = np.ascontiguousarray(arr, np.int64)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return a
    t = torch.from_numpy(a.copy())
    if dist.get_backend() == "nccl":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def gather_records(local_rec, n_total, device=None):
    """Gather every rank's u8[n_r, 1344] records (rank order = block order) to rank 0.

    local_rec is a numpy array or a torch tensor; a CUDA tensor (what pmp_infer_postprocess_records_device wrote) is
    gathered as it is over RCCL - device to device across xGMI, no host bounce - and rank 0 copies the concatenation to the
    host once, for the file writer.  Returns u8[n_total, 1344] (numpy) on rank 0 and None elsewhere.  Shards are padded to
    the largest shard so the collective has equal counts on every rank; gloo (CPU hosts, tests) moves CPU tensors.
    """
    import torch
    import torch.distributed as dist
    is_tensor = isinstance(local_rec, torch.Tensor)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_rec.cpu().numpy() if is_tensor else np.ascontiguousarray(local_rec)
    rank, world = dist.get_rank(), dist.get_world_size()
    counts = shard_counts(n_total, world)
    if local_rec.shape[0] != counts[rank]:
        raise ValueError("gather_records: rank %d holds %d records, shard is %d" % (rank, local_rec.shape[0], counts[rank]))
    cap = max(max(counts), 1)
    nccl = dist.get_backend() == "nccl"
    if nccl:
        dev = local_rec.device if (is_tensor and local_rec.is_cuda) else device
    else:
        dev = torch.device("cpu")
    src = local_rec if is_tensor else torch.from_numpy(np.ascontiguousarray(local_rec))
    src = src.reshape(-1, RECORD)
    if src.shape[0] == cap and src.device == dev and src.is_contiguous():
        buf = src                                              # full shard already where the collective needs it
    else:
        buf = torch.zeros((cap, RECORD), dtype=torch.uint8, device=dev)
        if counts[rank]:
            buf[:counts[rank]] = src.to(dev)
    big = torch.empty((world * cap, RECORD), dtype=torch.uint8, device=dev) if rank == 0 else None
    dist.gather(buf, list(big.split(cap)) if rank == 0 else None, dst=0)   # the views are the receive slots: no copy after
    if rank != 0:
        return None
    if all(c == cap for c in counts):                          # equal shards: the receive buffer IS the result
        return big.cpu().numpy()
    return torch.cat([big[r * cap:r * cap + c] for r, c in enumerate(counts)], 0).cpu().numpy()
