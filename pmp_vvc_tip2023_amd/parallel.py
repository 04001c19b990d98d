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


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


INIT_TIMEOUT_S = 180           # rendezvous + preflight: a rank that never arrives becomes an error, not a hang
COLLECTIVE_TIMEOUT_S = 1800    # steady state: a rank slowed by disk, a long first touch or a range-guard re-run is late, not missing


def init_process_group(device=None, timeout_s=None, force=False):
    """Join the torchrun rendezvous (no-op for world 1 unless `force` or PMP_DIST_FORCE=1: a one-rank group, which is how the
    single-GPU test box executes the RCCL code path).  Returns (rank, world, local_rank).  The short timeout (PMP_DIST_TIMEOUT_S)
    bounds the rendezvous and the preflight only: relax_timeout() raises it for the job's own collectives."""
    rank, world, local = env_world()
    force = force or os.environ.get("PMP_DIST_FORCE") == "1"
    if world > 1 or force:
        import datetime
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                if world > 1:
                    raise RuntimeError("init_process_group: WORLD_SIZE=%d without MASTER_PORT - start the ranks with torch.distributed.run "
                                       "or parallel.spawn_ranks" % world)
                os.environ["MASTER_PORT"] = str(_free_port())     # a forced one-rank group: any free port, so two such jobs never collide
            backend = os.environ.get("PMP_DIST_BACKEND") or ("nccl" if (device is not None and torch.cuda.is_available()) else "gloo")
            kw = {"device_id": device} if backend == "nccl" else {}
            t = float(os.environ.get("PMP_DIST_TIMEOUT_S", timeout_s or INIT_TIMEOUT_S))
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=t), **kw)
    return rank, world, local


def relax_timeout():
    """After the preflight: the job's own collectives (size all-reduces, the gather, closing barriers) get the steady-state timeout
    (PMP_DIST_COLLECTIVE_TIMEOUT_S, default 1800 s).  A rank that DIES is noticed by the launcher at once (spawn_ranks polls every
    rank), so the long bound only ever covers ranks that are slow.  Returns the seconds in force (None without a process group)."""
    import datetime
    import torch.distributed as dist
    if not dist.is_initialized():
        return None
    t = float(os.environ.get("PMP_DIST_COLLECTIVE_TIMEOUT_S", COLLECTIVE_TIMEOUT_S))
    try:
        dist.distributed_c10d._set_pg_timeout(datetime.timedelta(seconds=t), None)     # RCCL and gloo process groups both honour it
    except Exception as e:                                     # noqa: BLE001 - a torch without it: the init timeout stays in force
        t0 = float(os.environ.get("PMP_DIST_TIMEOUT_S", INIT_TIMEOUT_S))
        import sys
        print("pmp parallel: this torch cannot raise the process group's timeout (%s: %s); collectives keep the rendezvous bound of %g s - "
              "set PMP_DIST_TIMEOUT_S to what the slowest pass needs" % (type(e).__name__, e, t0), file=sys.stderr)
        return t0
    return t


def preflight(device=None):
    """First collective of a multi-rank job, with its content verified: every rank contributes (rank + 1) * [1, 2, 3, 4] to a
    gather on rank 0 and a sum over all ranks.  Turns a broken RCCL / IPC setup (e.g. HSA_ENABLE_IPC_MODE_LEGACY unset) or a
    duplicate-GPU placement into a readable error before any real work.  Returns {"ranks", "backend", "ms"}."""
    import time
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return {"ranks": 1, "backend": None, "ms": 0.0}
    rank, world, backend = dist.get_rank(), dist.get_world_size(), dist.get_backend()
    dev = device if backend == "nccl" else torch.device("cpu")
    t0 = time.perf_counter()
    mine = (torch.arange(1, 5, dtype=torch.int64) * (rank + 1)).to(dev)
    slots = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(world)] if rank == 0 else None
    try:
        dist.gather(mine, slots, dst=0)
        total = mine.clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
        if backend == "nccl":
            torch.cuda.synchronize(dev)
    except Exception as e:      # noqa: BLE001 - whatever the backend raises, the message must name the likely causes
        raise RuntimeError("rank %d/%d: the first %s collective failed (%s: %s).  Check: one GPU per rank, HSA_ENABLE_IPC_MODE_LEGACY=0 "
                           "in the ranks' environment (dmabuf IPC), MASTER_ADDR=127.0.0.1" % (rank, world, backend, type(e).__name__, e)) from e
    want = torch.arange(1, 5, dtype=torch.int64) * (world * (world + 1) // 2)
    if not torch.equal(total.cpu(), want):
        raise RuntimeError("rank %d/%d: %s all_reduce returned %s, expected %s" % (rank, world, backend, total.cpu().tolist(), want.tolist()))
    if rank == 0:
        for r, t in enumerate(slots):
            if not torch.equal(t.cpu(), torch.arange(1, 5, dtype=torch.int64) * (r + 1)):
                raise RuntimeError("rank 0: %s gather slot %d holds %s" % (backend, r, t.cpu().tolist()))
    return {"ranks": world, "backend": backend, "ms": (time.perf_counter() - t0) * 1e3}


def spawn_ranks(cmd, n, env_extra=None, capture_rank0=False, poll_s=0.05):
    """Start n fresh rank processes of `cmd` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; the caller has made no GPU call -
    a process that has initialised HIP must never be forked or re-executed) and watch ALL of them: the first rank that exits
    non-zero ends the job - the others, which would sit in a collective until its timeout, are killed by PID - and its code is
    returned.  Rank 0 inherits this process's stdout (the reference driver prints its progress there), the other ranks' stdout goes
    to stderr; capture_rank0: rank 0's stdout is collected instead (on a thread, so a full pipe never blocks it) and returned.
    Returns (exit code, rank-0 stdout bytes or None)."""
    import subprocess
    import sys
    import threading
    import time
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
    env.update(env_extra or {})
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen(list(cmd), env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=(subprocess.PIPE if capture_rank0 else sys.stdout) if r == 0 else sys.stderr, stderr=sys.stderr))
    chunks = []
    reader = None
    if capture_rank0:
        reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
        reader.start()
    rc = 0
    try:
        live = set(range(n))
        while live and rc == 0:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0:
                    rc = code
                    print("rank %d exited with code %d: stopping the other ranks" % (r, code), file=sys.stderr, flush=True)
                    break
            if live and rc == 0:
                time.sleep(poll_s)
    except BaseException:
        rc = rc or 1
        raise
    finally:
        for p in procs:                                        # by PID, never by pattern
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=30)
            except Exception:                                  # noqa: BLE001
                pass
        if reader is not None:
            reader.join(timeout=30)
    return rc, (b"".join(c for c in chunks if c) if capture_rank0 else None)


def all_reduce_sum(arr, device=None):
    """Element-wise sum of an int64 numpy array over all ranks (every rank gets the result): the size table of the sharded
    emission (emit.py).  RCCL moves a device tensor; gloo a CPU tensor."""
    import torch
    import torch.distributed as dist
    a = np.ascontiguousarray(arr, np.int64)
    if not dist.is_initialized():                              # (a forced one-rank group does run the collective)
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
    if not dist.is_initialized():                              # (a forced one-rank group does run the collective)
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
