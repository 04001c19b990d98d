"""CPU, world_size 2 over gloo: the N>1 path (shard -> per-rank post-processing -> gather to rank 0 -> file).
The per-rank compute is stood in by the oracle (the HIP path needs a GPU); what is under test is the sharding,
record packing, the gather collective and that rank 0's file equals the single-process file."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT, golden, golden_path
from pmp_vvc_tip2023_amd import parallel


def test_shard_bounds_cover_and_order():
    for n in (0, 1, 7, 480, 1980, 1024):
        for world in (1, 2, 3, 8):
            prev = 0
            for r in range(world):
                lo, hi = parallel.shard_bounds(n, r, world)
                assert lo == prev and hi >= lo
                prev = hi
            assert prev == n
            c = parallel.shard_counts(n, world)
            assert sum(c) == n and max(c) - min(c) <= 1
    with pytest.raises(ValueError):
        parallel.shard_bounds(4, 2, 2)


def test_record_round_trip():
    rng = np.random.default_rng(0)
    hor = rng.integers(0, 2, (5, 16, 16)).astype(np.uint8); ver = rng.integers(0, 2, (5, 16, 16)).astype(np.uint8)
    q8 = rng.integers(0, 4, (5, 8, 8)).astype(np.uint8); d8 = rng.integers(-1, 2, (5, 3, 16, 16)).astype(np.int8)
    rec = parallel.pack_records(hor, ver, q8, d8)
    assert rec.shape == (5, 1344)
    h, v, q, d = parallel.unpack_records(rec)
    assert np.array_equal(h, hor) and np.array_equal(v, ver) and np.array_equal(q, q8) and np.array_equal(d, d8)


WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %(root)r)
    import numpy as np
    from pmp_vvc_tip2023_amd import parallel, engine
    from oracle import postproc as P
    rank, world, _ = parallel.init_process_group(None)
    g = np.load(%(npz)r)
    qt, bt, dire = g["qt"], g["bt"], g["dire"]
    n = qt.shape[0]
    lo, hi = parallel.shard_bounds(n, rank, world)
    hor, ver, q, d = P.seq_post_process(qt[lo:hi], bt[lo:hi], dire[lo:hi], "Luma", 1, 64 * (hi - lo), 64, None)
    rec = parallel.gather_records(parallel.pack_records(hor, ver, q.astype(np.uint8), d), n)
    if rank == 0:
        h, v, q8, d8 = parallel.unpack_records(rec)
        engine.write_partition_file(%(out)r, int(g["F"]), int(g["H"]), int(g["W"]), h, v, q8, d8)
    else:
        assert rec is None
    import torch.distributed as dist
    dist.barrier(); dist.destroy_process_group()
''')


def test_two_rank_gloo_gather_matches_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "out.txt")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "npz": golden_path("g5_seq_Luma.npz"), "out": out})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\\n".join(logs)
    assert open(out, "rb").read() == open(golden_path("g5_partitionmat_Luma.txt"), "rb").read()


WORKER_T = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %(root)r)
    import numpy as np, torch
    from pmp_vvc_tip2023_amd import parallel
    rank, world, _ = parallel.init_process_group(None)
    for n_total in (5, 6, 1, 0):                       # unequal shards, equal shards, fewer records than ranks, none
        allrec = (np.arange(n_total * parallel.RECORD, dtype=np.int64) %% 251).astype(np.uint8).reshape(n_total, parallel.RECORD)
        lo, hi = parallel.shard_bounds(n_total, rank, world)
        for as_tensor in (True, False):
            local = torch.from_numpy(allrec[lo:hi].copy()) if as_tensor else allrec[lo:hi].copy()
            got = parallel.gather_records(local, n_total)
            if rank == 0:
                assert isinstance(got, np.ndarray) and got.shape == (n_total, parallel.RECORD) and np.array_equal(got, allrec), n_total
            else:
                assert got is None
    try:
        parallel.gather_records(np.zeros((3, parallel.RECORD), np.uint8), 2)    # wrong shard size is refused on every rank
        raise SystemExit("shard-size check missing")
    except ValueError:
        pass
    import torch.distributed as dist
    dist.barrier(); dist.destroy_process_group()
''')


def test_two_rank_gather_accepts_tensors_and_ragged_shards(tmp_path):
    """gather_records takes what the device path hands it (a torch tensor of packed records) as well as numpy arrays; shards
    may be unequal or empty.  On the GPU node the same code moves CUDA tensors over RCCL; here gloo moves CPU tensors."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker_t.py"
    script.write_text(WORKER_T % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\\n".join(logs)


def test_bench_self_launch_propagates_rank_failure():
    """`python bench.py --gpus 2` starts its own rank processes; on a box without a GPU every rank refuses to run (there is no
    CPU fallback) and the parent must exit non-zero without printing a result line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_parity.py::test_bench_launches_its_own_ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and not r.stdout.strip()
    # every rank refuses; the launcher stops the others as soon as the first one has (it may be gone before it could say so)
    assert 1 <= r.stderr.count("no CPU fallback") <= 2 and "stopping the other ranks" in r.stderr


def test_spawn_ranks_stops_everyone_when_one_rank_dies(tmp_path):
    """a rank other than 0 that dies first must end the job at once - rank 0 would otherwise sit in its first
    collective until the timeout.  spawn_ranks polls ALL children: here rank 1 exits with code 7 immediately while rank 0 sleeps
    for ten minutes; the call returns 7 within seconds and rank 0 is gone (killed by PID)."""
    import time
    script = tmp_path / "w.py"
    pidfile = tmp_path / "pid0"
    script.write_text(textwrap.dedent('''
        import os, sys, time
        if os.environ["RANK"] == "1":
            while not os.path.exists(%r):          # die only once rank 0 is up and has said its line
                time.sleep(0.02)
            sys.exit(7)
        print("rank 0 line", flush=True)
        with open(%r + ".tmp", "w") as f:
            f.write(str(os.getpid()))
        os.rename(%r + ".tmp", %r)
        time.sleep(600)
    ''' % (str(pidfile), str(pidfile), str(pidfile), str(pidfile))))
    t0 = time.time()
    rc, out = parallel.spawn_ranks([sys.executable, str(script)], 2, capture_rank0=True)
    assert rc == 7 and time.time() - t0 < 30
    assert out is not None and b"rank 0 line" in out
    pid = int(pidfile.read_text())
    time.sleep(0.2)
    with pytest.raises(ProcessLookupError):
        os.kill(pid, 0)
    # all ranks fine -> 0, rank 0's stdout relayed whole
    ok = tmp_path / "ok.py"
    ok.write_text("import os\nprint('r' + os.environ['RANK'] + ' of ' + os.environ['WORLD_SIZE'] + ' ' + os.environ['MASTER_ADDR'])\n")
    rc, out = parallel.spawn_ranks([sys.executable, str(ok)], 3, capture_rank0=True)
    assert rc == 0 and out.strip() == b"r0 of 3 127.0.0.1"


WORKER_P = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %(root)r)
    from pmp_vvc_tip2023_amd import parallel
    import numpy as np
    rank, world, _ = parallel.init_process_group(None, timeout_s=60)
    info = parallel.preflight(None)
    assert info["ranks"] == world == 2 and info["backend"] == "gloo" and info["ms"] > 0, info
    tot = parallel.all_reduce_sum(np.arange(12, dtype=np.int64).reshape(2, 6) * (rank + 1))
    assert np.array_equal(tot, np.arange(12).reshape(2, 6) * 3), tot
    import torch.distributed as dist
    dist.barrier(); dist.destroy_process_group()
''')


def test_preflight_and_size_table_reduce_over_gloo(tmp_path):
    """The first collective of every multi-rank job (parallel.preflight: verified gather + all-reduce) and the int64 table reduce of
    the sharded emission, world 2 over gloo."""
    script = tmp_path / "worker_p.py"
    script.write_text(WORKER_P % {"root": ROOT})
    rc, _ = parallel.spawn_ranks([sys.executable, str(script)], 2, env_extra={"OMP_NUM_THREADS": "1"})
    assert rc == 0


ONE_RANK = r"""
import sys
import numpy as np
import torch
import torch.distributed as dist
from pmp_vvc_tip2023_amd import parallel
rank, world, local = parallel.init_process_group(None, force=True)     # a one-rank group: the collectives below really run
assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "gloo"
info = parallel.preflight(None)
assert info["ranks"] == 1 and info["backend"] == "gloo"
assert parallel.relax_timeout() == 1800.0
rec = np.random.default_rng(0).integers(0, 255, (9, 1344)).astype(np.uint8)
assert np.array_equal(parallel.gather_records(rec, 9), rec)
assert np.array_equal(parallel.gather_records(torch.from_numpy(rec), 9), rec)
t = np.arange(12, dtype=np.int64).reshape(2, 6)
assert np.array_equal(parallel.all_reduce_sum(t), t)
dist.destroy_process_group()
assert parallel.preflight(None) == {"ranks": 1, "backend": None, "ms": 0.0} and parallel.relax_timeout() is None
print("one-rank ok")
"""


def test_forced_one_rank_group_runs_the_collectives(tmp_path):
    """PMP_DIST_FORCE / force=True: with a process group present the helpers take the collective path even at world size 1 - the same
    switch the GPU box uses to run the RCCL branch on its single GPU (tests/test_gpu_scale.py)."""
    import subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        env["MASTER_PORT"] = str(s.getsockname()[1])
    env.update(PYTHONPATH=ROOT, PMP_DIST_BACKEND="gloo", PMP_DIST_COLLECTIVE_TIMEOUT_S="1800")
    r = subprocess.run([sys.executable, "-c", ONE_RANK], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "one-rank ok" in r.stdout, r.stdout + r.stderr


def test_spawn_ranks_rank0_prints_on_the_launchers_stdout(tmp_path):
    """the self-launched driver's progress lines (rank 0's 'Save:', 'Sum time') belong on stdout, as the reference prints them;
    the other ranks' stdout goes to stderr."""
    import subprocess, sys
    prog = ("import sys; sys.path.insert(0, %r)\n"
            "from pmp_vvc_tip2023_amd import parallel\n"
            "rc, out = parallel.spawn_ranks([sys.executable, '-c', 'import os; print(\"hello from rank\", os.environ[\"RANK\"])'], 2)\n"
            "assert out is None\nsys.exit(rc)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == "hello from rank 0" and "hello from rank 1" in r.stderr
