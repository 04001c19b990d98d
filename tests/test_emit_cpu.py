"""Sharded emission (pmp_vvc_tip2023_amd/emit.py): every rank formats and pwrites its own block rows; the file must be byte-identical
to the single-writer one (a19, Map2Partition.py:385-412, pinned by G5/G7 through pmp_write_partition_file).  CPU only: the
formatter and the offset exchange are host code; world 4 runs over gloo."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT, golden, golden_path
from pmp_vvc_tip2023_amd import emit, engine as E, parallel


def _records(n, seed, wild=False):
    rng = np.random.default_rng(seed)
    if wild:     # arbitrary caller data: multi-digit and negative values take the slow formatter paths
        hor = rng.integers(0, 256, (n, 16, 16)).astype(np.uint8); ver = rng.integers(0, 120, (n, 16, 16)).astype(np.uint8)
        q8 = rng.integers(0, 256, (n, 8, 8)).astype(np.uint8); d8 = rng.integers(-128, 128, (n, 3, 16, 16)).astype(np.int8)
    else:        # the path's own value ranges
        hor = rng.integers(0, 2, (n, 16, 16)).astype(np.uint8); ver = rng.integers(0, 2, (n, 16, 16)).astype(np.uint8)
        q8 = rng.integers(0, 4, (n, 8, 8)).astype(np.uint8); d8 = rng.integers(-1, 2, (n, 3, 16, 16)).astype(np.int8)
    return parallel.pack_records(hor, ver, q8, d8), (hor, ver, q8, d8)


def test_pieces_and_offsets():
    assert emit.row_pieces(2, 9, 3) == [(0, 2, 3), (1, 0, 3), (2, 0, 3)]
    assert emit.row_pieces(1, 8, 4, max_rows=2) == [(0, 1, 3), (0, 3, 4), (1, 0, 2), (1, 2, 4)]
    assert emit.row_pieces(5, 5, 3) == []
    for frames, bh, world in ((3, 3, 4), (1, 2, 4), (8, 33, 8), (9, 17, 8)):
        cover = []
        for r in range(world):
            lo, hi = emit.shard_rows(frames, bh, r, world)
            cover += list(range(lo, hi))
        assert cover == list(range(frames * bh))
    sizes = np.arange(2 * 3 * 6, dtype=np.int64).reshape(6, 6) + 1      # frames 2, bh 3
    offs, total = emit.section_offsets(sizes, 2, 3)
    assert total == sizes.sum() and offs[0, 0, 0] == 0
    assert offs[0, 0, 1] == sizes[0, 0] and offs[0, 1, 0] == sizes[:3, 0].sum()
    assert offs[1, 0, 0] == sizes[:3].sum() and offs[1, 5, 2] == total - sizes[5, 5]


def test_rows_formatter_is_the_frame_formatter_on_a_slice():
    """pmp_format_partition_rows[_records] of rows [a, b) == the text of a frame of height 64 (b - a) made of those blocks."""
    W, bh = 200, 5
    bw = W // 64
    rec, (hor, ver, q8, d8) = _records(bh * bw, 5, wild=True)
    for a, b in ((0, 5), (1, 3), (4, 5), (2, 2)):
        sl = slice(a * bw, b * bw)
        buf, sizes = E.format_partition_rows_records(W, b - a, rec[sl])
        want = E.format_partition_text(1, 64 * (b - a), W, hor[sl], ver[sl], q8[sl], d8[sl])
        assert buf.raw[:int(sizes.sum())] == want and sizes.shape == (b - a, 6)
        for r in range(a, b):      # per-row sizes: the row on its own
            one = E.format_partition_rows_records(W, 1, rec[r * bw:(r + 1) * bw])[1]
            assert np.array_equal(one[0], sizes[r - a])


@pytest.mark.parametrize("threads", [1, 3])
@pytest.mark.parametrize("wild", [False, True])
def test_one_rank_file_bytes(tmp_path, threads, wild):
    F, H, W = 3, 200, 136                                   # ragged: right/bottom remainders dropped, bh = 3, bw = 2
    bh, bw = H // 64, W // 64
    rec, (hor, ver, q8, d8) = _records(F * bh * bw, 11, wild)
    ref = tmp_path / "ref.txt"
    E.write_partition_file(str(ref), F, H, W, hor, ver, q8, d8)
    out = tmp_path / "out_PartitionMat.txt"
    out.write_bytes(b"x" * (2 * os.path.getsize(ref)))      # a longer stale file must be cut
    em = emit.ShardEmitter(threads=threads)
    p = em.start(str(out), F, H, W, 0, F * bh, rec, binary=True)
    total = em.finish(p)
    em.close()
    assert total == os.path.getsize(ref) and out.read_bytes() == ref.read_bytes() and em.bytes_written == total
    refb = tmp_path / "ref.pmpb"
    E.write_partition_binary(str(refb), F, H, W, hor, ver, q8, d8)
    assert (tmp_path / "out_PartitionMat.pmpb").read_bytes() == refb.read_bytes()


def test_golden_file_through_the_sharded_writer(tmp_path):
    """G5 (reference-generated text): per-block arrays -> records -> sharded writer == the reference's bytes."""
    g = golden("g5_seq_Luma.npz")
    F, W, H = int(g["F"]), int(g["W"]), int(g["H"])
    from oracle import postproc as P
    hor, ver, q8, d8 = P.seq_post_process(g["qt"], g["bt"], g["dire"], "Luma", F, W, H, None)
    rec = parallel.pack_records(hor, ver, q8.astype(np.uint8), d8)
    out = tmp_path / "g5.txt"
    em = emit.ShardEmitter(threads=2)
    em.finish(em.start(str(out), F, H, W, 0, F * (H // 64), rec))
    em.close()
    assert out.read_bytes() == open(golden_path("g5_partitionmat_Luma.txt"), "rb").read()


WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %(root)r)
    import numpy as np
    from pmp_vvc_tip2023_amd import emit, parallel
    rank, world, _ = parallel.init_process_group(None)
    for tag, F, H, W in (("a", 3, 200, 136), ("b", 1, 130, 64), ("c", 5, 64, 320)):   # 9, 2 and 5 block rows over 4 ranks
        rec = np.load(os.path.join(%(dir)r, tag + ".npy"))
        bh, bw = H // 64, W // 64
        lo, hi = emit.shard_rows(F, bh, rank, world)
        em = emit.ShardEmitter(rank, world, threads=2)
        mine = np.ascontiguousarray(rec[lo * bw:hi * bw])
        p = em.start(os.path.join(%(dir)r, tag + "_PartitionMat.txt"), F, H, W, lo, hi, mine, binary=True)
        em.finish(p)
        em.close()
    import torch.distributed as dist
    dist.barrier(); dist.destroy_process_group()
''')


def test_four_ranks_gloo_ragged_frames_equal_one_writer(tmp_path):
    cases = {"a": (3, 200, 136), "b": (1, 130, 64), "c": (5, 64, 320)}
    refs = {}
    for i, (tag, (F, H, W)) in enumerate(cases.items()):
        rec, (hor, ver, q8, d8) = _records(F * (H // 64) * (W // 64), 20 + i, wild=(tag == "c"))
        np.save(tmp_path / (tag + ".npy"), rec)
        E.write_partition_file(str(tmp_path / (tag + "_ref.txt")), F, H, W, hor, ver, q8, d8)
        E.write_partition_binary(str(tmp_path / (tag + "_ref.pmpb")), F, H, W, hor, ver, q8, d8)
        (tmp_path / (tag + "_PartitionMat.txt")).write_bytes(b"stale " * 100000)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "dir": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="4", LOCAL_WORLD_SIZE="4", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(4)]
    logs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\\n".join(logs)
    for tag in cases:
        assert (tmp_path / (tag + "_PartitionMat.txt")).read_bytes() == (tmp_path / (tag + "_ref.txt")).read_bytes(), tag
        assert (tmp_path / (tag + "_PartitionMat.pmpb")).read_bytes() == (tmp_path / (tag + "_ref.pmpb")).read_bytes(), tag


def test_unfinished_job_leaves_a_part_file_not_a_holey_final_file(tmp_path):
    """a rank that dies between finish() and drain() must not leave a full-size PartitionMat.txt with NUL holes under the final
    name.  Writes go to <name>.part; the final name appears in drain(), behind every rank's writes - and an older complete file stays
    intact until then."""
    F, H, W = 2, 128, 128
    bh, bw = H // 64, W // 64
    rec, (hor, ver, q8, d8) = _records(F * bh * bw, 31)
    out = tmp_path / "seq_Luma_QP22_PartitionMat.txt"
    out.write_bytes(b"an older, complete file\n")
    em = emit.ShardEmitter(threads=1)
    p = em.start(str(out), F, H, W, 0, F * bh, rec, binary=True)
    em.finish(p)
    for w in em.writes:                                      # the writes themselves are done - the job "dies" before drain()
        w.result()
    assert out.read_bytes() == b"an older, complete file\n"
    assert os.path.exists(str(out) + ".part") and not os.path.exists(str(tmp_path / "seq_Luma_QP22_PartitionMat.pmpb"))
    em.close()                                               # the surviving path: drain() renames
    ref = tmp_path / "ref.txt"
    E.write_partition_file(str(ref), F, H, W, hor, ver, q8, d8)
    assert out.read_bytes() == ref.read_bytes() and not os.path.exists(str(out) + ".part")
    assert os.path.exists(str(tmp_path / "seq_Luma_QP22_PartitionMat.pmpb")) and not os.path.exists(str(tmp_path / "seq_Luma_QP22_PartitionMat.pmpb.part"))
