"""Sharded PartitionMat emission: every rank formats and writes the text of its own block rows; nobody funnels through rank 0.

The reference's serial tail is the per-value text emission of get_sequence_partition_for_VTM (Map2Partition.py:401-412), one
process, one `write` per value.  Frames are self-contained in the file (Map2Partition.py:389-412) and inside a frame each of the
six sections (hor, ver, qt, dire 0..2) is row-major, so a run of whole BLOCK ROWS of one frame is six contiguous byte ranges.
Line lengths vary ("-1\\n" vs "0\\n"), so the ranks exchange the exact byte count of every (block row, section) pair - one
all-reduce of an int64[frames * H/64, 6] table per file, a few KB - derive every range's offset with an exclusive scan in file
order, and write concurrently with pwrite.  The bytes are identical to the single-writer file (a19): tests/test_emit_cpu.py.
Every rank writes into `<name>.part`; rank 0 renames it to the final name in drain(), after a barrier behind every rank's writes: a
job that dies half-way (the launcher kills the peers of a failed rank) leaves a `.part` file, never a full-size PartitionMat.txt with
holes that VTM's parser would read as data.

The unit of sharding is the block row (H/64 per frame), not the frame: real jobs have fewer sub-sampled frames than GPUs
(ssRatio 30), and 9 frames on 8 ranks would otherwise leave one rank with twice the work.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import engine as E
from . import parallel

RECORD = parallel.RECORD


def shard_rows(frames, bh, rank, world):
    """Block rows [g_lo, g_hi) of rank `rank`; global index g = frame * bh + row."""
    return parallel.shard_bounds(int(frames) * int(bh), rank, world)


def row_pieces(g_lo, g_hi, bh, max_rows=0):
    """Block rows [g_lo, g_hi) as runs inside one frame, each at most max_rows long (0: no limit): [(frame, a, b)]."""
    out, g = [], g_lo
    while g < g_hi:
        f, a = divmod(g, bh)
        b = min(bh, a + (g_hi - g))
        if max_rows:
            b = min(b, a + max_rows)
        out.append((f, a, b))
        g += b - a
    return out


def section_offsets(sizes_all, frames, bh):
    """int64[frames * bh, 6] byte counts -> (int64[frames, 6, bh] file offset of every (frame, section, block row), file size).
    File order: frame, then section, then row (Map2Partition.py:401-412)."""
    sizes_all = np.asarray(sizes_all, np.int64).reshape(frames, bh, 6)
    flat = sizes_all.transpose(0, 2, 1).reshape(-1)
    cs = np.cumsum(flat)
    return (cs - flat).reshape(frames, 6, bh), (int(cs[-1]) if flat.size else 0)


def _pwrite_all(fd, view, off):
    while len(view):
        k = os.pwrite(fd, view, off)
        view = view[k:]
        off += k


class _Pending:
    __slots__ = ("path", "frames", "height", "width", "bh", "bw", "pieces", "futs", "binary", "rec", "g_lo")


class ShardEmitter:
    """One per rank.  start() hands a pass's records to the formatter threads; finish() - called from the main thread, in the same
    order on every rank - exchanges the size table and queues the writes; drain() waits for them."""

    def __init__(self, rank=0, world=1, threads=0, device=None):
        self.rank, self.world, self.device = rank, world, device
        if threads <= 0:
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)) or 1)
            threads = max(2, min(8, (os.cpu_count() or 2) // max(local_world, 1)))
        self.threads = threads
        self.pool = ThreadPoolExecutor(max_workers=threads)
        self.writes = []
        self.renames = []            # rank 0: (part, final) of the files whose writes are queued
        self.bytes_written = 0

    # ---------------------------------------------------------------------------------------------------------------- phases
    def start(self, path, frames, height, width, g_lo, g_hi, rec, binary=False):
        """rec: u8[(g_hi - g_lo) * (width // 64), 1344] host records of this rank's block rows, in block order.  The array must stay
        untouched until finish() has returned for this pass."""
        p = _Pending()
        p.path, p.frames, p.height, p.width, p.binary, p.rec, p.g_lo = path, int(frames), int(height), int(width), binary, rec, g_lo
        p.bh, p.bw = p.height // 64, p.width // 64
        n_rows = g_hi - g_lo
        # enough pieces to keep the formatter threads busy, none smaller than two block rows
        max_rows = max(2, -(-n_rows // (2 * self.threads))) if n_rows > 0 else 0
        p.pieces = row_pieces(g_lo, g_hi, p.bh, max_rows) if p.bh else []
        p.futs = []
        for (f, a, b) in p.pieces:
            lo = (f * p.bh + a - g_lo) * p.bw
            p.futs.append(self.pool.submit(E.format_partition_rows_records, p.width, b - a, rec[lo:lo + (b - a) * p.bw]))
        return p

    def finish(self, p):
        G = p.frames * p.bh
        local = np.zeros((G, 6), np.int64)
        done = []
        for (f, a, b), fut in zip(p.pieces, p.futs):
            buf, sizes = fut.result()
            local[f * p.bh + a:f * p.bh + b] = sizes
            done.append((f, a, b, buf, sizes))
        sizes_all = parallel.all_reduce_sum(local, self.device) if self.world > 1 else local
        offs, total = section_offsets(sizes_all, p.frames, p.bh)
        part = p.path + ".part"
        if self.rank == 0:      # creates the file / cuts a longer stale one; concurrent pwrites of other ranks all lie below `total`
            fd = os.open(part, os.O_WRONLY | os.O_CREAT, 0o644)
            try:
                os.ftruncate(fd, total)
            finally:
                os.close(fd)
            self.renames.append((part, p.path))
        for item in done:
            self.writes.append(self.pool.submit(self._write_piece, part, offs, item))
        if p.binary:
            self._binary(p)
        p.rec = None
        return total

    def drain(self):
        """Called at the same points on every rank (end of a sequence, close): waits for this rank's writes, then - behind a barrier
        that puts EVERY rank's writes before it - rank 0 gives the finished files their final names."""
        for w in self.writes:
            self.bytes_written += w.result()
        self.writes = []
        if self.world > 1:
            parallel.all_reduce_sum(np.zeros(1, np.int64), self.device)      # barrier on the backend the job runs on
        for part, final in self.renames:
            os.replace(part, final)
        self.renames = []

    def close(self):
        self.drain()
        self.pool.shutdown(wait=True)

    # ---------------------------------------------------------------------------------------------------------------- workers
    @staticmethod
    def _write_piece(path, offs, item):
        f, a, b, buf, sizes = item
        seg = sizes.sum(axis=0)                       # bytes of this piece per section
        runs, pos = [], 0
        for s in range(6):
            off, ln = int(offs[f, s, a]), int(seg[s])
            if runs and runs[-1][0] + runs[-1][2] == off:      # whole frames: the six sections are one contiguous range
                runs[-1][2] += ln
            else:
                runs.append([off, pos, ln])
            pos += ln
        view = memoryview(buf)
        fd = os.open(path, os.O_WRONLY | os.O_CREAT, 0o644)
        try:
            for off, at, ln in runs:
                _pwrite_all(fd, view[at:at + ln], off)
        finally:
            os.close(fd)
        return pos

    def _binary(self, p):
        """Binary side channel (include/pmp.h, PMPB1): fixed-size matrices, so the offsets need no exchange."""
        final = p.path[:-4] + ".pmpb"
        path = final + ".part"
        R, Cc = 16 * p.bh, 16 * p.bw
        per = 5 * R * Cc + R * Cc // 4
        if self.rank == 0:
            fd = os.open(path, os.O_WRONLY | os.O_CREAT, 0o644)
            try:
                os.ftruncate(fd, 40 + p.frames * per)
                hdr = b"PMPB1\0\0\0" + np.array([p.frames, p.height, p.width, R, Cc, 0, 0, 0], "<i4").tobytes()
                _pwrite_all(fd, memoryview(hdr), 0)
            finally:
                os.close(fd)
            self.renames.append((path, final))
        rec, g_lo, bw = np.array(p.rec), p.g_lo, p.bw     # own copy: these jobs outlive finish(), the caller's buffer does not

        def job(f, a, b):
            lo = (f * p.bh + a - g_lo) * bw
            oh, ov, oq, od = E.tile_partition_rows_records(p.width, b - a, rec[lo:lo + (b - a) * bw])
            base = 40 + f * per
            fd = os.open(path, os.O_WRONLY | os.O_CREAT, 0o644)
            try:
                _pwrite_all(fd, memoryview(oh.reshape(-1)), base + a * 16 * Cc)
                _pwrite_all(fd, memoryview(ov.reshape(-1)), base + R * Cc + a * 16 * Cc)
                _pwrite_all(fd, memoryview(oq.reshape(-1)), base + 2 * R * Cc + a * 8 * (Cc // 2))
                for k in range(3):
                    _pwrite_all(fd, memoryview(od[k].reshape(-1).view(np.uint8)), base + 2 * R * Cc + R * Cc // 4 + k * R * Cc + a * 16 * Cc)
            finally:
                os.close(fd)
            return 0
        for (f, a, b) in p.pieces:
            self.writes.append(self.pool.submit(job, f, a, b))
