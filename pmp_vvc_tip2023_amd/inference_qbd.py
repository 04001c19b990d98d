"""Network inference + post-processing driver: drop-in for the reference's Inference_QBD.py.

    python -m pmp_vvc_tip2023_amd.inference_qbd --jobID 0000 --inputDir /input/ --outDir /output/ \\
           --batchSize 200 --startSeqID 0 --seqNum 22                     (Inference_QBD.py:257-267, same flags)
    torchrun --nproc-per-node 8 -m pmp_vvc_tip2023_amd.inference_qbd ...  (blocks sharded over the GPUs)

Output, byte-compatible with what the patched VTM-10.0 reads (EncAppCfg.cpp:4234-4404):
    <outDir>/<jobID>/PartitionMat/<seq-file-stem>_<Luma|Chroma>_QP<22|27|32|37>_PartitionMat.txt   (:153,:237)
    <outDir>/<jobID>/Time_Sta_<start>_<end>.txt    5 comma-terminated columns x 4 QP rows per sequence  (:243-253)

Mirrors, function by function: load_sequences_info (:48-76), import_yuv420 (:78-102), output_block_yuv (:104-149,
on the GPU), inference_VVC_seqs (:151-255).  Differences, all flags with the reference's values as defaults where
the reference hard-codes them:
  --seqTable   Training_Sequences.txt in the reference (:50, file not shipped); default VVC_Test_Sequences.txt
  --cfgDir     ".\\per-sequence" (:162)                      --modelDir "./CTU_Models" (:219-220), falls back to weights/
  --ssRatio    30 (:26); the codec demo uses TemporalSubsampleRatio 8 (encoder_intra_vtm.cfg:61)
  sequence file stem: the reference's rstrip(".yuv") strips a character SET (:166); a real suffix strip is used
  (identical for every name in VVC_Test_Sequences.txt and what EncAppCfg.cpp:4235-4242 expects).
  Nets are loaded once per (component, QP) instead of once per sequence (:208-224).
  --batchSize is the reference's blocks-per-pass knob; passes of 4096 blocks are used unless --strictBatch (identical results).
  Frames are uploaded once per sequence and cut on the GPU; the blocks stay device-resident for all passes (--hostBlocks
  restores the reference's host-side block arrays).
  Multi-GPU (--gpus N / torchrun): a sequence's BLOCK ROWS are sharded over the ranks; every rank formats and pwrites the text of
  its own rows (emit.py; the ranks exchange only a table of byte counts), so nothing funnels through rank 0.  --emit gather keeps
  the older path: RCCL gather of the 1344-byte records to rank 0, which writes the file alone.
  The job is pipelined: the next sequence's frames are read on a host thread while this one runs; while the GPU runs pass k+1
  the host copies pass k's records (pinned buffers), formats and writes them.
"""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import emit
from . import engine as E
from . import parallel

QPS = (22, 27, 32, 37)
COMP_COLUMN = {"Luma": 0, "Chroma": 1}
DEFAULT_MODEL_DIR = "./CTU_Models"                 # Inference_QBD.py:219-220


def load_sequences_info(seqs_info_path, ss_ratio, num=None):
    """Inference_QBD.py:48-76: rows `name,file,W,H,frames,fps` until a line containing 'end!!!!'."""
    data = []
    with open(seqs_info_path, "r") as fp:
        for line in fp:
            if "end!!!!" in line:
                break
            line = line.rstrip("\n").strip()
            if line:
                data.append(line.split(","))
    if num is not None:
        data = data[:num]
    names = [d[0] for d in data]
    files = [d[1] for d in data]
    width = [int(d[2]) for d in data]
    height = [int(d[3]) for d in data]
    frames = [int(d[4]) for d in data]
    sub = [(f + ss_ratio - 1) // ss_ratio for f in frames]
    blocks = [(w // 64) * (h // 64) * s for w, h, s in zip(width, height, sub)]
    return names, files, width, height, frames, sub, blocks


def parse_seq_cfg(path):
    """Inference_QBD.py:171-186: InputFile and InputBitDepth from a VTM per-sequence cfg ('#' starts a comment)."""
    seq_path, is10bit = None, False
    with open(path) as fp:
        for line in fp:
            if "InputFile" in line:
                line = line.rstrip("\n").split("#")[0].replace(" ", "")
                seq_path = line.split(":", 1)[1]
            elif "InputBitDepth" in line:
                line = line.rstrip("\n").split("#")[0].replace(" ", "")
                is10bit = line.split(":", 1)[1] == "10"
    if seq_path is None:
        raise ValueError("%s: no InputFile entry" % path)
    return seq_path, is10bit


def import_yuv420(file_path, width, height, frm_num, SubSampleRatio=1, is10bit=False, frames=None, alloc=None):
    """Inference_QBD.py:78-102: every SubSampleRatio-th frame of a planar 4:2:0 file -> y[F,H,W], u,v[F,H/2,W/2].
    frames=(k0, k1) reads only the sub-sampled frames k0 <= k < k1 (a rank's shard: nothing else is touched on disk).
    alloc(shape, dtype) supplies the arrays (pinned host memory in the driver); the planes are read straight into them."""
    pix = width * height
    sub = (frm_num + SubSampleRatio - 1) // SubSampleRatio
    k0, k1 = (0, sub) if frames is None else (max(0, int(frames[0])), min(sub, int(frames[1])))
    nf = max(0, k1 - k0)
    dt = np.uint16 if is10bit else np.uint8
    alloc = alloc or np.zeros
    y = alloc((nf, height, width), dt); u = alloc((nf, height // 2, width // 2), dt); v = alloc((nf, height // 2, width // 2), dt)
    bps = 2 if is10bit else 1
    with open(file_path, "rb", buffering=0) as fp:
        for k in range(k0, k1):
            i = k * SubSampleRatio
            fp.seek(i * (pix * 3 // 2) * bps, 0)
            _read_plane(fp, y[k - k0]); _read_plane(fp, u[k - k0]); _read_plane(fp, v[k - k0])
    return y, u, v


def _read_plane(fp, a):
    """One plane straight into its (C-contiguous) array: no intermediate copy, whatever memory `a` lives in."""
    buf = memoryview(a.reshape(-1)).cast("B")
    got = 0
    while got < len(buf):
        k = fp.readinto(buf[got:])
        if not k:
            raise IOError("%s: short read (the file holds fewer frames than the sequence table says)" % getattr(fp, "name", "input"))
        got += k


def shard_frames(lo, hi, per_frame):
    """Sub-sampled frames [f0, f1) that hold blocks lo <= b < hi (per_frame blocks each, frame-major order)."""
    return lo // per_frame, (hi + per_frame - 1) // per_frame


def strip_yuv_suffix(name):
    return name[:-4] if name.endswith(".yuv") else name


def build_parser():
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("--jobID", type=str, default="0000")
    p.add_argument("--inputDir", type=str, default="/input/")
    p.add_argument("--outDir", type=str, default="/output/")
    p.add_argument("--batchSize", default=200, type=int, help="blocks per forward pass (library chunk)")
    p.add_argument("--startSeqID", default=0, type=int)
    p.add_argument("--seqNum", default=22, type=int)
    # paths the reference hard-codes
    p.add_argument("--seqTable", default="VVC_Test_Sequences.txt")
    p.add_argument("--cfgDir", default=os.path.join(".", "per-sequence"))
    p.add_argument("--modelDir", default=DEFAULT_MODEL_DIR)
    p.add_argument("--ssRatio", default=30, type=int, help="temporal sub-sampling (Inference_QBD.py:26)")
    p.add_argument("--qps", default="22,27,32,37")
    p.add_argument("--comps", default="Luma,Chroma")
    p.add_argument("--device", default=None, type=int, help="GPU index (default: LOCAL_RANK)")
    p.add_argument("--gpus", default=1, type=int,
                   help="shard every sequence's blocks over this many GPUs of the node: started without a launcher the driver spawns "
                        "its own rank processes (one per GPU, RCCL gather of the records to rank 0); under torchrun it is one rank")
    p.add_argument("--binary", action="store_true", help="also write <name>_PartitionMat.pmpb (binary side channel, include/pmp.h)")
    p.add_argument("--strictBatch", action="store_true",
                   help="run exactly --batchSize blocks per pass (default: the library's 4096-block chunk; same results)")
    p.add_argument("--allowSyntheticMTT", action="store_true",
                   help="run the MTT nets on the documented SYNTHETIC weights when a <Comp>_BD_<qp> model file is missing (the "
                        "reference checkout ships none); without this flag a missing model file is an error, as in the reference")
    p.add_argument("--precision", default="f16x3", choices=["f16x3", "bf16x6", "fp32"],
                   help="convolution datapath (all fp32-equivalent, include/pmp.h): f16x3 is fastest and range-guarded")
    p.add_argument("--emit", default="sharded", choices=["sharded", "gather"],
                   help="sharded (default): every rank formats and pwrites the text of its own block rows, the ranks exchange only byte "
                        "counts; gather: RCCL gather of the records to rank 0, which writes the file alone")
    p.add_argument("--emitThreads", default=0, type=int, help="formatter / writer threads per rank (0 = cores / ranks, at most 8)")
    p.add_argument("--overlap", action="store_true",
                   help="turn the library's overlap mode on (include/pmp.h: pmp_set_overlap): a pass of >= 1024 blocks runs as two chunks on "
                        "two streams, so one chunk's small launches fill the gaps of the other's large ones; bit-identical files.  Opt-in "
                        "(round 6): it takes 0.2-1.7 %% off a bare 4096-block step but nothing off a whole job (8 x 4K frames, all eight "
                        "files: 1.702 s on, 1.701 s off, profiles/r05e_driver_bench.txt) and costs a second workspace")
    p.add_argument("--hostBlocks", action="store_true",
                   help="keep the cut blocks in host memory and upload them for every (component, QP) pass, as the reference "
                        "does; default: frames are uploaded once, cut on the GPU and the blocks stay device-resident")
    return p


class PinnedPool:
    """Page-locked host buffers, reused: frames go up (and records come down) by DMA without a staging copy, and pinning - a
    system call per buffer - is paid once per size, not once per sequence.  torch is the allocator here, nothing else."""

    def __init__(self, torch):
        import threading
        self.torch, self.free, self.lock = torch, [], threading.Lock()   # the reader thread takes, the main thread gives

    def take(self, nbytes):
        nbytes = max(int(nbytes), 1)
        with self.lock:
            best = None
            for i, t in enumerate(self.free):
                if t.numel() >= nbytes and (best is None or t.numel() < self.free[best].numel()):
                    best = i
            if best is not None:
                return self.free.pop(best)
        return self.torch.empty(nbytes, dtype=self.torch.uint8, pin_memory=True)

    def give(self, t):
        with self.lock:
            if len(self.free) >= 8:                   # bound the pinned footprint: drop the smallest
                self.free.sort(key=lambda x: x.numel())
                self.free.pop(0)
            self.free.append(t)

    def array(self, shape, dtype):
        """(ndarray view of a pinned buffer, the buffer) - give() the buffer back when the array is dead."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        t = self.take(n)
        return t.numpy()[:n].view(dtype).reshape(shape), t


class DeviceBlocks:
    """Device-resident blocks of one sequence shard: the sub-sampled frames are uploaded once, cut by
    pmp_cut_blocks_device (Inference_QBD.py:104-149 on the GPU) and reused by all eight (component, QP) passes; only the split
    flags (1344 B per block) come back.  torch is the device allocator here, nothing else."""

    def __init__(self, eng, dev, y, u, v, bitdepth, lo, hi):
        import torch
        self.eng, self.torch, self.dev = eng, torch, dev
        F, H, W = y.shape
        per_frame = (H // 64) * (W // 64)

        def up(a):   # torch has no uint16: 10-bit planes travel as int16 bit patterns
            return torch.from_numpy(np.ascontiguousarray(a).view(np.int16) if a.dtype == np.uint16 else np.ascontiguousarray(a)).to(dev, non_blocking=True)
        ty, tu, tv = up(y), up(u), up(v)
        self.by = torch.empty((F * per_frame, 68, 68), dtype=torch.uint8, device=dev)
        self.bu = torch.empty((F * per_frame, 34, 34), dtype=torch.uint8, device=dev)
        self.bv = torch.empty((F * per_frame, 34, 34), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        eng.cut_blocks_device(ty.data_ptr(), tu.data_ptr(), tv.data_ptr(), F, H, W, bitdepth, self.by.data_ptr(), self.bu.data_ptr(), self.bv.data_ptr())
        eng.synchronize()
        del ty, tu, tv
        self.by, self.bu, self.bv = self.by[lo:hi], self.bu[lo:hi], self.bv[lo:hi]   # leading-dimension slices: still contiguous
        self.n = hi - lo

    def enqueue(self, comp, qp):
        """Enqueue one pass on the library's stream and return at once: the shard's packed records u8[n, 1344] as a DEVICE tensor
        (a fresh one per pass - the previous pass's records may still be on their way to the host).  They are final after
        eng.synchronize() (include/pmp.h: the range guard is settled there)."""
        chroma = comp == "Chroma"
        rec = self.torch.empty((self.n, parallel.RECORD), dtype=self.torch.uint8, device=self.dev)
        self.eng.infer_postprocess_records_device(comp, qp, self.by.data_ptr(), self.bu.data_ptr() if chroma else None,
                                                  self.bv.data_ptr() if chroma else None, self.n, rec.data_ptr())
        return rec

    def infer_postprocess_records(self, comp, qp):
        """One pass, finished: the records as a device tensor."""
        rec = self.enqueue(comp, qp)
        self.eng.synchronize()
        return rec


def _emit(rec, save_path, frames, height, width, binary):
    h, v, q, d = parallel.unpack_records(rec)
    E.write_partition_file(save_path, frames, height, width, h, v, q, d)
    if binary:
        E.write_partition_binary(save_path[:-4] + ".pmpb", frames, height, width, h, v, q, d)


def resolve_model_dir(model_dir):
    """--modelDir as given must exist; only the untouched default (./CTU_Models, Inference_QBD.py:219-220) falls back to
    the packaged weights/ directory when it is absent.  A mistyped directory is an error, not a silent fallback."""
    if os.path.isdir(model_dir):
        return model_dir
    if model_dir == DEFAULT_MODEL_DIR:
        return None                                   # weights.default_weight_dir()
    raise FileNotFoundError("--modelDir %s does not exist" % model_dir)


def _resolve(path, base):
    return path if os.path.isabs(path) or os.path.exists(path) else os.path.join(base, path)


class Stages:
    """Wall-clock seconds the MAIN thread of this rank spent per stage (tools/driver_bench.py prints them)."""
    NAMES = ("setup", "weights", "read_wait", "h2d_cut", "first_enqueue", "gpu_wait", "enqueue", "d2h", "emit_start", "emit_finish", "gather", "drain", "teardown")

    def __init__(self):
        self.t = dict.fromkeys(self.NAMES, 0.0)
        self.prefetch_read = 0.0        # seconds the reader thread spent in file I/O (hidden behind the passes unless read_wait shows it)
        self.blocks = 0
        self.passes = 0

    def add(self, name, t0):
        t1 = time.perf_counter()
        self.t[name] += t1 - t0
        return t1


LAST_STAGES = None      # the Stages of the last inference_VVC_seqs() run in this process (tools/driver_bench.py reads it)


def inference_VVC_seqs(args):
    """Inference_QBD.py:151-255."""
    global LAST_STAGES
    t_entry = time.perf_counter()
    rank, world, local = parallel.env_world()
    if world > 1 and args.device is not None:
        raise SystemExit("--device selects the GPU of a single-process run; with %d ranks every rank takes the GPU of its LOCAL_RANK" % world)
    dev_id = args.device if args.device is not None else local
    backend = os.environ.get("PMP_DIST_BACKEND", "nccl")
    if world > 1:
        import torch
        ndev = torch.cuda.device_count()
        if local >= ndev:
            if backend == "nccl":   # two RCCL ranks on one device fail or hang at init (duplicate GPU): refuse with a readable message
                raise SystemExit("rank %d of %d has no GPU of its own (%d visible): RCCL needs one GPU per rank (PMP_DIST_BACKEND=gloo "
                                 "lets several ranks share a GPU, for smoke tests only)" % (rank, world, ndev))
            dev_id = local % max(ndev, 1)
    model_dir = resolve_model_dir(args.modelDir)
    eng = E.Engine(dev_id, weight_dir=model_dir, allow_synthetic_mtt=args.allowSyntheticMTT)
    # --batchSize is the reference's blocks-per-forward-pass (a GPU memory knob there).  Results do not depend on it (tested:
    # ragged chunks are bit-identical to one pass); small passes only leave most of an MI355X idle, so it is ignored unless
    # --strictBatch asks for it (clamped to the library's 4096-block pass; the reference accepts any value)
    eng.set_chunk(min(max(1, args.batchSize), 4096) if args.strictBatch else 4096)
    eng.set_precision(args.precision)
    eng.set_overlap(args.overlap)            # opt-in: measured to buy nothing at job level (see --overlap)
    device = None
    if world > 1:
        import torch
        device = torch.device("cuda", dev_id)
        torch.cuda.set_device(device)
    parallel.init_process_group(device)
    if world > 1:
        info = parallel.preflight(device)            # a broken RCCL / IPC setup fails HERE, with a message, not in the first real pass
        parallel.relax_timeout()                     # PMP_DIST_TIMEOUT_S bounded the rendezvous; the passes' collectives get PMP_DIST_COLLECTIVE_TIMEOUT_S
        if rank == 0:
            print("ranks: %d (%s), first collective %.1f ms" % (info["ranks"], info["backend"], info["ms"]), flush=True)

    torch_dev = None
    if not args.hostBlocks:
        try:
            import torch
            if torch.cuda.is_available():
                torch_dev = torch.device("cuda", dev_id)
            else:
                print("WARNING: torch sees no GPU although the library runs on one: blocks stay on the HOST (as with --hostBlocks)",
                      file=sys.stderr, flush=True)
        except ImportError:
            torch_dev = None
    pinned = None
    if torch_dev is not None:
        import torch
        pinned = PinnedPool(torch)

    save_dir = os.path.join(args.outDir, args.jobID, "PartitionMat")
    os.makedirs(save_dir, exist_ok=True)             # every rank writes its own rows into the files
    qps = [int(q) for q in args.qps.split(",") if q]
    comps = [c for c in args.comps.split(",") if c]
    sharded = args.emit == "sharded"

    names, files, widths, heights, frames, sub_frames, _ = load_sequences_info(_resolve(args.seqTable, args.inputDir), args.ssRatio)
    end = min(args.startSeqID + args.seqNum, len(names))
    nseq = max(0, end - args.startSeqID)
    seqs_block_time = np.zeros(max(nseq, 1))
    seqs_net_time = np.zeros((max(nseq, 1), 4, 2))
    seqs_post_time = np.zeros((max(nseq, 1), 4, 2))
    st = Stages()
    LAST_STAGES = st

    emitter = emit.ShardEmitter(rank, world, threads=args.emitThreads, device=device) if sharded else None
    writers = ThreadPoolExecutor(max_workers=4) if (rank == 0 and not sharded) else None
    pending = []
    # Weights once per (comp, qp), not once per sequence.  A missing file raises HERE, before any output - but only the first
    # pass's nets are loaded now: the others go up while the GPU runs the pass before theirs (30 ms of packing each, hidden).
    for comp in comps:
        for qp in qps:
            eng.check_available(comp, qp)
    announced = set()

    def load_weights(comp, qp):
        eng.load(comp, qp)
        if rank == 0:
            for (net, q), src in sorted(eng.provenance.items()):
                if (net, q) not in announced:
                    announced.add((net, q))
                    print("weights %s QP%d: %s" % (net, q, src), flush=True)
                    if str(src).startswith("synthetic") and "warned" not in announced:
                        announced.add("warned")
                        print("WARNING: MTT nets run on SYNTHETIC weights (--allowSyntheticMTT): the PartitionMat files are not usable "
                              "for encoding", file=sys.stderr, flush=True)

    # ---- the reader: one thread, one sequence ahead.  A rank reads only the frames that hold its own block rows.
    def load_sequence(seq_id):
        t0 = time.perf_counter()
        seq_name = names[seq_id]
        width, height, numfrm, sub_numfrm = widths[seq_id], heights[seq_id], frames[seq_id], sub_frames[seq_id]
        seq_path, is10bit = parse_seq_cfg(os.path.join(args.cfgDir, seq_name + ".cfg"))
        seq_path = _resolve(seq_path, args.inputDir)
        bh, bw = height // 64, width // 64
        per_frame = bh * bw
        if sharded:      # whole block rows per rank (emit.py)
            g_lo, g_hi = emit.shard_rows(sub_numfrm, bh, rank, world)
            lo, hi = g_lo * bw, g_hi * bw
        else:            # contiguous block ranges, gathered to rank 0
            lo, hi = parallel.shard_bounds(per_frame * sub_numfrm, rank, world)
            g_lo = g_hi = 0
        d = dict(name=seq_name, stem=strip_yuv_suffix(files[seq_id]), width=width, height=height, sub_numfrm=sub_numfrm, is10bit=is10bit,
                 bh=bh, bw=bw, per_frame=per_frame, n_total=per_frame * sub_numfrm, lo=lo, hi=hi, g_lo=g_lo, g_hi=g_hi, y=None, bufs=[])
        if hi > lo and per_frame:
            f0, f1 = shard_frames(lo, hi, per_frame)
            alloc = None
            if pinned is not None:
                def alloc(shape, dtype):
                    a, t = pinned.array(shape, dtype)
                    d["bufs"].append(t)
                    return a
            d["y"], d["u"], d["v"] = import_yuv420(seq_path, width, height, numfrm, args.ssRatio, is10bit, frames=(f0, f1), alloc=alloc)
            d["f0"] = f0
        d["read_s"] = time.perf_counter() - t0
        return d

    reader = ThreadPoolExecutor(max_workers=1)
    seq_ids = list(range(args.startSeqID, end))
    nxt = reader.submit(load_sequence, seq_ids[0]) if seq_ids else None
    st.add("setup", t_entry)                          # context, rendezvous, weights of every (component, QP): once per job
    passes = [(comp, qp) for comp in comps for qp in qps]
    rec_host = [None, None]      # pinned double buffer for the records of the pass being formatted / the pass being copied

    for si, seq_id in enumerate(seq_ids):
        t0 = tw = time.perf_counter()
        sq = nxt.result()
        nxt = reader.submit(load_sequence, seq_ids[si + 1]) if si + 1 < len(seq_ids) else None
        st.prefetch_read += sq["read_s"]
        tw = st.add("read_wait", tw)
        width, height, sub_numfrm, per_frame, n_total = sq["width"], sq["height"], sq["sub_numfrm"], sq["per_frame"], sq["n_total"]
        lo, hi, bw = sq["lo"], sq["hi"], sq["bw"]
        if rank == 0:
            print(sq["name"], flush=True)
        # ---- load input blocks: frames go up once, are cut on the GPU and the blocks never leave it
        dblk = None
        by = np.zeros((0, 68, 68), np.uint8); bu = np.zeros((0, 34, 34), np.uint8); bv = np.zeros((0, 34, 34), np.uint8)
        if sq["y"] is not None:
            o = sq["f0"] * per_frame
            if torch_dev is not None:
                dblk = DeviceBlocks(eng, torch_dev, sq["y"], sq["u"], sq["v"], 10 if sq["is10bit"] else 8, lo - o, hi - o)
            else:
                by, bu, bv = eng.output_block_yuv(sq["y"], sq["u"], sq["v"], 10 if sq["is10bit"] else 8)
                by, bu, bv = (a[lo - o:hi - o] for a in (by, bu, bv))
        for t in sq["bufs"]:
            pinned.give(t)
        sq["y"] = sq["u"] = sq["v"] = None
        st.add("h2d_cut", tw)
        st.blocks += (hi - lo) * len(passes)
        st.passes += len(passes)
        seqs_block_time[si] = time.perf_counter() - t0

        n_local = hi - lo
        tw = time.perf_counter()
        if passes:
            load_weights(*passes[0])
        tw = st.add("weights", tw)
        cur = dblk.enqueue(*passes[0]) if (dblk is not None and passes) else None
        tw = st.add("first_enqueue", tw)               # the job's first pass also sizes and allocates the activation workspace
        if len(passes) > 1:
            load_weights(*passes[1])                   # next to pass 0 on the GPU
            st.add("weights", tw)
        prev = None
        for k, (comp, qp) in enumerate(passes):
            comp_id = COMP_COLUMN[comp]                # Time_Sta columns: Luma first, whatever --comps lists
            qi = (qp - 22) // 5 if qp in QPS else 0
            save_path = os.path.join(save_dir, "%s_%s_QP%d_PartitionMat.txt" % (sq["stem"], comp, qp))
            t0 = tw = time.perf_counter()
            reruns = eng.saturation_reruns()
            if dblk is not None:
                eng.synchronize()                      # pass k is done and final (range guard settled)
                tw = st.add("gpu_wait", tw)
                local = cur
                if k + 1 < len(passes):                # the GPU starts on pass k+1 before the host touches pass k's records
                    cur = dblk.enqueue(*passes[k + 1])
                    tw = st.add("enqueue", tw)
                    if k + 2 < len(passes):
                        load_weights(*passes[k + 2])   # no-op after the first sequence
                        tw = st.add("weights", tw)
            else:
                load_weights(comp, qp)
                local = parallel.pack_records(*eng.infer_postprocess(comp, qp, by, bu, bv)) if n_local else np.zeros((0, parallel.RECORD), np.uint8)
                tw = st.add("gpu_wait", tw)
            seqs_net_time[si, qi, comp_id] = time.perf_counter() - t0
            if eng.saturation_reruns() != reruns:   # f16x3 range guard (include/pmp.h): results are right, the pass cost 3x
                print("WARNING: rank %d: %s %s QP%d drove an activation beyond the fp16 range of the f16x3 datapath; the pass was "
                      "re-run on the fp32 datapath (consider --precision bf16x6 for this model)" % (rank, sq["name"], comp, qp), file=sys.stderr, flush=True)
            t0 = tw = time.perf_counter()
            if rank == 0:
                print("Save:", save_path, flush=True)
            if sharded:
                if dblk is not None:                   # D2H into a pinned buffer on torch's stream, next to pass k+1 on the library's
                    import torch
                    need = n_local * parallel.RECORD
                    if rec_host[k & 1] is None or rec_host[k & 1].numel() < need:
                        rec_host[k & 1] = pinned.take(need)
                    ht = rec_host[k & 1][:need].view(n_local, parallel.RECORD)
                    ht.copy_(local)
                    host = ht.numpy()
                else:
                    host = np.ascontiguousarray(local).reshape(n_local, parallel.RECORD)
                tw = st.add("d2h", tw)
                if prev is not None:                   # pass k-1: exchange the byte counts, queue the writes (its formatting ran beside pass k)
                    emitter.finish(prev)
                    tw = st.add("emit_finish", tw)
                prev = emitter.start(save_path, sub_numfrm, height, width, sq["g_lo"], sq["g_hi"], host, binary=args.binary)
                tw = st.add("emit_start", tw)
            else:
                rec = parallel.gather_records(local, n_total, device)
                tw = st.add("gather", tw)
                if rank == 0:
                    # text emission (645 k lines per 1080p frame and file) runs on writer threads - the C writer releases
                    # the GIL - so it overlaps the next (component, QP) pass instead of serialising rank 0
                    pending.append(writers.submit(_emit, rec, save_path, sub_numfrm, height, width, args.binary))
            seqs_post_time[si, qi, comp_id] = time.perf_counter() - t0

        tw = time.perf_counter()
        if prev is not None:
            emitter.finish(prev)
            tw = st.add("emit_finish", tw)
        if emitter is not None:
            emitter.drain()                            # bound memory: a sequence's files are on disk before the next one starts
        for fut in pending:
            fut.result()
        pending = []
        st.add("drain", tw)
        del dblk
    tw = time.perf_counter()
    reader.shutdown(wait=True)
    if emitter is not None:
        emitter.close()
    if writers:
        writers.shutdown(wait=True)
    if rank == 0:  # Time_Sta log, Inference_QBD.py:243-253 (net column = inference + GPU post-processing here)
        sta = os.path.join(args.outDir, args.jobID, "Time_Sta_%d_%d.txt" % (args.startSeqID, args.startSeqID + args.seqNum))
        with open(sta, "w") as fp:
            for si in range(nseq):
                for qp_id in range(4):
                    for s in (seqs_block_time[si], seqs_net_time[si, qp_id, 0], seqs_net_time[si, qp_id, 1],
                              seqs_post_time[si, qp_id, 0], seqs_post_time[si, qp_id, 1]):
                        fp.write(str(s))
                        fp.write(",")
                    fp.write("\n")
        print("Sum time:", np.sum(seqs_block_time) + np.sum(seqs_net_time) + np.sum(seqs_post_time))
    eng.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    st.add("teardown", tw)
    return st


def launch_ranks(n, argv):
    """`--gpus N` without a launcher: N fresh rank processes (the parent makes no GPU call), all of them watched: the first rank
    that fails ends the job with its exit code (parallel.spawn_ranks)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rc, _ = parallel.spawn_ranks([sys.executable, "-m", "pmp_vvc_tip2023_amd.inference_qbd"] + list(argv), n,
                                 env_extra={"PYTHONPATH": root + os.pathsep + os.environ.get("PYTHONPATH", "")})
    return rc


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = build_parser().parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        rc = launch_ranks(args.gpus, argv)
        if rc:
            raise SystemExit(rc)
        return
    t0 = time.time()
    inference_VVC_seqs(args)
    print("Total inference time:", time.time() - t0)


if __name__ == "__main__":
    main()
