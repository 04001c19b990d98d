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
"""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

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


def import_yuv420(file_path, width, height, frm_num, SubSampleRatio=1, is10bit=False, frames=None):
    """Inference_QBD.py:78-102: every SubSampleRatio-th frame of a planar 4:2:0 file -> y[F,H,W], u,v[F,H/2,W/2].
    frames=(k0, k1) reads only the sub-sampled frames k0 <= k < k1 (a rank's shard: nothing else is touched on disk)."""
    pix = width * height
    sub = (frm_num + SubSampleRatio - 1) // SubSampleRatio
    k0, k1 = (0, sub) if frames is None else (max(0, int(frames[0])), min(sub, int(frames[1])))
    nf = max(0, k1 - k0)
    dt = np.uint16 if is10bit else np.uint8
    y = np.zeros((nf, height, width), dt); u = np.zeros((nf, height // 2, width // 2), dt); v = np.zeros_like(u)
    with open(file_path, "rb") as fp:
        for k in range(k0, k1):
            i = k * SubSampleRatio
            fp.seek(i * pix * 3 if is10bit else i * pix * 3 // 2, 0)
            y[k - k0] = np.fromfile(fp, dtype=dt, count=pix).reshape(height, width)
            u[k - k0] = np.fromfile(fp, dtype=dt, count=pix // 4).reshape(height // 2, width // 2)
            v[k - k0] = np.fromfile(fp, dtype=dt, count=pix // 4).reshape(height // 2, width // 2)
    return y, u, v


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
    p.add_argument("--hostBlocks", action="store_true",
                   help="keep the cut blocks in host memory and upload them for every (component, QP) pass, as the reference "
                        "does; default: frames are uploaded once, cut on the GPU and the blocks stay device-resident")
    return p


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
            return torch.from_numpy(np.ascontiguousarray(a).view(np.int16) if a.dtype == np.uint16 else np.ascontiguousarray(a)).to(dev)
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
        torch.cuda.synchronize(dev)

    def infer_postprocess_records(self, comp, qp):
        """One pass; returns the shard's packed records u8[n, 1344] as a DEVICE tensor (a fresh one per pass: the previous
        pass's records may still be in flight to the writer)."""
        chroma = comp == "Chroma"
        rec = self.torch.empty((self.n, parallel.RECORD), dtype=self.torch.uint8, device=self.dev)
        self.eng.infer_postprocess_records_device(comp, qp, self.by.data_ptr(), self.bu.data_ptr() if chroma else None,
                                                  self.bv.data_ptr() if chroma else None, self.n, rec.data_ptr())
        self.eng.synchronize()      # the library runs on its own stream; the gather (torch's stream) must see finished records
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


def inference_VVC_seqs(args):
    """Inference_QBD.py:151-255."""
    rank, world, local = parallel.env_world()
    dev_id = args.device if args.device is not None else local
    if args.device is None and world > 1:
        import torch
        dev_id = local % max(torch.cuda.device_count(), 1)   # several ranks may share a GPU in smoke tests (gloo)
    model_dir = resolve_model_dir(args.modelDir)
    eng = E.Engine(dev_id, weight_dir=model_dir, allow_synthetic_mtt=args.allowSyntheticMTT)
    # --batchSize is the reference's blocks-per-forward-pass (a GPU memory knob there).  Results do not depend on it (tested:
    # ragged chunks are bit-identical to one pass); small passes only leave most of an MI355X idle, so it is ignored unless
    # --strictBatch asks for it (clamped to the library's 4096-block pass; the reference accepts any value)
    eng.set_chunk(min(max(1, args.batchSize), 4096) if args.strictBatch else 4096)
    eng.set_precision(args.precision)
    device = None
    if world > 1:
        import torch
        device = torch.device("cuda", dev_id)
        torch.cuda.set_device(device)
    parallel.init_process_group(device)

    torch_dev = None
    if not args.hostBlocks:
        try:
            import torch
            if torch.cuda.is_available():
                torch_dev = torch.device("cuda", dev_id)
        except ImportError:
            torch_dev = None

    save_dir = os.path.join(args.outDir, args.jobID, "PartitionMat")
    if rank == 0:
        os.makedirs(save_dir, exist_ok=True)
    qps = [int(q) for q in args.qps.split(",") if q]
    comps = [c for c in args.comps.split(",") if c]

    names, files, widths, heights, frames, sub_frames, _ = load_sequences_info(_resolve(args.seqTable, args.inputDir), args.ssRatio)
    end = min(args.startSeqID + args.seqNum, len(names))
    nseq = max(0, end - args.startSeqID)
    seqs_block_time = np.zeros(max(nseq, 1))
    seqs_net_time = np.zeros((max(nseq, 1), 4, 2))
    seqs_post_time = np.zeros((max(nseq, 1), 4, 2))

    writers = ThreadPoolExecutor(max_workers=4) if rank == 0 else None
    pending = []
    for comp in comps:  # weights once per (comp, qp), not once per sequence; a missing file raises here, before any output
        for qp in qps:
            eng.load(comp, qp)
    if rank == 0:
        for (net, qp), src in sorted(eng.provenance.items()):
            print("weights %s QP%d: %s" % (net, qp, src), flush=True)
        if any(str(src).startswith("synthetic") for src in eng.provenance.values()):
            print("WARNING: MTT nets run on SYNTHETIC weights (--allowSyntheticMTT): the PartitionMat files are not usable "
                  "for encoding", file=sys.stderr, flush=True)

    for si, seq_id in enumerate(range(args.startSeqID, end)):
        seq_name, stem = names[seq_id], strip_yuv_suffix(files[seq_id])
        width, height, numfrm, sub_numfrm = widths[seq_id], heights[seq_id], frames[seq_id], sub_frames[seq_id]
        seq_path, is10bit = parse_seq_cfg(os.path.join(args.cfgDir, seq_name + ".cfg"))
        seq_path = _resolve(seq_path, args.inputDir)
        if rank == 0:
            print(seq_name, flush=True)
        # ---- load input blocks: every rank reads and cuts only the frames that hold its own block range
        t0 = time.time()
        per_frame = (width // 64) * (height // 64)
        n_total = per_frame * sub_numfrm
        lo, hi = parallel.shard_bounds(n_total, rank, world)
        dblk = None
        by = np.zeros((0, 68, 68), np.uint8); bu = np.zeros((0, 34, 34), np.uint8); bv = np.zeros((0, 34, 34), np.uint8)
        if hi > lo and per_frame:
            f0, f1 = shard_frames(lo, hi, per_frame)
            y, u, v = import_yuv420(seq_path, width, height, numfrm, args.ssRatio, is10bit, frames=(f0, f1))
            if torch_dev is not None:   # frames go up once, are cut on the GPU and the blocks never leave it
                dblk = DeviceBlocks(eng, torch_dev, y, u, v, 10 if is10bit else 8, lo - f0 * per_frame, hi - f0 * per_frame)
            else:
                by, bu, bv = eng.output_block_yuv(y, u, v, 10 if is10bit else 8)
                by, bu, bv = (a[lo - f0 * per_frame:hi - f0 * per_frame] for a in (by, bu, bv))
            del y, u, v
        seqs_block_time[si] = time.time() - t0

        for comp in comps:
            comp_id = COMP_COLUMN[comp]                # Time_Sta columns: Luma first, whatever --comps lists
            for qp in qps:
                qi = (qp - 22) // 5 if qp in QPS else 0
                t0 = time.time()
                reruns = eng.saturation_reruns()
                if dblk is not None:     # records stay on the device until rank 0 has them all
                    local = dblk.infer_postprocess_records(comp, qp)
                else:
                    local = parallel.pack_records(*eng.infer_postprocess(comp, qp, by, bu, bv))
                seqs_net_time[si, qi, comp_id] = time.time() - t0
                if eng.saturation_reruns() != reruns:   # f16x3 range guard (include/pmp.h): results are right, the pass cost 3x
                    print("WARNING: rank %d: %s %s QP%d drove an activation beyond the fp16 range of the f16x3 datapath; the pass was "
                          "re-run on bf16x6 (consider --precision bf16x6 for this model)" % (rank, seq_name, comp, qp), file=sys.stderr, flush=True)
                t0 = time.time()
                rec = parallel.gather_records(local, n_total, device)
                if rank == 0:
                    # text emission (645 k lines per 1080p frame and file) runs on writer threads - the C writer releases
                    # the GIL - so it overlaps the next (component, QP) pass instead of serialising rank 0
                    save_path = os.path.join(save_dir, "%s_%s_QP%d_PartitionMat.txt" % (stem, comp, qp))
                    print("Save:", save_path, flush=True)
                    pending.append(writers.submit(_emit, rec, save_path, sub_numfrm, height, width, args.binary))
                seqs_post_time[si, qi, comp_id] = time.time() - t0

        for fut in pending:   # bound memory: a sequence's files are on disk before the next one starts
            fut.result()
        pending = []
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


def launch_ranks(n, argv):
    """`--gpus N` without a launcher: N fresh rank processes (the parent makes no GPU call), exit code = first failure."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    procs = [subprocess.Popen([sys.executable, "-m", "pmp_vvc_tip2023_amd.inference_qbd"] + list(argv), env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(n)]
    rc = 0
    try:
        for p in procs:
            p.wait()
            rc = rc or p.returncode
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
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
