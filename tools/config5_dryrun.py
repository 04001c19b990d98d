"""BASELINE.json configs[4] harness, DRY RUN (build container only, CPU): the PMP encoder against the anchor encoder at the four QPs,
BD-rate and encoder-time saving.

    PLUMBING ONLY.  The trained MTT-net weights (*_BD_*.pkl) and the Class A1/A2 test sequences are absent from the reference mount
    (SURVEY.md F2), so this runs on a tiny synthetic sequence with the documented SYNTHETIC MTT weights: the numbers it prints say
    nothing about the method.  The day the blobs appear, point --yuv/--width/--height/--frames/--models at them: config 5 is this
    one command (on a GPU box the product's driver writes the partition files into <workdir>/PartitionMat instead of the oracle).

What runs: EncoderAppAnchor (the reference's patched VTM-10.0 with Partition_Map_Acceleration_fal = 0, Lib/CommonLib/TypeDef.h:61 -
the stock search) and EncoderApp (= 1, acceleration level L0: Acceleration_Config_fal 0, TypeDef.h:63) on the same frames with the
reference's own all-intra cfg (codec/demo/cfg/encoder_intra_vtm.cfg); both built from the sources in place by
tools/vtm_build/CMakeLists.txt.  The partition files come from the oracle's CPU restatement of the nets and post-processing through
the product's file writer here (this container has no GPU); on a GPU box the product's driver writes them.

    python tools/config5_dryrun.py [--width 256 --height 128 --frames 2 --qps 22,27,32,37]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

VTM_BUILD = os.path.join(ROOT, "tools", "vtm_build", "_build")
ENC_CFG = "/root/reference/codec/demo/cfg/encoder_intra_vtm.cfg"


def bd_rate(rate_a, psnr_a, rate_t, psnr_t):
    """Bjontegaard delta rate (%) of the test curve against the anchor: cubic fit of log10(rate) over PSNR, both integrated over the
    common PSNR interval.  Negative = the test needs fewer bits at equal quality."""
    la, lt = np.log10(np.asarray(rate_a, float)), np.log10(np.asarray(rate_t, float))
    pa, pt = np.asarray(psnr_a, float), np.asarray(psnr_t, float)
    ca, ct = np.polyfit(pa, la, 3), np.polyfit(pt, lt, 3)
    lo, hi = max(pa.min(), pt.min()), min(pa.max(), pt.max())
    if hi <= lo:
        return float("nan")
    ia, it = np.polyint(ca), np.polyint(ct)
    avg = ((np.polyval(it, hi) - np.polyval(it, lo)) - (np.polyval(ia, hi) - np.polyval(ia, lo))) / (hi - lo)
    return (10.0 ** avg - 1.0) * 100.0


def parse_encoder_log(text):
    """-> (kbps, Y-PSNR, YUV-PSNR, encoder seconds) from a VTM encoder log (the 'Total Frames | Bitrate ...' table + ' Total Time')."""
    m = re.search(r"Total Frames\s*\|\s*Bitrate[^\n]*\n\s*(\d+)\s+a\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", text)
    t = re.search(r"Total Time:\s*([\d.]+)\s*sec", text)
    if not m or not t:
        raise ValueError("encoder log not understood:\n" + text[-1500:])
    return float(m.group(2)), float(m.group(3)), float(m.group(6)), float(t.group(1))


def write_maps_with_oracle(dirpath, stem, y, u, v, width, height, frames, qp, models):
    """PartitionMat pair of one QP: oracle nets + oracle post-processing on the CPU, emitted by the PRODUCT's writer."""
    from oracle import nets_torch as O, postproc as P
    from pmp_vvc_tip2023_amd import engine, weights as W
    P.build()
    by, bu, bv = P.cut_blocks(y, u, v, 8)
    os.makedirs(dirpath, exist_ok=True)
    for comp in ("Luma", "Chroma"):
        luma = comp == "Luma"
        wq, _ = W.load_net_weights(comp + "_Q", qp, models)
        wb, src = W.load_net_weights(comp + "_MSBD", qp, models, allow_synthetic=True)
        x = O.luma_input(by) if luma else O.chroma_input(by, bu, bv)
        qt, bt, dire = O.infer_qbd(wq, wb, x, luma)
        hor, ver, q8, d8 = P.seq_post_process(qt, bt, dire, comp, frames, width, height, None)
        engine.write_partition_file(os.path.join(dirpath, "%s_%s_QP%d_PartitionMat.txt" % (stem, comp, qp)), frames, height, width, hor, ver,
                                    q8.astype(np.uint8), d8)
    return src


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--qps", default="22,27,32,37")
    ap.add_argument("--yuv", default=None, help="8-bit 4:2:0 input (default: a synthetic recipe-R sequence)")
    ap.add_argument("--models", default=None, help="CTU_Models directory (default: the packaged QT nets + synthetic MTT nets)")
    ap.add_argument("--out", default=None, help="also write the report here")
    args = ap.parse_args()
    qps = [int(q) for q in args.qps.split(",")]
    from pmp_vvc_tip2023_amd import synth

    subprocess.check_call(["cmake", "--build", VTM_BUILD, "--target", "EncoderApp", "EncoderAppAnchor", "-j8"], stdout=subprocess.DEVNULL)
    W_, H_, F_ = args.width, args.height, args.frames
    tmp = tempfile.mkdtemp(prefix="pmp_cfg5_")
    stem = "Dry_%dx%d_30" % (W_, H_)
    if args.yuv:
        raw = np.fromfile(args.yuv, np.uint8)[:F_ * W_ * H_ * 3 // 2].reshape(F_, -1)
        y = raw[:, :W_ * H_].reshape(F_, H_, W_); u = raw[:, W_ * H_:W_ * H_ * 5 // 4].reshape(F_, H_ // 2, W_ // 2)
        v = raw[:, W_ * H_ * 5 // 4:].reshape(F_, H_ // 2, W_ // 2)
    else:
        y, u, v = synth.recipe_r_frames(F_, H_, W_, 55)
    with open(os.path.join(tmp, stem + ".yuv"), "wb") as f:
        for i in range(F_):
            f.write(y[i].tobytes()); f.write(u[i].tobytes()); f.write(v[i].tobytes())
    open(os.path.join(tmp, "seq.cfg"), "w").write("InputFile : %s.yuv\nInputBitDepth : 8\nFrameRate : 30\nFrameSkip : 0\nSourceWidth : %d\n"
                                                  "SourceHeight : %d\nFramesToBeEncoded : %d\nLevel : 4\n" % (stem, W_, H_, F_))
    rows = []
    src = None
    for qp in qps:
        t0 = time.time()
        src = write_maps_with_oracle(os.path.join(tmp, "PartitionMat"), stem, y, u, v, W_, H_, F_, qp, args.models)
        t_maps = time.time() - t0
        res = {}
        for name, exe in (("anchor", "EncoderAppAnchor"), ("pmp", "EncoderApp")):
            r = subprocess.run([os.path.join(VTM_BUILD, exe), "-c", "seq.cfg", "-c", ENC_CFG, "-f", str(F_), "-ts", "1", "-q", str(qp),
                                "-b", "%s_%d.bin" % (name, qp), "-o", ""], cwd=tmp, capture_output=True, text=True, timeout=7200)
            if r.returncode != 0:
                raise SystemExit("%s failed at QP %d:\n%s" % (exe, qp, (r.stdout + r.stderr)[-2000:]))
            res[name] = parse_encoder_log(r.stdout)
        rows.append((qp, res["anchor"], res["pmp"], t_maps))
    lines = ["config 5 DRY RUN - plumbing only: %dx%d, %d frames, all-intra, MTT weights: %s" % (W_, H_, F_, src),
             "%4s | %10s %8s %8s | %10s %8s %8s | %7s" % ("QP", "anchor kbps", "Y-PSNR", "enc s", "PMP kbps", "Y-PSNR", "enc s", "time saving")]
    for qp, a, p, tm in rows:
        lines.append("%4d | %10.2f %8.3f %8.2f | %10.2f %8.3f %8.2f | %6.1f %%" % (qp, a[0], a[1], a[3], p[0], p[1], p[3], 100 * (1 - p[3] / a[3])))
    if len(rows) >= 4:
        bd = bd_rate([r[1][0] for r in rows], [r[1][1] for r in rows], [r[2][0] for r in rows], [r[2][1] for r in rows])
        ts = 100 * (1 - sum(r[2][3] for r in rows) / sum(r[1][3] for r in rows))
        lines.append("BD-rate (Y) of the PMP encoder against the anchor: %+.2f %%;  encoder time saving over the four QPs: %.1f %%" % (bd, ts))
    lines.append("(synthetic MTT weights and a synthetic sequence: these figures test the harness, not the method)")
    report = "\n".join(lines)
    print(report)
    if args.out:
        open(args.out, "w").write(report + "\n")


if __name__ == "__main__":
    main()
