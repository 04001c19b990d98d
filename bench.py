#!/usr/bin/env python3
"""bench.py — throughput of the partition-map prediction hot path on MI355X.

One "step" = one pass of the device-resident hot path (QT-net + MTT-net inference + Map2Partition
post-processing, pmp_infer_postprocess_device) over one batch of synthetic blocks already resident in HBM.
Workload at every N: BASELINE.json configs[1] — Luma QT+MTT nets, QP22, batch = 1024 synthetic 128x128 CTUs per GPU
(recipe R; QT weights real, MTT weights synthetic because the reference's *_BD_*.pkl are missing).
Unit: the nets consume 64x64 blocks (+4 px context); one VTM CTU is 128x128 = 4 blocks, so a step is 4096 blocks
(one library pass: the default chunk is 4096 blocks) and CTU/s = blocks/s / 4.  `value` is CTU/s;
blocks/s is reported next to it.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: one process per GPU.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) this process is one rank;
started plainly with --gpus N it launches its own N rank processes - fresh children, before this process has made any GPU
call - relays rank 0's JSON line and exits non-zero if any rank failed.  Blocks are sharded over the ranks (weak scaling:
--ctus per GPU) with no data-path collective; the one exchange is the RCCL gather of the packed split-flag records to rank 0.

Rank 0 prints ONE JSON line on stdout; diagnostics go to stderr.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

FLOP_PER_BLOCK = {"Luma": 6.991e9, "Chroma": 2.300e9}      # conv MACs x 2, QT + MTT
# Peak of the datapath actually used: the exact-fp32 MFMA, or the dense 16-bit MFMA peak divided by the
# number of 16-bit products the split spends per fp32-accurate product: 3 (two fp16 terms) or 6 (three bf16 terms)
# (MI355X_MICROARCH.md: 157.3 TF fp32 matrix, ~2.5 PF bf16/fp16 dense).
PEAK_TFLOPS = {"fp32": 157.3, "bf16x6": 2500.0 / 6.0, "f16x3": 2500.0 / 3.0}
PEAK_NOTE = {"f16x3": "fp16 dense MFMA peak 2500 TFLOP/s / 3 products per fp32-accurate product = 833.3 "
                      "(a register-resident 16-bit MFMA loop on this pool sustains 1925 TFLOP/s = 641.7 per fp32-accurate product, "
                      "profiles/r01_mfma_peak_microbench.txt)",
             "fp32": "v_mfma_f32_16x16x4_f32 peak 157.3 TFLOP/s (register-resident loop on this pool: 154.7)",
             "bf16x6": "bf16 dense MFMA peak 2500 TFLOP/s / 6 products per fp32-accurate product = 416.7 "
                       "(register-resident bf16 loop on this pool sustains 1925 TFLOP/s at 1.97 GHz = 320.9 per fp32 product, "
                       "profiles/r01_mfma_peak_microbench.txt)"}
DOMINANT = "conv_mfma_3x3_c64"                               # 3x3 64->64 convs: 57.8 % of the MTT-net FLOPs
KERNEL_SOURCE = {"f16x3": "conv_f16x3.hip", "bf16x6": "conv_bf16x6.hip", "fp32": "conv_mfma.hip"}   # the dominant class's kernel file per datapath


def pmc_traffic(precision, blocks_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed PMC record (profiles/pmc_traffic.json, tools/make_traffic.py) -
    but only if that record was taken on THIS build of the kernel: the record carries the sha256 of the kernel source it was measured
    on, and a kernel file that has changed since gets `traffic: null` with the reason instead of somebody else's bytes."""
    import hashlib
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        tj = json.load(open(tp))
    except Exception as e:      # noqa: BLE001
        return None, "profiles/pmc_traffic.json not readable (%s)" % e
    key = DOMINANT + ":" + precision
    if tj.get(key) is None:
        return None, "profiles/pmc_traffic.json holds no PMC record for the %s datapath" % precision
    build = tj.get("_build:" + precision)
    src = os.path.join(ROOT, "pmp_vvc_tip2023_amd", "csrc", KERNEL_SOURCE[precision])
    try:
        sha = hashlib.sha256(open(src, "rb").read()).hexdigest()
    except OSError:
        sha = None
    if not build or not build.get("kernel_sha256"):
        return None, "the PMC record of the %s datapath carries no build stamp (taken before round 6): not attributed to this build" % precision
    if sha != build["kernel_sha256"]:
        return None, ("the PMC record was taken on another build of %s (sha256 %s..., git %s; this run: %s...): not attributed to this build"
                      % (KERNEL_SOURCE[precision], build["kernel_sha256"][:12], (build.get("git_head") or "?")[:10], (sha or "unreadable")[:12]))
    scale = blocks_per_launch / float(tj.get("_blocks_per_launch:" + precision, blocks_per_launch))
    return round(tj[key] * scale), ("profiles/pmc_traffic.json: rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE; %s) of THIS kernel build "
                                    "(%s sha256 %s..., recorded at git %s), scaled to the blocks per launch; not collected in this run"
                                    % (build.get("pmc_summary", "?"), KERNEL_SOURCE[precision], sha[:12], (build.get("git_head") or "?")[:10]))


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(sample_blocks, seed, eng=None):
    """The oracle (a port of the reference's torch CPU path + C post-processing) on this host's cores.  With `eng`, the same
    sample also goes through the HIP path and the JSON gets a `parity` object (logits max-abs error, split-flag
    mismatches) - the oracle is the checker here, never the thing timed as the product."""
    from oracle import nets_torch as O, postproc as P
    from pmp_vvc_tip2023_amd import synth, weights as W
    ncpu = os.cpu_count() or 1
    y, _, _ = synth.recipe_r_blocks(sample_blocks, seed)
    wq, _ = W.load_net_weights("Luma_Q", 22)
    wbd, _ = W.load_net_weights("Luma_MSBD", 22, allow_synthetic=True)
    x = O.luma_input(y)
    # torch's CPU convs do not scale to hundreds of threads on 64x64 maps: calibrate the thread count on 8 blocks
    best = (None, 1e30)
    for th in sorted({min(ncpu, t) for t in (8, 16, 32, 64)}):   # hundreds of threads run at < 1 block/s: not probed
        torch.set_num_threads(th)
        O.infer_qbd(wq, wbd, x[:8], True, batch=8)           # warm-up (oneDNN primitive creation)
        t = time.perf_counter()
        O.infer_qbd(wq, wbd, x[:8], True, batch=8)
        dt = time.perf_counter() - t
        log("cpu_baseline: %3d threads -> %.1f blocks/s on 8 blocks" % (th, 8 / dt))
        if dt < best[1]:
            best = (th, dt)
        if dt > 20:
            break
    cores = best[0]
    torch.set_num_threads(cores)
    sample_blocks = max(8, min(sample_blocks, int(20.0 / (best[1] / 8)) // 8 * 8))   # ~20 s of CPU work
    x = x[:sample_blocks]
    t0 = time.perf_counter()
    qt, bt, dire = O.infer_qbd(wq, wbd, x, True, batch=64)
    t1 = time.perf_counter()
    ohor, over, oq8, od8 = P.seq_post_process(qt, bt, dire, "Luma", 1, 64 * sample_blocks, 64, None)
    t2 = time.perf_counter()
    log("cpu_baseline: nets %.2fs, post-proc %.3fs for %d blocks on %d threads" % (t1 - t0, t2 - t1, sample_blocks, cores))
    out = {"value": round(sample_blocks / 4.0 / (t2 - t0), 3), "unit": "CTU/s", "cores": cores, "threads": cores, "host_cores": ncpu,
           "kind": "port", "blocks_per_s": round(sample_blocks / (t2 - t0), 2),
           "sample": "%d luma blocks QP22 (recipe R seed %d): torch-CPU fp32 QT+MTT forward (batch 64, %d threads) + "
                     "C oracle post-processing (1 thread); the host has %d cores - the thread count is the fastest of {8, 16, 32, 64} "
                     "(capped at the core count) on an 8-block probe, because torch's CPU convolutions stop scaling on 64x64 maps "
                     "well below the core count (probe timings on stderr)" % (sample_blocks, seed, cores, ncpu)}
    out["all_cores"] = cpu_all_cores(cores, ncpu, seed, best[1] / 8)
    parity = None
    if eng is not None:
        yb = np.ascontiguousarray(y[:sample_blocks])
        gq, gb, gd = eng.inference_pre_QBD("Luma", 22, yb)
        ghor, gver, gq8, gd8 = eng.infer_postprocess("Luma", 22, yb)
        # flags of the device logits through the oracle's post-processing: the bit-exactness claim of the integer stage
        phor, pver, pq8, pd8 = P.seq_post_process(gq, gb, gd, "Luma", 1, 64 * sample_blocks, 64, None)
        parity = {"blocks": int(sample_blocks),
                  "logit_max_abs_err": float(max(np.abs(gq - qt).max(), np.abs(gb - bt).max(), np.abs(gd - dire).max())),
                  "tolerance": 1e-3,
                  "flag_mismatch_vs_oracle_postproc_of_device_logits": int((ghor != phor).sum() + (gver != pver).sum() +
                                                                          (gq8 != pq8.astype(gq8.dtype)).sum() + (gd8 != pd8).sum()),
                  "flag_mismatch_vs_cpu_end_to_end": int((ghor != ohor).sum() + (gver != over).sum() +
                                                         (gq8 != oq8.astype(gq8.dtype)).sum() + (gd8 != od8).sum()),
                  "flags_compared": int(ghor.size + gver.size + gq8.size + gd8.size)}
    return out, parity


def cpu_worker(blocks, seed, threads):
    """`bench.py --cpu-worker`: one process of the whole-host CPU baseline.  Torch CPU only (no GPU call).  Warms up, says
    "ready", waits for "go" on stdin, runs the oracle on its own slice and prints its wall-clock window."""
    from oracle import nets_torch as O, postproc as P
    from pmp_vvc_tip2023_amd import synth, weights as W
    torch.set_num_threads(threads)
    y, _, _ = synth.recipe_r_blocks(blocks, seed)
    wq, _ = W.load_net_weights("Luma_Q", 22)
    wbd, _ = W.load_net_weights("Luma_MSBD", 22, allow_synthetic=True)
    x = O.luma_input(y)
    q, b, d = O.infer_qbd(wq, wbd, x[:8], True, batch=8)
    P.seq_post_process(q, b, d, "Luma", 1, 64 * 8, 64, None)
    print("ready", flush=True)
    budget = float(sys.stdin.readline().split()[1])           # "go <seconds>": work in batches of 16 blocks until the budget is used up
    t0 = time.time()
    done = 0
    while done < blocks and (done == 0 or time.time() - t0 < budget):
        m = min(16, blocks - done)
        q, b, d = O.infer_qbd(wq, wbd, x[done:done + m], True, batch=16)
        P.seq_post_process(q, b, d, "Luma", 1, 64 * m, 64, None)
        done += m
    print("done %.6f %.6f %d" % (t0, time.time(), done), flush=True)


def cpu_limits():
    """What the box lets this job use: scheduler affinity and the cgroup CPU quota (cores; None = unlimited / unreadable)."""
    out = {"affinity": None, "cgroup_quota_cores": None}
    try:
        out["affinity"] = len(os.sched_getaffinity(0))
    except Exception:      # noqa: BLE001
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                out["cgroup_quota_cores"] = None if txt[0] == "max" else round(int(txt[0]) / int(txt[1]), 2)
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                out["cgroup_quota_cores"] = None if q < 0 else round(q / per, 2)
            break
        except Exception:  # noqa: BLE001
            continue
    return out


def cpu_all_cores(threads, ncpu, seed, s_per_block):
    """What the WHOLE host does on the same workload: host_cores // threads fresh worker processes (torch CPU, `threads` threads each,
    no GPU import), each on its own slice of recipe-R blocks, released together; blocks / (last end - first start).  `value` above
    stays the single calibrated process for continuity; this is the box's actual CPU capacity for the path (blocks are independent,
    so the reference's CPU path shards the same way)."""
    import subprocess
    procs_n = max(1, ncpu // max(threads, 1))
    # every worker runs for a fixed budget of wall time (workers slow each other down - shared caches, memory bandwidth, a CPU quota of
    # the container - so a fixed block count would make this leg's duration unpredictable); a generous supply of blocks each
    budget_s = 15.0
    blocks = max(16, min(1024, int(2 * budget_s / max(s_per_block, 1e-3)) // 16 * 16))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    import tempfile
    import threading
    errs = [tempfile.TemporaryFile(mode="w+") for _ in range(procs_n)]      # a worker's stderr: its tail goes to the log if it fails
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", "%d,%d,%d" % (blocks, seed + 100 + i, threads)],
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=errs[i], env=env, text=True) for i in range(procs_n)]

    def line(p, deadline_s, what):
        """One line of a worker's stdout, or an error once the deadline has passed (a reader thread: a hung worker must not hang the bench)."""
        got = []
        th = threading.Thread(target=lambda: got.append(p.stdout.readline()), daemon=True)
        th.start()
        th.join(deadline_s)
        if th.is_alive() or not got:
            raise RuntimeError("a CPU worker did not %s within %g s" % (what, deadline_s))
        return got[0]
    try:
        t_up = time.time()
        for p in procs:      # import torch + oneDNN warm-up under a CPU quota shared by all workers: minutes at worst, not hours
            if line(p, max(10.0, 300.0 - (time.time() - t_up)), "come up").strip() != "ready":
                raise RuntimeError("a CPU worker did not come up")
        for p in procs:
            p.stdin.write("go %g\n" % budget_s); p.stdin.flush()
        wins = []
        t_go = time.time()
        for p in procs:      # the budget, plus the one batch a worker finishes after it
            tok = line(p, max(10.0, budget_s + 120.0 - (time.time() - t_go)), "finish").split()
            if len(tok) != 4 or tok[0] != "done":
                raise RuntimeError("a CPU worker died")
            wins.append((float(tok[1]), float(tok[2]), int(tok[3])))
        for p in procs:
            p.wait(timeout=30)
    except Exception as e:      # noqa: BLE001 - a baseline that cannot be taken is reported as such, it never fails the bench
        for p in procs:
            if p.poll() is None:
                p.kill()
        log("cpu_baseline: whole-host leg failed: %s" % e)
        for i, f in enumerate(errs):
            f.seek(0)
            tail = f.read()[-600:].strip()
            if tail:
                log("cpu_baseline: worker %d stderr tail: %s" % (i, tail))
        return {"error": str(e)}
    wall = max(w[1] for w in wins) - min(w[0] for w in wins)
    total = sum(w[2] for w in wins)
    log("cpu_baseline: whole host: %d processes x %d threads, %d blocks in %.2f s" % (procs_n, threads, total, wall))
    lim = cpu_limits()
    return {"processes": procs_n, "threads_each": threads, "cores_used": procs_n * threads, "host_cores": ncpu, "cpu_limits": lim, "blocks": total,
            "blocks_per_s": round(total / wall, 2), "value": round(total / 4.0 / wall, 3), "unit": "CTU/s", "wall_s": round(wall, 2),
            "note": "fresh worker processes (torch CPU only), each the single-process configuration above on its own recipe-R slice, released "
                    "together for %g s of wall time each; rate = all blocks / (last end - first start).  If this is no more than the single "
                    "process's rate, the box does not give this job more cores than one process already uses (cpu_limits: scheduler affinity "
                    "and the cgroup's CPU quota)" % budget_s}


def trained_like_extra(eng, dev, n, step, timed):
    """The same step with TRAINED-LIKE MTT weights (trained_like.msbd_weights: tensors bootstrapped from the real QT-net tensors,
    trunks at 1e3, gate products at 1e4) instead of the benign uniform ones: does the headline describe a net with trained-scale
    activations?  Reports the step time, the f16x3 activation-scale exponents the library chose and the range guard's re-run count
    (must be 0: a re-run is the 2.8x slower fp32 datapath).  Also at a stress setting (trunk x 64, gates x 16)."""
    from pmp_vvc_tip2023_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import trained_like                  # tools/trained_like.py: test-weight data, not part of the product package
    out = {}
    for tag, gains in (("", {}), ("_stress_k64_g16", {"trunk_gain": 64.0, "gate_gain": 16.0})):
        eng.load("Luma", 22, msbd_weights=trained_like.msbd_weights("Luma", 22, **gains))
        rep = eng.activation_report("Luma", 22)
        eng.clear_saturation()
        from pmp_vvc_tip2023_amd import sensors
        with sensors.Sampler(sensors.for_torch_device(dev.index or 0)) as smp:
            dt = timed(lambda: step("Luma", 22), 5)
        eng.synchronize()
        sm = smp.summary()
        out["sclk_mhz_mean" + tag] = sm["sclk_mhz"]["mean"] if sm["sclk_mhz"] else None      # the same kernels on sparser activations: does the clock move?
        out["power_w_mean" + tag] = sm["power_w"]["mean"] if sm["power_w"] else None
        out["ms_per_step" + tag] = round(dt * 1e3, 3)
        out["ctu_per_s" + tag] = round(n / 4.0 / dt, 2)
        out["saturation_reruns" + tag] = eng.saturation_reruns()
        out["activation_exps" + tag] = rep["exps"]
        out["segment_amax" + tag] = [round(float(m), 1) for m in rep["seg_amax"]]
    eng.load("Luma", 22, msbd_weights=synth.synth_msbd_weights("Luma", 22))      # back to the headline's weights
    out["note"] = ("Luma QP22, device-resident step, 5 steps after 1 warm-up; MTT weights bootstrapped from the real Luma_Q/Chroma_Q tensors "
                   "(tests/golden/g2b_msbd_trained_like.npz pins them against the reference modules); activation_exps = per-segment power-of-two "
                   "activation scales chosen by the library's calibration pass (include/pmp.h); not used for `value`")
    return out


def fp32_exact_extra(eng, n, step, timed):
    """The same step on the EXACT-arithmetic datapath (PMP_PRECISION_F32: v_mfma_f32_16x16x4_f32, a bit-exact fmaf chain - the
    reference's own arithmetic, and what a call costs when the f16x3 range guard re-runs it): 3 steps, with the dominant class timed by
    events around its launches as in the headline's roofline, against the fp32 matrix peak."""
    eng.set_precision("fp32")
    try:
        dt = timed(lambda: step("Luma", 22), 3)
        names = [eng.lib.pmp_ktime_name(k).decode() for k in range(eng.lib.pmp_ktime_classes())]
        eng.ktime_enable(1 << names.index(DOMINANT))
        for _ in range(3):
            step("Luma", 22)
        launches, ms, flops = eng.ktime()[DOMINANT]
        eng.ktime_enable(0)
    finally:
        eng.set_precision("f16x3")
    tf = flops / (ms * 1e-3) / 1e12 if ms else None
    return {"ctu_per_s": round(n / 4.0 / dt, 2), "ms_per_step": round(dt * 1e3, 3), "steps": 3, "dtype": "f32",
            "net_tflops": round(n / dt * FLOP_PER_BLOCK["Luma"] / 1e12, 2),
            "roofline": {"bound": "mfma", "kernel": DOMINANT, "achieved": round(tf, 2) if tf else None, "peak": PEAK_TFLOPS["fp32"], "unit": "TFLOP/s",
                         "frac": round(tf / PEAK_TFLOPS["fp32"], 4) if tf else None, "launches": launches,
                         "avg_launch_ms": round(ms / launches, 4) if launches else None, "peak_basis": PEAK_NOTE["fp32"]},
            "note": "Luma QP22, the headline's blocks and weights, pmp_set_precision(PMP_PRECISION_F32); not used for `value`"}


def measure_extras(eng, args, dev, n, y, u, v, step):
    """Side measurements AFTER the timed region (they never enter `value`), N = 1 only:
    * e2e_host_buffers: the same step through the host-pointer boundary (pmp_infer_postprocess: H2D of the u8 blocks, the
      pass, D2H of the split flags) - SURVEY.md 8(d) config 2's "end-to-end" number next to the device-resident one;
    * chroma and per-QP luma throughput of the device-resident step (3 steps each)."""
    out = {}

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t) / reps

    chroma = args.comp == "Chroma"
    dt = timed(lambda: eng.infer_postprocess(args.comp, args.qp, y, u if chroma else None, v if chroma else None), 3)
    h2d = n * (68 * 68 + (2 * 34 * 34 if chroma else 0))
    out["e2e_host_buffers"] = {"value": round(n / 4.0 / dt, 2), "unit": "CTU/s", "ms_per_step": round(dt * 1e3, 3), "steps": 3,
                               "h2d_bytes_per_step": h2d, "d2h_bytes_per_step": n * 1344,
                               "note": "pmp_infer_postprocess on pageable host buffers: H2D + pass + D2H, synchronised per step"}
    if args.comp == "Luma" and args.qp == 22:
        per_qp = {}
        for qp in (22, 27, 32, 37):
            eng.load("Luma", qp)
            per_qp[str(qp)] = round(n / 4.0 / timed(lambda: step("Luma", qp), 3), 2)
        eng.load("Luma", 22)
        eng.set_overlap(True)       # pmp_set_overlap: two chunks of the step in flight on two streams (never inside the timed region: two
        ov_luma = timed(lambda: step("Luma", 22), 5)       # launches share the device then and per-launch durations lose their meaning)
        eng.load("Chroma", 22)
        ov_chroma = timed(lambda: step("Chroma", 22), 5)
        eng.set_overlap(False)
        chroma_dt = timed(lambda: step("Chroma", 22), 3)
        out["extra"] = {"luma_ctu_per_s_by_qp": per_qp, "chroma_qp22_ctu_per_s": round(n / 4.0 / chroma_dt, 2),
                        "overlap_mode": {"luma_qp22_ctu_per_s": round(n / 4.0 / ov_luma, 2), "luma_ms_per_step": round(ov_luma * 1e3, 3),
                                         "chroma_qp22_ctu_per_s": round(n / 4.0 / ov_chroma, 2), "chroma_ms_per_step": round(ov_chroma * 1e3, 3),
                                         "note": "opt-in pmp_set_overlap(ctx, 1) / PMP_OVERLAP=1: the step as two 2048-block chunks on two streams; "
                                                 "bit-identical records; not used for `value`"},
                        "note": "device-resident step, 3 steps each after 1 warm-up; same blocks; chroma counts the 64x64-luma-area "
                                "block (34x34 chroma inputs) as the unit, as the luma figure does"}

        if args.precision == "f16x3":
            out["extra"]["trained_like"] = trained_like_extra(eng, dev, n, step, timed)
            out["extra"]["fp32_exact"] = fp32_exact_extra(eng, n, step, timed)

        def classes(comp):
            """hipEvent time of every kernel class over 3 steps (events around every launch: a few % slower than the untimed step)."""
            eng.ktime_enable(0xFFFF)
            for _ in range(3):
                step(comp, 22)
            kt = eng.ktime()
            eng.ktime_enable(0)
            return {k: {"launches_per_step": round(ln / 3.0, 2), "ms_per_step": round(ms / 3.0, 4),
                        "tflops": round(fl / (ms * 1e-3) / 1e12, 2) if ms else None} for k, (ln, ms, fl) in kt.items() if ln}
        peak = PEAK_TFLOPS[args.precision]
        cl_luma, cl_chroma = classes("Luma"), classes("Chroma")
        out["breakdown"] = {"luma": cl_luma, "chroma": cl_chroma,
                            "note": "per kernel class, %d blocks per step; conv classes carry algorithmic FLOPs (2 per MAC), the others 0" % n}
        c_net = n / chroma_dt * FLOP_PER_BLOCK["Chroma"] / 1e12
        dom = max((k for k in cl_chroma if cl_chroma[k]["tflops"]), key=lambda k: cl_chroma[k]["ms_per_step"])
        out["chroma_roofline"] = {"bound": "mfma", "kernel": dom, "achieved": cl_chroma[dom]["tflops"], "peak": round(peak, 1), "unit": "TFLOP/s",
                                  "frac": round(cl_chroma[dom]["tflops"] / peak, 4), "net_tflops": round(c_net, 2),
                                  "net_frac": round(c_net / peak, 4), "blocks_per_s": round(n / chroma_dt, 1),
                                  "note": "Chroma QT+MTT QP22, 2.300 GFLOP per block (SURVEY 8d); dominant class by time"}
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this parent never touches the GPU - a
    process that has initialised HIP must not be replaced or forked into ranks), watch ALL of them - the first rank that fails
    ends the job at once with its exit code, the others are killed by PID - and relay rank 0's stdout."""
    from pmp_vvc_tip2023_amd import parallel            # numpy only: no GPU call
    rc, out0 = parallel.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], n, env_extra={"PMP_BENCH_CHILD": "1"},
                                    capture_rank0=True)
    out0 = out0 or b""
    if rc == 0:
        sys.stdout.write(out0.decode())
        sys.stdout.flush()
    return rc or (0 if out0.strip() else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ctus", type=int, default=1024, help="128x128 CTUs per GPU per step (4 blocks each)")
    ap.add_argument("--batch", type=int, default=0, help="blocks per GPU per step (overrides --ctus; tests and tools)")
    ap.add_argument("--comp", default="Luma", choices=["Luma", "Chroma"])
    ap.add_argument("--qp", type=int, default=22)
    ap.add_argument("--chunk", type=int, default=0, help="blocks per pass inside the library (0 = library default)")
    ap.add_argument("--cpu-sample", type=int, default=512, help="blocks for the CPU baseline (0 = skip)")
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "bf16x6", "fp32"],
                    help="conv datapath: 2-term fp16 split (3 MFMA products) or 3-term bf16 split (6 products), both "
                         "fp32-equivalent, or exact fp32 MFMA")
    ap.add_argument("--breakdown", action="store_true", help="extra pass with every kernel class timed (stderr)")
    ap.add_argument("--lib", default=None, help="path of another build of libpmp_hip.so (same-box A/B timing, tools/lib_ab.py)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the side measurements after the timed region (host-buffer end-to-end rate, chroma, per-QP luma)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)      # internal: "<blocks>,<seed>,<threads>" (cpu_all_cores)
    args = ap.parse_args()

    if args.cpu_worker:
        b, sd, th = (int(t) for t in args.cpu_worker.split(","))
        cpu_worker(b, sd, th)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # no launcher: become one (before any GPU call)
        raise SystemExit(launch_ranks(args.gpus))
    # stdout carries exactly ONE line, the JSON: whatever a library prints on fd 1 (gloo / RCCL banners) goes to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log("warning: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    n_gpus = world if world > 1 else 1
    # PMP_DIST_FORCE=1: a ONE-rank process group - how the single-GPU test box runs the RCCL branch of the step (communicator
    # creation, the device-tensor gather on the side stream) before an 8-GPU node ever does
    use_dist = n_gpus > 1 or os.environ.get("PMP_DIST_FORCE") == "1"

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        if world > 1 and os.environ.get("PMP_DIST_BACKEND", "nccl") == "nccl":
            raise SystemExit("bench.py: rank %d of %d has no GPU of its own (%d visible): RCCL needs one GPU per rank "
                             "(PMP_DIST_BACKEND=gloo lets several ranks share a GPU, for smoke tests only)"
                             % (rank, world, torch.cuda.device_count()))
        local_rank = local_rank % max(torch.cuda.device_count(), 1)   # smoke tests: several ranks on one GPU (gloo)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    preflight = None
    if use_dist:
        import torch.distributed as dist
        from pmp_vvc_tip2023_amd import parallel
        os.environ.setdefault("PMP_DIST_BACKEND", "nccl")        # "nccl" = RCCL; "gloo" only to smoke-test the control flow
        try:
            parallel.init_process_group(dev)                     # bounded: a rank that never arrives is an error, not a hang
            # first collective, content verified, BEFORE any timing: a broken RCCL / IPC setup ends here with a message
            preflight = parallel.preflight(dev)
        except Exception as e:                                    # noqa: BLE001
            log("bench.py: rank %d: multi-GPU preflight failed: %s" % (rank, e))
            raise SystemExit(3)
        log("rank %d: preflight ok: %d ranks over %s, %.1f ms" % (rank, preflight["ranks"], preflight["backend"], preflight["ms"]))
        parallel.relax_timeout()                                 # the short bound was for the rendezvous; a slow rank is not a missing one

    from pmp_vvc_tip2023_amd import _lib, engine, synth
    if args.lib:
        _lib.load(args.lib)
    eng = engine.Engine(local_rank, allow_synthetic_mtt=True)
    if args.chunk:
        eng.set_chunk(args.chunk)
    eng.set_precision(args.precision)
    # One explicit side stream for everything in a step: the library's launches, the record packing and the RCCL gather
    # are ordered on it (torch's default stream has handle 0, which pmp_set_stream reads as "use the context's own").
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    eng.set_stream(stream.cuda_stream)
    eng.load(args.comp, args.qp)
    log("rank %d: weights %s" % (rank, {k[0]: v for k, v in eng.provenance.items()}))

    n = args.batch if args.batch > 0 else 4 * args.ctus
    y, u, v = synth.recipe_r_blocks(n, 1 + rank)            # seed 1 on rank 0
    d_y = torch.from_numpy(y).to(dev)
    d_u = torch.from_numpy(u).to(dev)
    d_v = torch.from_numpy(v).to(dev)
    # one packed result record per block, written by the post-processing kernel itself (include/pmp.h, PMP_RECORD_BYTES):
    # hor[256] | ver[256] | qt[64] | dire[768] = 1344 bytes - the unit of the gather
    res = torch.empty((n, 1344), dtype=torch.uint8, device=dev)
    gathered = torch.empty((world * n, 1344), dtype=torch.uint8, device=dev) if (use_dist and rank == 0) else None
    pu = d_u.data_ptr() if args.comp == "Chroma" else None
    pv = d_v.data_ptr() if args.comp == "Chroma" else None

    gather_events = []          # (start, end) event pairs around the collective, on the stream it is ordered on
    gather_host_s = [0.0]

    def step(comp=args.comp, qp=args.qp, timed=False):
        chroma = comp == "Chroma"
        eng.infer_postprocess_records_device(comp, qp, d_y.data_ptr(), d_u.data_ptr() if chroma else None,
                                             d_v.data_ptr() if chroma else None, n, res.data_ptr())
        if use_dist:
            # the path's only exchange: split-flag records of every shard go to rank 0, which owns the file writer
            if dist.get_backend() == "nccl":
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                dist.gather(res, list(gathered.split(n)) if rank == 0 else None, dst=0)      # RCCL, device to device over xGMI
                if timed:
                    e1.record(stream)
                    gather_events.append((e0, e1))
            else:
                eng.synchronize()
                th = time.perf_counter()
                rc = res.cpu()
                dist.gather(rc, [torch.empty_like(rc) for _ in range(world)] if rank == 0 else None, dst=0)
                gather_host_s[0] += (time.perf_counter() - th) if timed else 0.0

    for _ in range(args.warmup):
        step()
    eng.synchronize()
    torch.cuda.synchronize(dev)
    mask = 1 << [eng.lib.pmp_ktime_name(k).decode() for k in range(eng.lib.pmp_ktime_classes())].index(DOMINANT)
    eng.ktime_enable(mask)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    from pmp_vvc_tip2023_amd import sensors
    sampler = sensors.Sampler(sensors.for_torch_device(local_rank))     # sysfs reads on a host thread: this GPU's clock and socket power
    sampler.__enter__()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timed=True)
    eng.synchronize()            # every step's range-guard snapshot looked at (and a saturated step re-run) inside the timed region
    own_elapsed = time.perf_counter() - t0
    sampler.__exit__()
    sens = sampler.summary()
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    kt = eng.ktime()
    eng.ktime_enable(0)
    multi = None
    if dist:
        cpu_side = dist.get_backend() != "nccl"
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if cpu_side else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # what the collective cost and how even the ranks were: if the N-GPU number disappoints, the line says why
        g_ms = (sum(a.elapsed_time(b) for a, b in gather_events) if gather_events else gather_host_s[0] * 1e3) / max(args.steps, 1)
        def sv(key, stat):
            return float(sens[key][stat]) if sens.get(key) else -1.0
        mine = torch.tensor([own_elapsed / args.steps * 1e3, g_ms, sv("sclk_mhz", "mean"), sv("sclk_mhz", "min"), sv("power_w", "mean"),
                             sv("power_w", "max"), float(eng.workspace_bytes())], dtype=torch.float64, device="cpu" if cpu_side else dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)

        def col(i, nd=1):
            return [None if float(e[i].item()) < 0 else round(float(e[i].item()), nd) for e in every]
        multi = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                 "gather_bytes_per_step": n * 1344 * world, "gather_ms": round(float(every[0][1].item()), 4),
                 "gather_ms_by_rank": [round(float(e[1].item()), 4) for e in every],
                 "ms_per_step_by_rank": [round(float(e[0].item()), 4) for e in every],
                 "preflight_ms": round(preflight["ms"], 2) if preflight else None,
                 "sclk_mhz_mean_by_rank": col(2), "sclk_mhz_min_by_rank": col(3), "power_w_mean_by_rank": col(4), "power_w_max_by_rank": col(5),
                 "workspace_bytes_by_rank": [int(e[6].item()) for e in every],
                 "note": "gather_ms: events around the collective on rank 0's stream (it waits for the slowest rank's records, so skew "
                         "shows up here); ms_per_step_by_rank: each rank's own loop before the closing barrier; sclk / power: every rank's own "
                         "GPU sampled from sysfs hwmon by a host thread during the timed region (pmp_vvc_tip2023_amd/sensors.py; null = not "
                         "readable) - the dominant kernels are clock-limited under the package power cap, so a sub-linear curve with lower "
                         "clocks than the 1-GPU line's `sensors` is node power, with even clocks and a long gather_ms it is the exchange"}

    blocks_per_s = n * n_gpus * args.steps / elapsed
    launches, ms, flops = kt[DOMINANT]
    roof = None
    if launches:
        achieved = flops / (ms * 1e-3) / 1e12
        traffic, traffic_source = pmc_traffic(args.precision, min(n, args.chunk if args.chunk else 4096))
        peak = PEAK_TFLOPS[args.precision]
        roof = {"bound": "mfma", "kernel": DOMINANT, "achieved": round(achieved, 2), "peak": round(peak, 1),
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                "traffic_source": traffic_source,
                "peak_basis": PEAK_NOTE[args.precision],
                "launches": launches, "avg_launch_ms": round(ms / launches, 4),
                "flop_per_launch": flops / launches}
        # The same launches against the OTHER roof.  A 3x3 64->64 layer moves, per output pixel, 64 channels in + 64 out + (every second
        # launch: the residual block's second convolution) 64 residual, at ACT_BYTES per element in this datapath's activation format;
        # 73 728 FLOP per pixel.  On the f16x3 path the two fractions are about equal: the class sits on the ridge of both roofs.
        act_bytes = {"f16x3": 4, "bf16x6": 6, "fp32": 4}[args.precision]
        alg_bytes = flops / launches / 73728.0 * 64 * act_bytes * 2.5
        t_launch = ms / launches * 1e-3
        roof["hbm_view"] = {"algorithmic_bytes_per_launch": round(alg_bytes), "achieved": round(alg_bytes / t_launch / 1e9, 1),
                            "traffic_rate": round(traffic / t_launch / 1e9, 1) if traffic else None, "peak": 8000, "unit": "GB/s",
                            "frac": round(alg_bytes / t_launch / 8e12, 4), "copy_bandwidth_from_profiles_r03_notes": 5110,
                            "note": "bytes = pixels x 64 channels x %d B x (in + out + residual on every second launch); the PMC traffic is lower "
                                    "than that where the residual still sits in L2 / MALL; a device-to-device copy sustained 5110 GB/s on "
                                    "this pool in round 3 (tools/hbm_probe.py, profiles/r03_notes.txt section 3: a recorded figure, not re-measured by this run)" % act_bytes}

    if args.breakdown and rank == 0:
        eng.ktime_enable(0xFFFF)
        for _ in range(3):
            step()
        torch.cuda.synchronize(dev)
        tot = 0.0
        per = 1024.0 / n / 3                                   # reported per 1024 blocks (one library chunk), as in DESIGN.md
        for k, (ln, kms, fl) in eng.ktime().items():
            tot += kms
            log("  %-20s launches %6.2f  %9.3f ms per 1024 blocks  %7.2f TFLOP/s" % (k, ln * per, kms * per, (fl / (kms * 1e-3) / 1e12) if kms else 0))
        log("  sum of kernel time %.3f ms per 1024 blocks (step = %d blocks)" % (tot * per, n))
        eng.ktime_enable(0)

    extras = {}
    if rank == 0 and n_gpus == 1 and not args.no_extras:
        extras = measure_extras(eng, args, dev, n, y, u, v, step)

    if rank == 0:
        out = {
            "metric": "CTUs/sec (%s QT+MTT inference+post-proc)" % args.comp.lower(), "value": round(blocks_per_s / 4.0, 2), "unit": "CTU/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else args.precision, "data": "synthetic",
            "config": {"workload": "%s QT+MTT nets QP%d, batch=%g synthetic 128x128 CTUs = %d 64x64 blocks (68x68 u8 inputs) per GPU, "
                                   "device-resident infer+Map2Partition" % (args.comp, args.qp, n / 4.0, n),
                       "blocks_per_gpu": n, "global_blocks": n * n_gpus, "parallelism": "dp%d (blocks sharded, gather of split flags to rank 0)" % n_gpus,
                       "weights": "QT real (reference trained_models), MTT synthetic seed=qp",
                       "datapath": {"fp32": "fp32 MFMA",
                                    "bf16x6": "bf16 MFMA, every fp32 operand split into 3 bf16 terms, 6 products, fp32 accumulate (fp32-equivalent logits)",
                                    "f16x3": "fp16 MFMA, every fp32 operand split into 2 fp16 terms (weights pre-scaled by 2^k), 3 products, "
                                             "fp32 accumulate (fp32-equivalent logits)"}[args.precision]},
            "blocks_per_s": round(blocks_per_s, 1),
            "net_tflops": round(blocks_per_s * FLOP_PER_BLOCK[args.comp] / 1e12, 2),
            "roofline": roof,
            "sensors": sens,
        }
        if multi:
            out["multi_gpu"] = multi
            out["rccl_ranks"] = multi["rccl_ranks"]
        out.update(extras)
        if args.cpu_sample > 0 and n_gpus == 1 and args.comp == "Luma":
            out["cpu_baseline"], out["parity"] = cpu_baseline(args.cpu_sample, 1, eng)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), file=json_out, flush=True)
    eng.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
