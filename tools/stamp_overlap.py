"""How the workgroups of the f16x3 3x3 64->64 kernel share a CU in time (diagnostic build, run on the GPU box).

  PMP_STAMP_DUMP=/tmp/stamps.bin python tools/conv_x6_bench.py h2 stamps && python tools/stamp_overlap.py /tmp/stamps.bin

The stamp build (ABL 128, conv_f16x3.hip) leaves 16 u64 per workgroup: prologue, K-steps, end-of-group stores, barriers,
accumulate total, epilogue, begin, end (shader-clock ticks of wave 0), HW_ID, XCC_ID.  Workgroups are grouped by
(XCC, SE/SH/CU) and their phases laid on the CU's time line."""
import sys
import numpy as np

d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16)
pro, acc, epi, beg, end = (d[:, i].astype(np.int64) for i in (0, 4, 5, 6, 7))
hw, xcc = d[:, 8].astype(np.int64), d[:, 9].astype(np.int64)
cu = (xcc & 0xF) * 256 + ((hw >> 8) & 0xFF)
ok = (beg > 0) & (end > beg)
if not ok.all():
    print("dropping %d workgroups without stamps" % int((~ok).sum()))
    pro, acc, epi, beg, end, hw, xcc, cu, d = pro[ok], acc[ok], epi[ok], beg[ok], end[ok], hw[ok], xcc[ok], cu[ok], d[ok]
print("workgroups %d on %d CUs; life %.0f ticks (prologue %.0f, accumulate %.0f, epilogue %.0f)" % (
    len(d), len(np.unique(cu)), (end - beg).mean(), pro.mean(), acc.mean(), epi.mean()))
res = np.zeros(4); kph = np.zeros(4); span_tot = 0.0; gaps = []; offs = []
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    idx = idx[np.argsort(beg[idx])]
    t0, t1 = beg[idx].min(), end[idx].max()
    ev = []   # (time, d_resident, d_kphase)
    for i in idx:
        ev += [(beg[i], 1, 0), (beg[i] + pro[i], 0, 1), (beg[i] + acc[i], 0, -1), (end[i], -1, 0)]
    ev.sort(key=lambda e: (e[0], -e[1], -e[2]))   # at equal times arrivals before departures
    r = k = 0; last = t0
    for t, dr, dk in ev:
        res[max(0, min(r, 3))] += t - last; kph[max(0, min(k, 3))] += t - last; last = t
        r += dr; k += dk
    span_tot += t1 - t0
    # slot hand-over gap: a workgroup's begin minus the latest end before it (when both slots were taken)
    ends = np.sort(end[idx])
    for i in idx[2:]:
        prev = ends[ends <= beg[i]]
        if len(prev): gaps.append(beg[i] - prev[-1])
    # phase offset between the two workgroups resident together: begin difference / life
    for a, b in zip(idx[:-1], idx[1:]):
        if beg[b] < end[a]: offs.append((beg[b] - beg[a]) / float(end[a] - beg[a]))
print("CU time with n workgroups resident:   " + "  ".join("%d: %.1f%%" % (n, 100 * res[n] / span_tot) for n in range(4)))
print("CU time with n workgroups in K-loops: " + "  ".join("%d: %.1f%%" % (n, 100 * kph[n] / span_tot) for n in range(4)))
gaps = np.array(gaps); offs = np.array(offs)
if len(gaps): print("begin - most recent end on the CU: median %.0f, p10 %.0f, p90 %.0f ticks" % (np.median(gaps), np.percentile(gaps, 10), np.percentile(gaps, 90)))
if len(offs):
    h, _ = np.histogram(offs, bins=10, range=(0, 1))
    print("start offset of consecutive co-resident workgroups / life, deciles: " + " ".join("%d" % v for v in h))
