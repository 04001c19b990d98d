"""Summarise rocprofv3 CSV output (kernel stats + PMC passes) into small text/JSON files for profiles/.
Usage: python tools/summarize_profile.py gpurun_out/prof_<tag> profiles/<name>"""
import csv, glob, json, os, re, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    m = re.search(r"(conv_mfma_kernel)<(\d+), ?(\d+), ?(\d+)>", name)
    if m:
        return "conv_mfma<%s,%s,NT=%s>" % m.groups()[1:]
    m = re.search(r"(conv_x6_kernel)<(\d+), ?(\d+), ?(\d+)", name)
    if m:
        return "conv_x6_kernel<%s, %s, %s>" % m.groups()[1:]
    m = re.search(r"(conv_h2_kernel)<(\d+), ?(\d+), ?(\d+), ?(\w+)(?:, ?(\d+), ?(\w+))?(?:, ?(\w+))?", name)
    if m:   # product: <KH, KW, NT, SC[, LEAN]>; notebook (tools/abl): <KH, KW, NT, SC, ABL, LEAN, W8>.  LEAN = the 168-VGPR three-workgroups-per-CU form
        lean = m.group(7) if m.group(6) is not None else m.group(8)
        return "conv_h2_kernel<%s, %s, %s, sc=%s>%s" % (m.group(2), m.group(3), m.group(4), m.group(5), " 3wg" if lean in ("true", "1") else "")
    name = name.replace("void ", "").replace("pmp::", "").replace("(anonymous namespace)::", "")
    return name[:70] or "?"


lines = []
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    lines.append("# rocprofv3 --kernel-trace --stats (bench.py --steps 3 --warmup 1): per-kernel summary")
    lines.append("%-44s %8s %14s %12s %8s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for r in rows:
        lines.append("%-44s %8s %14.3f %12.2f %8s" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                      float(r["AverageNs"]) / 1e3, r["Percentage"]))
pmc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        pmc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
if pmc:
    lines.append("")
    lines.append("# PMC counters (separate rocprofv3 --pmc passes), averaged per launch")
    for k in sorted(pmc, key=lambda k: -pmc[k].get("SQ_WAVE_CYCLES", 0)):
        lines.append(k)
        for c in sorted(pmc[k]):
            lines.append("    %-32s %18.1f  (%d launches)" % (c, pmc[k][c] / cnt[k][c], cnt[k][c]))
open(dst + ".txt", "w").write("\n".join(lines) + "\n")
json.dump({k: {c: pmc[k][c] / cnt[k][c] for c in pmc[k]} for k in pmc}, open(dst + "_pmc.json", "w"), indent=1)
print("\n".join(lines[:60]))
