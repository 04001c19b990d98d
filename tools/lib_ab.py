"""Same-box A/B of two builds of libpmp_hip.so on the full luma step (run on the GPU box): alternates short bench.py runs of
each library (a fresh process per run) and prints step time + per-class kernel time.  Boxes of the pool differ by a few
percent, so only alternating runs on one box compare builds.
Usage: python tools/lib_ab.py <libA.so> <libB.so> [rounds] [extra bench.py args...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [os.path.abspath(p) for p in sys.argv[1:3]]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
extra = sys.argv[4:]
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--lib", l, "--steps", "10", "--warmup", "3", "--cpu-sample", "0",
                            "--no-extras", "--breakdown"] + extra, capture_output=True, text=True)
        if p.returncode != 0:
            print(p.stderr[-2000:]); raise SystemExit(1)
        d = json.loads(p.stdout.strip().splitlines()[-1])
        br = " | ".join(" ".join(x.split()[:1] + x.split()[3:4] + x.split()[8:9]) for x in p.stderr.splitlines() if x.startswith("  conv_") or x.startswith("  stem"))
        res[l].append(d["ms_per_step"])
        print("round %d %-28s %.3f ms/step  dominant %.1f TF | %s" % (r, os.path.basename(l), d["ms_per_step"], d["roofline"]["achieved"], br), flush=True)
for l in libs:
    v = sorted(res[l])
    print("%-28s median %.3f ms/step (min %.3f)" % (os.path.basename(l), v[len(v) // 2], v[0]))
