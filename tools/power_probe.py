"""Socket power and clocks while the bench runs (run on the GPU box): samples rocm-smi next to a child `python bench.py`.  Is the part at
its power limit during the passes?  Usage: python tools/power_probe.py [bench args...]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMI = "/opt/rocm/bin/rocm-smi"


def sample():
    try:
        out = subprocess.run([SMI, "--showpower", "--showclocks", "--showmaxpower", "--showuse", "--showtemp", "--json"], capture_output=True, text=True, timeout=5).stdout
        d = json.loads(out)
        return d.get("card0", d)
    except Exception as e:
        return {"error": repr(e)}


idle = sample()
print("idle:", json.dumps(idle)[:1500], flush=True)
child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + (sys.argv[1:] or ["--steps", "100", "--warmup", "5", "--cpu-sample", "0", "--no-extras"]),
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
rows = []
t0 = time.time()
while child.poll() is None:
    s = sample()
    s["t"] = round(time.time() - t0, 2)
    rows.append(s)
    time.sleep(0.1)
out = child.stdout.read()
keys = sorted({k for r in rows for k in r if k != "t"})
print("samples: %d" % len(rows))
for k in keys:
    vals = []
    for r in rows:
        try:
            vals.append(float(str(r.get(k)).split()[0].strip("()MHzW%")))
        except Exception:
            pass
    if vals:
        vs = sorted(vals)
        print("  %-52s min %.1f  median %.1f  p90 %.1f  max %.1f" % (k, vs[0], vs[len(vs) // 2], vs[int(len(vs) * 0.9)], vs[-1]))
    else:
        print("  %-52s %s" % (k, rows[len(rows) // 2].get(k)))
print("timeline (t, power, sclk):")
for r in rows[::5]:
    print("   ", r["t"], {k: r[k] for k in r if "ower" in k or "sclk" in k or "use" in k.lower()})
print(out[-600:])
