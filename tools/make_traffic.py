"""profiles/<name>_pmc.json -> profiles/pmc_traffic.json (HBM bytes per launch of bench.py's dominant kernel class).
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section) -> doubled.  The kernel template conv_mfma_kernel<3,3,4,..> also serves the few
3x3 32->64 launches of the attention trunks, so this is the average over a slightly wider set than the class."""
import hashlib, json, subprocess, sys
KERNEL_SOURCE = {"f16x3": "pmp_vvc_tip2023_amd/csrc/conv_f16x3.hip", "bf16x6": "pmp_vvc_tip2023_amd/csrc/conv_bf16x6.hip", "fp32": "pmp_vvc_tip2023_amd/csrc/conv_mfma.hip"}
src = json.load(open(sys.argv[1]))
out = {}
import os
if os.path.isfile("profiles/pmc_traffic.json"):
    out = json.load(open("profiles/pmc_traffic.json"))
only = sys.argv[3] if len(sys.argv) > 3 else None      # round 5: an f16x3 run also holds a few tiny fp32 launches (the activation-scale calibration):
for k, v in src.items():                                # python tools/make_traffic.py <pmc.json> <blocks per launch> f16x3 updates that entry alone
    mode = "fp32" if k.startswith("conv_mfma_kernel<3, 3, 4") else ("bf16x6" if k.startswith("conv_x6_kernel<3, 3, 4") else
                                                                      ("f16x3" if k.startswith(("conv_h2_kernel<3, 3, 4, sc=false", "conv_h2_kernel<3, 3, 4, sc=0")) else None))
    if mode and (only is None or mode == only) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        out["conv_mfma_3x3_c64:" + mode] = round((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0)
        out["_blocks_per_launch:" + mode] = int(sys.argv[2]) if len(sys.argv) > 2 else 4096   # blocks per launch of the profiled run (bench.py scales)
        # which build these bytes were measured on: bench.py prints `traffic` only while the kernel source it runs still has this hash
        # (round 6); run from the repo root right after the profile, before touching the kernel file
        try:
            head = subprocess.check_output(["git", "rev-parse", "HEAD"], text=True).strip()
        except Exception:      # noqa: BLE001
            head = None
        out["_build:" + mode] = {"kernel_source": KERNEL_SOURCE[mode], "kernel_sha256": hashlib.sha256(open(KERNEL_SOURCE[mode], "rb").read()).hexdigest(),
                                 "git_head": head, "pmc_summary": sys.argv[1]}
        out["_detail:" + mode] = {"kernel": k, "FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"],
                          "note": "bytes/launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, averaged over the launches of the kernel"}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(out)
