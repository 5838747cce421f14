#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table of a HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py graphnet_amd/csrc/edgeconv_v2.hip [name-substring]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                      "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                     stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": subprocess.run(["c++filt", t.split(": ", 1)[1]], stdout=subprocess.PIPE, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print(f"{'kernel':70s} {'sgpr':>5s} {'vgpr':>5s} {'agpr':>5s} {'scratch':>8s} {'occ':>4s} {'lds':>7s}")
for r in rows:
    n = re.sub(r"\(.*", "", r["name"]).replace("void ", "").replace("gn::", "")
    if flt in n:
        print(f"{n[:70]:70s} {r.get('TotalSGPRs','?'):>5s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} "
              f"{r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('LDS Size [bytes/block]','?'):>7s}")
