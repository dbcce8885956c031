#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy report for one csrc/*.hip file (hipcc remarks).
    python tools/kres.py ln-asr_amd/csrc/gmm.hip [filter-substring]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on",
                    "-I/opt/rocm/include", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                   capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "").replace("void lnasr::", "")
        name = re.sub(r"\((?!anonymous).*", "", name).replace("lnasr::", "")
        cur = {"name": name}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"),
                     ("vspill", r"VGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("sgpr", r" SGPRs: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print("%-58s %5s %5s %5s %6s %6s %4s %7s" % ("kernel", "vgpr", "agpr", "sgpr", "vspill", "sspill", "occ", "lds"))
for c in rows:
    if flt in c["name"]:
        print("%-58s %5d %5d %5d %6d %6d %4d %7d" % (c["name"][:58], c.get("vgpr", -1), c.get("agpr", -1), c.get("sgpr", -1),
              c.get("vspill", -1), c.get("sspill", -1), c.get("occ", -1), c.get("lds", -1)))
if r.returncode:
    print(r.stderr[-3000:])
