#!/usr/bin/env python3
"""register / scratch use of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage):
    python tools/kres.py part_hash_inst.hip -DBTLBF_PART_H=4 [filter-substring]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "btl_bloomfilter_amd", "csrc", sys.argv[1])
flags = [a for a in sys.argv[2:] if a.startswith("-")]
filt = [a for a in sys.argv[2:] if not a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
       "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/tmp/kres.o", src] + flags
r = subprocess.run(cmd, capture_output=True, text=True)
open("/tmp/kres.last", "w").write(r.stderr)
rows, cur = [], None
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("btlbf::", "").replace("void ", "")}
        rows.append(cur)
    for key, pat in (("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("sgpr", r"SGPRs: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None and " " + key.upper()[:1] in " " + line.upper():
            cur[key] = int(m.group(1))
if r.returncode:
    print(r.stderr[-3000:])
for row in rows:
    if all(f in row["name"] for f in filt):
        print("%-90s vgpr %3d  scratch %4d  sgpr %3d" % (row["name"][:90], row.get("vgpr", -1), row.get("scratch", -1), row.get("sgpr", -1)))
