"""VGPRs / AGPRs / SGPRs / scratch / occupancy / static LDS of every kernel of csrc/*.hip (hipcc -Rpass-analysis=kernel-resource-usage;
no GPU needed):  python scripts/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import glob
import os
import re
import subprocess

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SRC = os.path.join(ROOT, "tianshou_amd", "csrc")
print("kernel resource usage at HEAD (hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage; static LDS only --\n"
      "the fused kernels take their LDS dynamically, see DESIGN.md 4.x)\n"
      "file | kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | occupancy waves/SIMD | static LDS B\n")
total = 0
for f in sorted(glob.glob(os.path.join(SRC, "*.hip"))):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                          "-I", SRC, "-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/tmp/_res.o"],
                         capture_output=True, text=True, cwd="/tmp").stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?:\s*\[[^\]]*\])?:\s+(\S+)\s+\[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": v}
        else:
            cur[k] = v
        if k == "LDS Size":
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(anonymous namespace\)::", "", name)
            name = re.sub(r"\(.*", "", name)[:90]
            print(" | ".join([os.path.basename(f)[:-4], name, cur.get("VGPRs", "?"), cur.get("AGPRs", "?"), cur.get("TotalSGPRs", "?"),
                              cur.get("ScratchSize", "?"), cur.get("Occupancy", "?"), v]))
            total += 1
print(f"\n{total} kernels")
