"""Per-kernel register / spill / scratch table of one kernel source, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.
    python scripts/kernel_resources.py kernels/propagate_mfma.hip [name-regex] [extra hipcc flags ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.environ.get("CUSMC_CSRC", os.path.join(ROOT, "cusmc_amd", "csrc"))  # (override: another checkout, for A/B)
FLAGS = ("-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed "
         "-mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-atomic-optimizer-strategy=None "
         "-Rpass-analysis=kernel-resource-usage").split()
KEYS = r"(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\])"


def main():
    src, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    extra = sys.argv[3:]
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-c", src, "-o", "/tmp/kernel_resources.o"], cwd=CSRC,
                       capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
    cur, rows = None, []
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True,
                                          text=True).stdout.strip()}
            rows.append(cur)
            continue
        m = re.search(KEYS + r": (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    for c in rows:
        name = re.sub(r"^void cusmc::", "", c["name"])
        name = re.sub(r"\(.*$", "", name)
        if pat and not re.search(pat, name):
            continue
        print("%-62s vgpr %3d agpr %3d sgpr %3d spill %3d scratch %4d occ %d" % (
            name, c.get("VGPRs", -1), c.get("AGPRs", -1), c.get("SGPRs", -1), c.get("VGPRs Spill", -1),
            c.get("ScratchSize", -1), c.get("Occupancy", -1)))


if __name__ == "__main__":
    main()
