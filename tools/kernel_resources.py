"""per-kernel registers / spills / occupancy of one csrc file as the compiler reports them (no GPU needed):
    kernel_resources.py <file.hip> [substring filter] [extra hipcc flags ...]"""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.abspath(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", f"-I{REPO}/include", "-ffp-contract=off",
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
err = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(src))).stderr
cur = None
rows = {}
for ln in err.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", ln)
    if not m:
        continue
    if m.group(1) == "Function Name":
        cur = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
    elif cur:
        rows[cur][m.group(1).split(" [")[0]] = m.group(2)
for k, r in rows.items():
    if flt in k:
        name = re.sub(r"\(.*$", "", k.replace("void ", ""))
        print(f"{name:60s} VGPR {r.get('VGPRs'):>4s} AGPR {r.get('AGPRs'):>3s} scratch {r.get('ScratchSize'):>4s} "
              f"spill {r.get('VGPRs Spill'):>3s} waves/SIMD {r.get('Occupancy')} LDS {r.get('LDS Size')}")
