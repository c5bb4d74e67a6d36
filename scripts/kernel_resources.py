"""Per-kernel register / scratch / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: python scripts/kernel_resources.py cuvs_amd/csrc/ivf_pq_search.hip [name filter] [extra hipcc flags...]"""
import re, subprocess, sys
src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Iinclude",
       "-Icuvs_amd/csrc", "-Wno-unused-result", "-ffp-contract=off", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: [^ ]+ *Function Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r"\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
for k, v in rows.items():
    if filt in k:
        name = re.sub(r"cuvs_amd::\(anonymous namespace\)::", "", k)
        name = re.sub(r"\(.*\)$", "", name).replace("void ", "")
        print(f"{name[:90]:90s} vgpr {v.get('VGPRs',0):4d} agpr {v.get('AGPRs',0):3d} sgpr {v.get('TotalSGPRs',0):4d} scratch {v.get('ScratchSize',0):4d} "
              f"vspill {v.get('VGPRs Spill',0):3d} sspill {v.get('SGPRs Spill',0):3d} occ {v.get('Occupancy',0)} lds {v.get('LDS Size',0)}")
