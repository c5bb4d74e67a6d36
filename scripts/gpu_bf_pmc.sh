#!/bin/bash
# PMC pass (counters only) over the brute-force threshold-append kernel on a 2M x 128 / 10k search
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/bfprof
cat > /tmp/bf2m.py <<'PY'
import sys, os, torch, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench, cuvs_amd
from cuvs_amd.neighbors import brute_force
res = cuvs_amd.common.Resources(); dev = torch.device("cuda:0")
x2 = bench.gen_rows(2_000_000, 128, 1234, dev); q2 = bench.gen_rows(10000, 128, 4321, dev)
idx2 = brute_force.build(x2, resources=res)
for _ in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    brute_force.search(idx2, q2, 10, resources=res); res.sync(); torch.cuda.synchronize()
    print("ms", (time.perf_counter() - t) * 1e3)
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/bfpmc$i
  (cd /tmp && timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/bfpmc$i -o pmc -- python /tmp/bf2m.py > $GRAFT_REPO_ROOT/gpurun_out/bfprof/pmc$i.log 2>&1)
  f=$(find /tmp/bfpmc$i -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
s = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if "dist_mfma_kernel" in r["Kernel_Name"] and ", 2, " in r["Kernel_Name"]:
        s[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in s: print(k, s[k], "dispatches", n[k])
PY
done
grep "^ms" gpurun_out/bfprof/pmc1.log
