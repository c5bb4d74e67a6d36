#!/bin/bash
# kernel trace of the C1 brute-force search (100k x 128, 1000 queries, k = 10)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/c1prof
cat > /tmp/c1.py <<'PY'
import sys, os, torch, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench, cuvs_amd
from cuvs_amd.neighbors import brute_force
res = cuvs_amd.common.Resources(); dev = torch.device("cuda:0")
x = bench.gen_rows(100_000, 128, 1234, dev); q = bench.gen_rows(1000, 128, 4321, dev)
idx = brute_force.build(x, resources=res)
for _ in range(20):
    brute_force.search(idx, q, 10, resources=res)
res.sync(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50):
    brute_force.search(idx, q, 10, resources=res)
res.sync(); torch.cuda.synchronize()
print("ms", (time.perf_counter() - t) / 50 * 1e3)
PY
rm -rf /tmp/c1prof; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c1prof -o c1 -- python /tmp/c1.py > $GRAFT_REPO_ROOT/gpurun_out/c1prof/run.log 2>&1
cd $GRAFT_REPO_ROOT; grep "^ms" gpurun_out/c1prof/run.log; find /tmp/c1prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/c1prof/kernel_stats.csv \;
find /tmp/c1prof -name "*kernel_trace.csv" -exec cp {} gpurun_out/c1prof/kernel_trace.csv \;
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/c1prof/kernel_stats.csv")))
for r in rows[:14]: print(r["Name"][:80], r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3))
tr = list(csv.DictReader(open("gpurun_out/c1prof/kernel_trace.csv")))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
last = tr[-14:]
t0 = int(last[0]["Start_Timestamp"])
for r in last: print("%8.1f %8.1f %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Kernel_Name"][:70]))
PY
