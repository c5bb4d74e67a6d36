#!/bin/bash
# kernel-level profile of the 10M x 128 / 10k brute-force search
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/bfprof
cat > /tmp/bf10m.py <<'PY'
import sys, torch, time
import os; sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench, cuvs_amd
from cuvs_amd.neighbors import brute_force
res = cuvs_amd.common.Resources(); dev = torch.device("cuda:0")
x2 = bench.gen_rows(10_000_000, 128, 1234, dev); q2 = bench.gen_rows(10000, 128, 4321, dev)
idx2 = brute_force.build(x2, resources=res)
for _ in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    brute_force.search(idx2, q2, 10, resources=res); res.sync(); torch.cuda.synchronize()
    print("ms", (time.perf_counter() - t) * 1e3)
PY
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bfprof -o bf -- python /tmp/bf10m.py > $GRAFT_REPO_ROOT/gpurun_out/bfprof/run.log 2>&1
cd $GRAFT_REPO_ROOT; grep -E "^ms|Error|error" gpurun_out/bfprof/run.log | head; find /tmp/bfprof -name "*kernel_stats.csv" -exec cp {} gpurun_out/bfprof/kernel_stats.csv \;
head -12 gpurun_out/bfprof/kernel_stats.csv | cut -c1-220
