#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cat > /tmp/prd.py <<'PY'
import sys, os, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench, cuvs_amd
from cuvs_amd.neighbors import cagra
res = cuvs_amd.common.Resources(); dev = torch.device("cuda:0")
x = bench.gen_rows(500_000, 64, 7, dev, latent=16, n_modes=1)
idx = cagra.build(cagra.IndexParams(intermediate_graph_degree=128, graph_degree=64), x, resources=res); res.sync()
PY
for d in 0; do
  rm -rf /tmp/prp; (cd /tmp && CUVS_AMD_PRUNE_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prp -o p -- python /tmp/prd.py > /dev/null 2>&1)
  f=$(find /tmp/prp -name "*kernel_stats.csv" | head -1); python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'prune_kernel' in r['Name']: print('dbg $d: prune_kernel ms', float(r['TotalDurationNs'])/1e6)
"
done
