#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests/test_bench_shapes_gpu.py tests/test_ivf_pq_gpu.py tests/test_fuzz_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_o -o kt -- python $GRAFT_REPO_ROOT/scripts/pq_scan_variants.py --steps 10 "F4=1,LUT=f16,ACC=f32" > $GRAFT_REPO_ROOT/gpurun_out/r04o_kt.log 2>&1)
find /tmp/prof_o -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04o_kernel_stats.csv \;
grep -v "^\[bench\]" gpurun_out/r04o_kt.log | grep "search" | tail -2
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r04o_kernel_stats.csv')):
    n=r['Name']
    if any(s in n for s in ('pq_head','ov_count','ov_fill','pq_filter4','select_k_minima','pool_merge','pq_rescore')):
        print(n[38:100], r['Calls'], r['AverageNs'])
PY
