#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests/test_ivf_pq_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -p no:cacheprovider -x -k "coarse or k_up_to_256 or c3_shape_lists" 2>&1 | tail -15
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o kt -- python $GRAFT_REPO_ROOT/scripts/coarse_dtype_profile.py 8192 > $GRAFT_REPO_ROOT/gpurun_out/r04k_coarse.log 2>&1)
find /tmp/prof_k -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04k_kernel_stats.csv \;
grep "^coarse" gpurun_out/r04k_coarse.log
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r04k_kernel_stats.csv')):
    n=r['Name']
    if any(s in n for s in ('coarse','dist_tile','select_k','lowp','round_to','scale_to')):
        print(n[:110], r['Calls'], r['AverageNs'])
PY
