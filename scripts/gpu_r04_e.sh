#!/bin/bash
# round 4: fused coarse search (tests + timing), inner-product regression check of the filter
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -X faulthandler -m pytest tests/test_ivf_pq_gpu.py tests/test_bench_shapes_gpu.py tests/test_fuzz_gpu.py tests/test_list_shard_gpu.py tests/test_brute_force_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -x > gpurun_out/r04e_tests.log 2>&1
echo "tests rc=$?"; grep "passed\|failed\|rror" gpurun_out/r04e_tests.log | tail -8
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o kt -- python scripts/pq_scan_variants.py --steps 10 "F4=1,LUT=f16,ACC=f32" > gpurun_out/r04e_kt.log 2>&1
find /tmp/prof_e -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04e_kernel_stats.csv \;
python - <<'P'
import csv
rows=list(csv.reader(open('gpurun_out/r04e_kernel_stats.csv')))
for x in rows[1:]:
    c=int(x[1])
    if c % 12 == 0 and c <= 120 and float(x[2])/12/1e6 > 0.015:
        print(f"{float(x[2])/12/1e6:8.3f} ms/search x{c//12:2d}  {x[0][:110].replace('cuvs_amd::(anonymous namespace)::','')}")
P
timeout 600 python scripts/pq_scan_variants.py --steps 5 --metric inner_product "F4=0,LUT=f16,ACC=f32" "F4=1,LUT=f16,ACC=f32" "F4=1,DBG=1024,LUT=f16,ACC=f32" > gpurun_out/r04e_ip.log 2>&1
grep -v "^\[bench\]" gpurun_out/r04e_ip.log | tail -8
