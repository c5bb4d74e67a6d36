#!/bin/bash
# round 4: parallel slow-path append (IP), asynchronous IVF-Flat tail (device-side guarded fallback): tests + timings
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r04g_tests.log 2>&1
echo "tests rc=$?"; grep "passed\|failed\|rror" gpurun_out/r04g_tests.log | tail -8
timeout 600 python scripts/pq_scan_variants.py --steps 5 --metric inner_product "F4=0,LUT=f16,ACC=f32" "F4=1,LUT=f16,ACC=f32" > gpurun_out/r04g_ip.log 2>&1
grep -v "^\[bench\]" gpurun_out/r04g_ip.log | tail -3
timeout 600 python scripts/pq_scan_variants.py --steps 5 "F4=0,LUT=f16,ACC=f32" "F4=1,LUT=f16,ACC=f32" > gpurun_out/r04g_l2.log 2>&1
grep -v "^\[bench\]" gpurun_out/r04g_l2.log | tail -3
timeout 300 python scripts/bench_other.py bf flat 2>&1 | grep '^{' | cut -c1-700 | tee gpurun_out/r04g_other.json
