#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${1:-r05n}
timeout 1200 python -m pytest tests/test_coarse_grouped_gpu.py tests/test_bench_shapes_gpu.py tests/test_brute_force_gpu.py tests/test_ivf_pq_gpu.py tests/test_cagra_gpu.py tests/test_serialize_format_gpu.py tests/test_serialize_filter_gpu.py tests/test_list_shard_world2_gpu.py "tests/test_reference_tables_gpu.py::test_ivf_pq_flat_layout_codes" -q --timeout 600 -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1
echo "rc=$?"; grep -E "passed|failed" gpurun_out/${T}_tests.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/${T}_tests.log | cut -c1-260 | head -40
timeout 900 python bench.py --steps 20 --warmup 5 --no-extras --no-pmc --rows 100000000 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/${T}_bench.err | cut -c1-200; grep '^{"metric"' gpurun_out/${T}_bench.json | cut -c1-400
python - <<'P'
import json,sys
for line in open(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/'+__import__('os').environ.get('T','r05n')+'_bench.json'):
    if line.startswith('{"metric"'):
        j=json.loads(line); print(j["ms_per_step"], j["scan3_equals_lut_scan"], j["roofline"]["scan_kernels"]["phase_ms_per_step"]); print(j["config"]["batch_sweep"])
P
