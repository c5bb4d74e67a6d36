#!/bin/bash
# first GPU call of the next round: what round 3 could no longer measure (scratch cache of DESIGN 3.1d) + the full suite
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -X faulthandler -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r04a_tests.log 2>&1
echo "tests rc=$?"; grep "passed\|failed\|rror" gpurun_out/r04a_tests.log | tail -5
# host timelines with and without the cache: IVF-Flat at C2, enqueue / drain times of the three searches
timeout 120 python scripts/host_trace_flat.py 2>&1 | grep "host trace\|call\|back-to" | tail -4
CUVS_AMD_ALLOC_CACHE=0 timeout 120 python scripts/host_trace_flat.py 2>&1 | grep "back-to" | sed 's/^/cache off: /'
timeout 120 python scripts/host_enqueue_probe.py 2>&1 | grep " ms\| us" | tee gpurun_out/r04a_host_probe.log
# C1 / C2 lines (brute force small shape, IVF-Flat) stand-alone
timeout 300 python scripts/bench_other.py bf flat 2>&1 | grep '^{' | cut -c1-500 | tee gpurun_out/r04a_other.json
# the sharded path on one rank
timeout 300 python scripts/shard_overhead_probe.py --skip-plain 2>&1 | grep " ms " | tee gpurun_out/r04a_shard_probe.log
