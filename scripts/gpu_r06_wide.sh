#!/bin/bash
# Round 6, the wide matrix-core path of the IVF-PQ search (ivf_pq_wide.hip): the evidence files under profiles/r06_wide_*
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash scripts/gpu_r06_wide.sh'
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
F='amdgpu.ids|^\[pq_scan3\] (overflow)|^\[pq_wide'
{
  for pd in 384 192 64; do timeout 300 python scripts/wide_search_bench.py 1000000 1024 $pd 32 10000 10 2>&1 | grep -v -E "$F"; done
  timeout 300 python scripts/wide_search_bench.py 1000000 1024 384 32 2000 10 2>&1 | grep -v -E "$F"
  timeout 300 python scripts/wide_search_bench.py 1000000 1024 384 32 10000 100 2>&1 | grep -v -E "$F"
  timeout 300 python scripts/wide_search_bench.py 1000000 1024 384 32 10000 10 f32 f32 2>&1 | grep -v -E "$F"
} > $O/r06_wide_search_1m768.log
timeout 300 python scripts/wide_shape_stats.py 2000000 2>&1 | grep -v -E "$F" | uniq > $O/r06_wide_cagra_build_shape_2m.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_w -o w -- python $GRAFT_REPO_ROOT/scripts/wide_search_bench.py 1000000 1024 384 32 10000 10 > /dev/null 2>&1)
python scripts/trace_last_search.py /tmp/prof_w | grep -v -E "at::native|copyBuffer" > $O/r06_wide_timeline_1m768_pq384.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cb -o cb -- python $GRAFT_REPO_ROOT/scripts/cagra_build_profile.py 10000000 2>&1 | grep -E "^build" > $O/r06_cagra_build_10m768.log)
python - <<PY
import csv, glob
f = glob.glob("/tmp/prof_cb/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("$O/r06_cagra_build_10m768_kernel_stats.csv", "w") as o:
    o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
    for r in rows[:30]:
        o.write('"%s",%s,%s,%s,%s\n' % (r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
PY
