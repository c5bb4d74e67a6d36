#!/bin/bash
# kernel stats of a CAGRA build (2M x 768 fp16)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/cagraprof
rm -rf /tmp/cgprof; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cgprof -o cg -- python $GRAFT_REPO_ROOT/scripts/bench_other.py cagra --cagra-rows 2000000 --cagra-latent 24 > $GRAFT_REPO_ROOT/gpurun_out/cagraprof/run.log 2>&1
cd $GRAFT_REPO_ROOT; find /tmp/cgprof -name "*kernel_stats.csv" -exec cp {} gpurun_out/cagraprof/kernel_stats.csv \;
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/cagraprof/kernel_stats.csv")))[:16]:
    print(r["Name"][:90].replace("cuvs_amd::(anonymous namespace)::", ""), r["Calls"], "total ms %.1f" % (float(r["TotalDurationNs"]) / 1e6), r["Percentage"])
PY
tail -1 gpurun_out/cagraprof/run.log | cut -c1-250
