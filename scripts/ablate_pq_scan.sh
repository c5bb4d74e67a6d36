cd $GRAFT_REPO_ROOT
for d in 0 1 2 4 6 7; do echo "dbg=$d"; CUVS_AMD_SCAN_DEBUG=$d python bench.py --rows 10000000 --n-lists 2048 --n-probes 32 --steps 3 --warmup 1 --no-cpu-baseline --gt-queries 10 --refine-ratio ${RATIO:-1} 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['roofline']['avg_launch_ms'], j['ms_per_step'])"; done
