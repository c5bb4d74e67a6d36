#!/bin/bash
# First contact with a real multi-GPU node, in the order that fails cheapest first (every step under its own timeout, so a hung
# collective costs minutes, not the lease):
#   1. devices visible, xGMI topology
#   2. RCCL world-N unit collectives of shard_comm.hip against numpy (scripts/rccl_collectives_check.py), N = 2, 4, 8
#   3. the world-N worker of tests/test_list_shard_world2_gpu.py with every rank on its OWN device over RCCL instead of the
#      host-staged transport (CUVS_AMD_WORLD_OWN_DEVICES=1)
#   4. bench.py --gpus 1 / 2 / 4 / 8, weak (N x 10k queries per step) and strong (one 10k batch over N list shards)
# Usage: scripts/gpu_first_8gpu.sh [rows]     (rows: corpus of step 4, default 100000000; outputs under gpurun_out/first_8gpu/)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
ROWS=${1:-100000000}
OUT=gpurun_out/first_8gpu
mkdir -p "$OUT"
NDEV=$(python -c 'import torch; print(torch.cuda.device_count())')
echo "[first-contact] $NDEV devices visible" | tee "$OUT/summary.txt"
(rocm-smi --showtopo 2>/dev/null || true) > "$OUT/topology.txt"
fail=0
for N in 2 4 8; do
  [ "$NDEV" -ge "$N" ] || { echo "[first-contact] skip world $N: only $NDEV devices" | tee -a "$OUT/summary.txt"; continue; }
  PORT=$((29500 + N))
  if timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
       scripts/rccl_collectives_check.py > "$OUT/collectives_world$N.log" 2>&1; then
    echo "[first-contact] world $N: RCCL collectives OK" | tee -a "$OUT/summary.txt"
  else
    echo "[first-contact] world $N: RCCL collectives FAILED (rc $?, see $OUT/collectives_world$N.log) - stopping before the larger steps" | tee -a "$OUT/summary.txt"
    fail=1; break
  fi
done
[ "$fail" = 0 ] || exit 1
if [ "$NDEV" -ge 2 ]; then
  if CUVS_AMD_WORLD_OWN_DEVICES=1 timeout 1500 python -m pytest tests/test_list_shard_world2_gpu.py -x -q -m gpu > "$OUT/world_tests_rccl.log" 2>&1; then
    echo "[first-contact] world 2 / 3 tests over RCCL (own devices): green" | tee -a "$OUT/summary.txt"
  else
    echo "[first-contact] world 2 / 3 tests over RCCL: FAILED (see $OUT/world_tests_rccl.log)" | tee -a "$OUT/summary.txt"; exit 1
  fi
fi
for N in 1 2 4 8; do
  [ "$NDEV" -ge "$N" ] || continue
  for MODE in weak strong; do
    [ "$N" = 1 ] && [ "$MODE" = strong ] && continue
    EXTRA="--no-variants --no-extras --no-pmc"; [ "$N" = 1 ] && EXTRA="--no-extras"
    if timeout 1500 python bench.py --gpus "$N" --scaling "$MODE" --rows "$ROWS" --steps 20 --warmup 3 $EXTRA > "$OUT/bench_n${N}_$MODE.json" 2> "$OUT/bench_n${N}_$MODE.log"; then
      tail -1 "$OUT/bench_n${N}_$MODE.json" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("[first-contact] N=%d %s: %.0f q/s, %.3f ms per step, recall %.4f" % (d["n_gpus"], d["scaling"], d["value"], d["ms_per_step"], d["recall_at_10"]))' | tee -a "$OUT/summary.txt"
    else
      echo "[first-contact] bench N=$N $MODE FAILED (see $OUT/bench_n${N}_$MODE.log)" | tee -a "$OUT/summary.txt"
    fi
  done
done
