#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:  bash tools/profile_trace.sh <tag> [scene detail]
# SQ counter passes over tests/gpu_trace_bench.py (k_trace_batch on primary and bounce rays of the bench scene).
set -u
TAG=${1:-t01}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
run() { # name, then rocprofv3 options
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 "$@" -d "$OUT/$name" --output-format csv -- python3 tests/gpu_trace_bench.py "${ARGS[@]}" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a "$OUT/passes.txt"
  return $rc
}
ARGS=("$@")
run pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY || exit 1
run pmc_sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM || exit 1
echo done
