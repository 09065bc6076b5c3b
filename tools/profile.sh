#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:  bash tools/profile.sh <tag> [bench args...]
# Leaves rocprofv3 CSVs under gpurun_out/prof_<tag>/ ; summarise with tools/profile_summary.py.
# Each counter pass is its own run (rocprofv3 does not split counters over passes; see the block limits).
set -u
TAG=${1:-r01}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# one stream: every kernel's duration is then its own (by default the any-hit launch shares the GPU with the next
# closest-hit launch and its dispatch time includes waiting for wave slots; k_extend's own duration is the same either way)
export PTX_OVERLAP_SHADOW=0
# --no-order-check: the untimed reference-order frame would add a second frame of k_shade / k_shadow launches to the sums
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-order-check --spp 64 $*"
run() { # name, then rocprofv3 options
  local name=$1; shift
  timeout -s KILL 420 rocprofv3 "$@" -d "$OUT/$name" --output-format csv -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a "$OUT/passes.txt"
  return $rc
}
run trace --kernel-trace --stats || exit 1
run pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY || exit 1
run pmc_sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM || exit 1
run pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum || exit 1
run pmc_tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum || exit 1
run pmc_grbm --kernel-trace --pmc GRBM_GUI_ACTIVE || exit 1      # the clock the kernels actually ran at: sum over 8 XCDs / 8 / kernel time
run pmc_fetch --pmc FETCH_SIZE || exit 1
run pmc_write --pmc WRITE_SIZE || exit 1
echo done
