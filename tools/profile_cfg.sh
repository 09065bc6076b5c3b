#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:  bash tools/profile_cfg.sh <tag> <config> [part]
# rocprofv3 kernel trace + SQ/TCC/FETCH/WRITE passes over tools/frame_breakdown.py <config> <part> (one BASELINE config's frame).
set -u
TAG=$1; CFG=$2; PART=${3:-uniform}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export PTX_OVERLAP_SHADOW=0      # one stream: every kernel's duration is its own (see tools/profile.sh)
run() { # name, then rocprofv3 options
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 "$@" -d "$OUT/$name" --output-format csv -- python3 tools/frame_breakdown.py "$CFG" "$PART" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a "$OUT/passes.txt"
  return $rc
}
run trace --kernel-trace --stats || exit 1
run pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY || exit 1
run pmc_sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM || exit 1
run pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum || exit 1
run pmc_grbm --kernel-trace --pmc GRBM_GUI_ACTIVE || exit 1      # the clock the kernels actually ran at: sum over 8 XCDs / 8 / kernel time
run pmc_fetch --pmc FETCH_SIZE || exit 1
run pmc_write --pmc WRITE_SIZE || exit 1
echo done
