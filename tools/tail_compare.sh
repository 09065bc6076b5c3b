#!/bin/bash
# bash tools/tail_compare.sh : frame wall time with and without the tail kernel (full frame, an eighth of it, C1, C5)
# -> gpurun_out/tail_compare.log
mkdir -p gpurun_out
out=gpurun_out/tail_compare.log
: > $out
run() {   # label, env assignments..., -- config part
  local label=$1; shift
  echo "== $label" | tee -a $out
  env "$@" 2>&1 | grep -E "unprofiled|\"part\"" | tee -a $out
  return ${PIPESTATUS[0]}
}
for tk in 0 1; do
  run "C3 full tail=$tk" PTX_TAIL_KERNEL=$tk timeout -k 10 150 python3 tools/frame_breakdown.py C3 uniform || exit 1
  run "C3 1/8 tail=$tk" PTX_TAIL_KERNEL=$tk PT_PARTITION=0,8 timeout -k 10 150 python3 tools/frame_breakdown.py C3 uniform || exit 1
  run "C1 tail=$tk" PTX_TAIL_KERNEL=$tk timeout -k 10 150 python3 tools/frame_breakdown.py C1 uniform || exit 1
done
