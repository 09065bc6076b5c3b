#!/bin/bash
# Run on the GPU box from the repo root: the "G lanes per ray" experiment (PTX_GROUP_LANES) over whole frames of C3 / C4:
# HIP-event time per kernel class, then kernel trace + two SQ counter passes on C3.   bash tools/profile_group.sh <tag>
set -u
TAG=${1:-r03_group}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export PTX_OVERLAP_SHADOW=0
for G in 4 8; do
  export PTX_GROUP_LANES=$G
  for CFG in C3 C4; do
    echo "== G=$G $CFG" | tee -a "$OUT/frames.txt"
    timeout -k 10 200 python3 tools/frame_breakdown.py $CFG uniform 2>&1 | tail -1 | tee -a "$OUT/frames.txt"
  done
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$OUT/g$G/trace" --output-format csv -- python3 tools/frame_breakdown.py C3 uniform > "$OUT/g$G.trace.log" 2>&1 || exit 1
  timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d "$OUT/g$G/pmc_sq1" --output-format csv -- python3 tools/frame_breakdown.py C3 uniform > "$OUT/g$G.sq1.log" 2>&1 || exit 1
  timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d "$OUT/g$G/pmc_sq2" --output-format csv -- python3 tools/frame_breakdown.py C3 uniform > "$OUT/g$G.sq2.log" 2>&1 || exit 1
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d "$OUT/g$G/pmc_grbm" --output-format csv -- python3 tools/frame_breakdown.py C3 uniform > "$OUT/g$G.grbm.log" 2>&1 || exit 1
  python3 tools/profile_summary.py "$OUT/g$G" > "$OUT/g${G}_summary.txt" 2>&1
  python3 tools/profile_summary.py "$OUT/g$G" --limiters "$OUT/g${G}_limiters.json" --frames 3 > /dev/null 2>&1
done
unset PTX_GROUP_LANES
echo done
