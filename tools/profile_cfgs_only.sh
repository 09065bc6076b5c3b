#!/bin/bash
# bash tools/profile_cfgs_only.sh r03 C2 C4 C5 : the per-config half of tools/profile_all.sh
set -u
TAG=${1:-r03}; shift
DST=$PWD/gpurun_out/profiles_$TAG
mkdir -p "$DST"
for CFG in "$@"; do
  bash tools/profile_cfg.sh ${TAG}_$CFG $CFG uniform > "$DST/passes_$CFG.txt" 2>&1 || { echo "$CFG profile failed"; tail -5 "$DST/passes_$CFG.txt"; continue; }
  python3 tools/profile_summary.py gpurun_out/prof_${TAG}_$CFG > "$DST/${TAG}_${CFG}_summary.txt"
  python3 tools/profile_summary.py gpurun_out/prof_${TAG}_$CFG --limiters "$DST/${TAG}_${CFG}_limiters.json" --frames 3 \
    --config "{\"config\": \"$CFG\", \"command\": \"tools/frame_breakdown.py $CFG uniform\", \"note\": \"3 frames of the uniform passes per profiler pass (warm-up, unprofiled-wall frame, HIP-event frame; for an adaptive config the warm-up also runs one adaptive phase): sums and launch counts cover all of that, averages are per launch\"}" \
    --source "profiles/${TAG}_${CFG}_summary.txt (tools/profile_cfg.sh ${TAG}_$CFG $CFG uniform, one stream)" > /dev/null
  rm -rf gpurun_out/prof_${TAG}_$CFG/*/
  echo "$CFG done"
done
