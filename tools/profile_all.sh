#!/bin/bash
# Run on the GPU box from the repo root:  bash tools/profile_all.sh r03
# The round's committed profiles: bench.py's configuration (C3) through tools/profile.sh, then one frame of C2 / C4 / C5
# through tools/profile_cfg.sh; text summaries + limiter JSONs (with their config block and frame count) land in
# gpurun_out/profiles_<tag>/ — copy them to profiles/.
set -u
TAG=${1:-r03}
DST=$PWD/gpurun_out/profiles_$TAG
mkdir -p "$DST"
bash tools/profile.sh ${TAG}_bench > "$DST/passes_bench.txt" 2>&1 || { echo "bench profile failed"; tail -5 "$DST/passes_bench.txt"; exit 1; }
python3 tools/profile_summary.py gpurun_out/prof_${TAG}_bench > "$DST/${TAG}_bench_spp64_summary.txt"
python3 tools/profile_summary.py gpurun_out/prof_${TAG}_bench --limiters "$DST/${TAG}_limiters.json" --frames 1 \
  --config '{"scene": "bistro_like", "detail": 1.0, "width": 1920, "height": 1080, "spp": 64, "bounces": 16, "flags": 1, "seed": 1, "world": 1}' \
  --source "profiles/${TAG}_bench_spp64_summary.txt (tools/profile.sh ${TAG}_bench, one stream (PTX_OVERLAP_SHADOW=0): rocprofv3 --kernel-trace --stats + separate --pmc passes over python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-order-check --spp 64; uncounted kernels = the one timed frame)" > /dev/null
grep -h '"metric"' gpurun_out/prof_${TAG}_bench/trace.log | tail -1 > "$DST/${TAG}_bench_line_under_rocprof.json"
rm -rf gpurun_out/prof_${TAG}_bench/*/   # raw CSVs: tens of MB
for CFG in C2 C4 C5; do
  bash tools/profile_cfg.sh ${TAG}_$CFG $CFG uniform > "$DST/passes_$CFG.txt" 2>&1 || { echo "$CFG profile failed"; tail -5 "$DST/passes_$CFG.txt"; continue; }
  python3 tools/profile_summary.py gpurun_out/prof_${TAG}_$CFG > "$DST/${TAG}_${CFG}_summary.txt"
  python3 tools/profile_summary.py gpurun_out/prof_${TAG}_$CFG --limiters "$DST/${TAG}_${CFG}_limiters.json" --frames 3 \
    --config "{\"config\": \"$CFG\", \"command\": \"tools/frame_breakdown.py $CFG uniform\", \"note\": \"3 frames of the config's uniform passes per pass of the profiler (2 warm-up + unprofiled-wall frame, then the HIP-event frame): sums and launch counts cover 3 frames, averages are per launch\"}" \
    --source "profiles/${TAG}_${CFG}_summary.txt (tools/profile_cfg.sh ${TAG}_$CFG $CFG uniform, one stream)" > /dev/null
  rm -rf gpurun_out/prof_${TAG}_$CFG/*/
done
echo done
