#!/bin/bash
# bash tools/profile_bench_only.sh r03 : the bench.py half of tools/profile_all.sh
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
rm -rf gpurun_out/prof_${TAG}_bench/*/
echo done
