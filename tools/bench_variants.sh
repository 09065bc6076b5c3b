#!/bin/bash
# On the GPU box: bench every library under rust-pathtracer_amd/lib/variants (tools/build_variant.sh) next to the shipped one,
# one stream, three frames each:  bash tools/bench_variants.sh [tag]  ->  gpurun_out/variants_<tag>.txt
TAG=${1:-v}
OUT=gpurun_out/variants_$TAG.txt
: > $OUT
for lib in "" rust-pathtracer_amd/lib/variants/*.so; do
  PT_VARIANT_LIB=$lib PTX_OVERLAP_SHADOW=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-order-check --steps 3 > gpurun_out/variant_tmp.json 2> gpurun_out/variant_tmp.err || { echo "$lib FAILED" >> $OUT; continue; }
  python - "$lib" >> $OUT <<'PY'
import json, sys
d = json.load(open("gpurun_out/variant_tmp.json"))
k = d["kernel_ms_per_frame"]
print(f"{(sys.argv[1].split('/')[-1] or 'shipped'):16s} frame {d['ms_per_step']:8.2f}  packet {k['packet_ms']:6.2f} extend {k['extend_ms']:7.2f} shade {k['shade_ms']:6.2f} shadow {k['shadow_ms']:7.2f} tail {k['tail_ms']:5.2f}  {d['verify']['sum_sha256_16']}")
PY
done
cat $OUT
