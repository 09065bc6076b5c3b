#!/bin/bash
# bash tools/run_variants.sh base v1 v2 ... : frame (one stream) per kernel class for libraries built by tools/build_variant.sh
# CFGS="C3 C2" selects the configurations -> gpurun_out/variants.log
mkdir -p gpurun_out; : > gpurun_out/variants.log
for v in "$@"; do
  if [ $v = base ]; then unset PT_VARIANT_LIB; else export PT_VARIANT_LIB=rust-pathtracer_amd/lib/variants/$v.so; fi
  for c in ${CFGS:-C3}; do
  PTX_OVERLAP_SHADOW=0 timeout -k 10 200 python3 tools/frame_breakdown.py $c uniform > /tmp/rv.log 2>&1 || { tail -5 /tmp/rv.log; exit 1; }
  PTX_OVERLAP_SHADOW=1 timeout -k 10 200 python3 tools/frame_breakdown.py $c uniform > /tmp/rv2.log 2>&1 || { tail -5 /tmp/rv2.log; exit 1; }
  python3 - $v $c <<'PY' | tee -a gpurun_out/variants.log
import json, sys
for f, lab in (('/tmp/rv.log', 'one stream'), ('/tmp/rv2.log', 'two streams')):
    for l in open(f):
        if '"part"' in l:
            d = json.loads(l); t = d['timings']
            print(sys.argv[1], d['config'][:14], lab, 'wall', d['wall_s'], {k: t[k] for k in ('packet_ms', 'extend_ms', 'shade_ms', 'shadow_ms', 'tail_ms')})
PY
  done
done
