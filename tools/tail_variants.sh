#!/bin/bash
# bash tools/tail_variants.sh base tw2 tw4 ... : frame wall time per library variant (tools/build_variant.sh) -> gpurun_out/tail_variants.log
mkdir -p gpurun_out
out=gpurun_out/tail_variants.log
: > $out
for v in "$@"; do
  if [ $v = base ]; then unset PT_VARIANT_LIB; else export PT_VARIANT_LIB=rust-pathtracer_amd/lib/variants/$v.so; fi
  for cfg in "C3 uniform PT_PARTITION=0,8" "C3 uniform PT_X=0" "C5 both PT_PARTITION=0,8" "C1 uniform PT_X=0"; do
    set -- $cfg
    env $3 timeout -k 10 200 python3 tools/frame_breakdown.py $1 $2 > /tmp/var_one.log 2>&1 || { tail -5 /tmp/var_one.log; exit 1; }
    python3 - "$v $3" <<'PY' | tee -a $out
import json, sys
for l in open('/tmp/var_one.log'):
    if not l.startswith('{'): continue
    r = json.loads(l)
    if 'unprofiled_uniform_wall_s' in r: print(sys.argv[1], r['config'][:12], 'uniform wall', r['unprofiled_uniform_wall_s'])
    elif r.get('part') == 'adaptive': print(sys.argv[1], r['config'][:12], 'adaptive wall', r['wall_s'], 'tail_ms', r['timings']['tail_ms'])
    else: print(sys.argv[1], r['config'][:12], '   profiled tail_ms', r['timings']['tail_ms'])
PY
  done
done
