#!/bin/bash
# bash tools/tail_sweep.sh : frame wall time against the tail kernel's knobs -> gpurun_out/tail_sweep.log
mkdir -p gpurun_out
out=gpurun_out/tail_sweep.log
: > $out
one() {  # label cfg part env...
  local label=$1 cfg=$2 part=$3; shift 3
  env "$@" timeout -k 10 200 python3 tools/frame_breakdown.py $cfg $part > /tmp/sweep_one.log 2>&1 || { tail -5 /tmp/sweep_one.log; exit 1; }
  python3 - "$label" <<'PY' | tee -a $out
import json, sys
rows = [json.loads(l) for l in open('/tmp/sweep_one.log') if l.startswith('{')]
for r in rows:
    if 'unprofiled_uniform_wall_s' in r: print(sys.argv[1], r['config'][:12], 'uniform wall', r['unprofiled_uniform_wall_s'])
    elif r.get('part') == 'adaptive': print(sys.argv[1], r['config'][:12], 'adaptive wall', r['wall_s'], 'tail_ms', r['timings']['tail_ms'], r['timings']['tail_launches'])
    else: print(sys.argv[1], r['config'][:12], '   profiled', r['wall_s'], 'tail_ms', r['timings']['tail_ms'], r['timings']['tail_launches'], 'extend', r['timings']['extend_launches'])
PY
}
for cap in ${CAPS:-400000 600000 800000 1200000 1560000}; do
  e="PTX_TAIL_CAP=$cap"
  one "cap=$cap 1/8" C3 uniform $e PT_PARTITION=0,8
  one "cap=$cap full" C3 uniform $e
  one "cap=$cap C5 1/8" C5 both $e PT_PARTITION=0,8
done
