# usage: bash tools/exp/variants.sh name1 name2 ...   (libraries under rust-pathtracer_amd/lib/variants)
for v in "$@"; do
  echo "== $v"
  PT_VARIANT_LIB=rust-pathtracer_amd/lib/variants/$v.so timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_frame'], d['rays_per_frame'])" || exit 1
done
