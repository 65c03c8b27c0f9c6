# A/B of library builds on the GPU box without rebuilding there: every argument is a library path; the default bench with every kernel
# bracketed, the libraries taken in turn, twice (box drift shows as the difference between the two rounds)
# usage: bash tools/ab_libs.sh maskedsst_amd/libmsst.so maskedsst_amd/libmsst_alt.so       (AB_ARGS="--batch 64" for other shapes)
for rnd in 1 2; do
  for lib in "$@"; do
    echo "== [$lib] round $rnd"
    python3 tools/with_lib.py $lib bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pipeline --no-traffic --no-alt --profile-all ${AB_ARGS:-} 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: round(v['avg_us'],1) for k,v in d['kernels'].items() if k.startswith('block') or k.startswith('reduce')})"
  done
done
