# timing experiments on the role-split forward (msst_fwd3.hip): rebuild with the given -D flags and time the forward of a B = 256 step (HIP events)
for e in "$@"; do
  python -c "from maskedsst_amd.build import build; build(force=True, extra_flags=tuple('$e'.split()))" > /dev/null 2>&1
  echo "== $e"; python bench.py --no-cpu-baseline --no-pipeline --profile-all --steps 6 $BENCH_ARGS 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], {k:round(v['avg_us'],1) for k,v in d['kernels'].items() if k.startswith('block_fwd')})"
done
