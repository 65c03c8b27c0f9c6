# A/B of build-flag variants of one source file on the GPU box: tools/ab_flags.sh file.hip "<flags>" "<flags>" ...
# prints the per-kernel event timings of the default bench for each variant (3 steps)
src=$1; shift
for flags in "$@"; do
  touch maskedsst_amd/csrc/$src
  python3 - <<PY
from maskedsst_amd.build import build
build(extra_flags=tuple("$flags".split()))
PY
  echo "[$flags]"
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline --profile-all ${AB_BENCH_ARGS} 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], {k: round(v['avg_us']) for k,v in d['kernels'].items() if k.startswith('block')})"
done
touch maskedsst_amd/csrc/$src
