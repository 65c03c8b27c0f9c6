# A/B of whole-step variants on the GPU box: each argument is "<env assignments> | <build flags>" (either side may be empty);
# rebuilds when flags are given, runs bench.py with every kernel bracketed and prints the per-kernel averages.
# usage: bash tools/exp_bench.sh " | " "MSST_LSE=0 | " " | -DMSST_F3_YSC1=1"      (flags containing -DMSST_LAB need MSST_ALLOW_LAB=1 on the env side)
for v in "$@"; do
  envs="${v%%|*}"; flags="${v#*|}"
  python3 -c "from maskedsst_amd.build import build; build(force=True, extra_flags=tuple('$flags'.split()))" > /dev/null 2>&1 || echo "BUILD FAILED: $flags"
  echo "== [$envs|$flags]"
  env $envs python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-traffic ${EXP_BENCH_ARGS:-} 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('unbracketed:', d['value'], d['ms_per_step'])"
  env $envs python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pipeline --no-traffic --profile-all ${EXP_BENCH_ARGS:-} 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: round(v['avg_us'],1) for k,v in d['kernels'].items() if k.startswith('block') or k.startswith('reduce')})"
done
python3 -c "from maskedsst_amd.build import build; build(force=True)" > /dev/null 2>&1
