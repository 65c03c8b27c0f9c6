# tools/_s4_ab_env.sh file.hip "<flags>" "ENV=val ..." ... : like _s4_ab.sh, variants = (flags, env) pairs
src=$1; shift
while [ $# -gt 0 ]; do
  flags="$1"; envs="$2"; shift; shift
  touch maskedsst_amd/csrc/$src
  python3 - <<PY
from maskedsst_amd.build import build
build(extra_flags=tuple("$flags".split()))
PY
  env $envs timeout 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-all 2>&1 | tail -1 > /tmp/b.json
  python3 - <<PY
import json
d=json.loads(open("/tmp/b.json").read())
print("[$flags | $envs]", round(d["value"]), "samples/s", {k: round(v["avg_us"],1) for k,v in d["kernels"].items() if k.startswith("block")}, flush=True)
PY
done
