# tools/_s4_ab.sh file.hip "<flags 1>" "<flags 2>" ... : bench each build variant (dropout 0.1, B=256), kernel table
src=$1; shift
for flags in "$@"; do
  touch maskedsst_amd/csrc/$src
  python3 - <<PY
from maskedsst_amd.build import build
build(extra_flags=tuple("$flags".split()))
PY
  timeout 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-all 2>&1 | tail -1 > /tmp/b.json
  python3 - <<PY
import json
d=json.loads(open("/tmp/b.json").read())
print("[$flags]", round(d["value"]), "samples/s", {k: round(v["avg_us"],1) for k,v in d["kernels"].items() if k.startswith("block")}, flush=True)
PY
done
