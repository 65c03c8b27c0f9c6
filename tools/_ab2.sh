#!/bin/bash
# like _ab.sh but times only the forward kernel (quick): tools/_ab2.sh file.hip "<flags>" ...
src=$1; shift
for flags in "$@"; do
  touch maskedsst_amd/csrc/$src
  python - <<PY
from maskedsst_amd.build import build
build(extra_flags=tuple("$flags".split()))
PY
  timeout 900 python bench.py --steps 3 --warmup 1 --batch 256 --no-cpu-baseline --profile-all --dropout 0.0 2>&1 | tail -1 > /tmp/b.json
  python - <<PY
import json
d=json.loads(open("/tmp/b.json").read())
print("[$flags]", d["value"], "samples/s", {k: round(v["avg_us"]) for k,v in d["kernels"].items() if k.startswith("block")})
PY
done
