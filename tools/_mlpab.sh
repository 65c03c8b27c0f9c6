for v in "1 256" "2 512" "2 256"; do
set -- $v
touch maskedsst_amd/csrc/msst_bwd.hip
python - <<PY
from maskedsst_amd.build import build
build(extra_flags=("-DMSST_MLP_WAVES=$1",))
PY
MSST_BWD_GRID=$2 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-all 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('waves $1 grid $2', d['value'], {k: round(v['avg_us']) for k,v in d['kernels'].items() if k.startswith('block') or k=='reduce_slabs'})"
done
