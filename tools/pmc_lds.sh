# LDS counters of one kernel for build-flag variants: tools/pmc_lds.sh file.hip kernel_substr "<flags>" ...
src=$1; kn=$2; shift; shift
set -eu; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?not on a gpurun box}"
for flags in "$@"; do
  touch maskedsst_amd/csrc/$src
  python3 - <<PY
from maskedsst_amd.build import build
build(extra_flags=tuple("$flags".split()))
PY
  rm -rf gpurun_out/pmcl; mkdir -p gpurun_out/pmcl
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace -f csv -d gpurun_out/pmcl -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --dropout 0 > /dev/null 2>&1
  echo "[$flags]"; python3 tools/pmc_summary.py gpurun_out/pmcl | grep -A6 "$kn" | head -7
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-all --dropout 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k: round(v['avg_us']) for k,v in d['kernels'].items() if k.startswith('block')})"
done
rm -rf gpurun_out/pmcl
