# A/B on the GPU box: packed-fp32 VALU (v_pk_mul/fma/add_f32) disabled per source file, cumulatively
# (tools/valu_microbench.hip: one v_pk_*_f32 beside another wave's MFMAs costs ~120 cycles, two scalar ops ~20)
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
run() {
  python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pipeline --profile-all ${AB_BENCH_ARGS} 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], {k: round(v['avg_us'],1) for k,v in d['kernels'].items()})"
}
echo "[base]"; run; run
for f in msst_bwd4.hip msst_fwd3.hip msst_bwd5.hip "msst_bwd.hip msst_fwd.hip msst_opt.hip"; do
  for g in $f; do
    python3 - <<PY
import os, subprocess
from maskedsst_amd import build as b
src = os.path.join(b.CSRC, "$g"); obj = os.path.join(b.HERE, "build", "$g.o")
subprocess.run([b._hipcc()] + b.FLAGS + "$NOPK".split() + ["-c", src, "-o", obj], check=True, capture_output=True)
PY
  done
  python3 -c "from maskedsst_amd.build import build; build()"
  echo "[+ nopk $f]"; run; run
done
