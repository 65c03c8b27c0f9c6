for t in 0 256; do
python3 - <<PY
from maskedsst_amd.build import build
import os
os.utime("maskedsst_amd/csrc/msst_fwd2.hip")
build(extra_flags=("-DMSST_STAMPS","-DMSST_F2_STAMP_TID=$t"))
PY
echo "== fwd_hw tid $t =="; timeout 300 python3 tools/stamps_fwd2.py 2>&1 | grep -v amdgpu.ids | tail -15
done
python3 -m maskedsst_amd.build --force > /dev/null 2>&1
