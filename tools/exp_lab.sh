# kernel-study (timing-only, WRONG results) builds on the GPU box: bash tools/exp_lab.sh "-DMSST_LAB_EXP=1" ...
# each variant is built with -DMSST_LAB (msst_version() < 0: only loadable with MSST_ALLOW_LAB=1) and timed at B = 256
export MSST_ALLOW_LAB=1
for e in "$@"; do
  python3 -c "from maskedsst_amd.build import build; build(force=True, extra_flags=tuple('-DMSST_LAB $e'.split()))" > /dev/null 2>&1 || echo "BUILD FAILED: $e"
  echo "== LAB [$e]"; python3 tools/dev_bwd3.py time small 2>&1 | grep "^flag" | tail -2
done
unset MSST_ALLOW_LAB
python3 -c "from maskedsst_amd.build import build; build(force=True)" > /dev/null 2>&1
