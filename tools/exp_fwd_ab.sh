# A/B of forward build variants by the stack / per-block ratio of tools/fwd_ab.py: each argument is a set of build flags
# usage: bash tools/exp_fwd_ab.sh "" "-DMSST_F3_TOUCH=0" ...     (FWD_AB_ARGS="--batch 64" for other shapes)
for flags in "$@"; do
  python3 -c "from maskedsst_amd.build import build; build(force=True, extra_flags=tuple('$flags'.split()))" > /dev/null 2>&1 || echo "BUILD FAILED: $flags"
  echo "== [$flags]"
  python3 tools/fwd_ab.py ${FWD_AB_ARGS:-} 2>&1 | tail -1
done
python3 -c "from maskedsst_amd.build import build; build(force=True)" > /dev/null 2>&1
