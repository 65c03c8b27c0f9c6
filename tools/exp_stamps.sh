# stamps of the attention backward for each quoted flag set (GPU box): bash tools/exp_stamps.sh "" "-DMSST_B4_SPLIT=0"
for e in "$@"; do
  python3 -c "from maskedsst_amd.build import build; build(force=True, extra_flags=tuple('-DMSST_STAMPS $e'.split()))" > /dev/null 2>&1 || echo "BUILD FAILED: $e"
  echo "== [$e]"; timeout 240 python3 tools/stamps_bwd4.py 2>&1 | tail -16
done
python3 -c "from maskedsst_amd.build import build; build(force=True)" > /dev/null 2>&1
