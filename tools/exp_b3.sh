# timing experiments on the round-3 attention backward: rebuild with -DMSST_B3_EXP=<n> and time B = 256 (results are wrong for n != 0)
for e in "$@"; do
  python -c "from maskedsst_amd.build import build; build(force=True, extra_flags=('$e',))" > /dev/null 2>&1
  echo "== $e"; python tools/dev_bwd3.py time small 2>&1 | grep "flag 0" | tail -1
done
