# timing experiments on the attention backward: rebuild with the given -D flags (one quoted string per variant) and time B = 256
for e in "$@"; do
  python -c "from maskedsst_amd.build import build; build(force=True, extra_flags=tuple('$e'.split()))" > /dev/null 2>&1
  echo "== $e"; python tools/dev_bwd3.py time small 2>&1 | grep "flag" | tail -2
done
