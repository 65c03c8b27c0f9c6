# round-5 A/B driver (GPU box): rebuild the library with each quoted flag set and time the attention backward at B = 256
# (tools/dev_bwd3.py: r4 kernel = "flag 0", r3 one-head kernel = "flag 128" as the box reference; also prints the parity of the
# build against the template kernel).  usage: bash tools/exp_r5.sh "" "-DMSST_B4_SPLIT=0" ...
for e in "$@"; do
  python3 -c "from maskedsst_amd.build import build; build(force=True, extra_flags=tuple('$e'.split()))" > /dev/null 2>&1 || echo "BUILD FAILED: $e"
  echo "== [$e]"; python3 tools/dev_bwd3.py time small 2>&1 | grep "flag\|== drop\|nan" | tail -6
done
python3 -c "from maskedsst_amd.build import build; build(force=True)" > /dev/null 2>&1
