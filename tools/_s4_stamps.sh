# stamps of one wave of block_fwd_hw and block_bwd_attn (stamps build), then restore the production build
python3 -m maskedsst_amd.build --stamps > /dev/null 2>&1
echo "== fwd_hw =="; timeout 300 python3 tools/stamps_fwd2.py 2>&1 | tail -20
echo "== bwd_attn =="; timeout 300 python3 tools/stamps_bwd.py 2>&1 | tail -20
python3 -m maskedsst_amd.build --force > /dev/null 2>&1
