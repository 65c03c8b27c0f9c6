python3 -m maskedsst_amd.build --stamps > /dev/null 2>&1
echo "== bwd_attn, 1 WG/CU =="; MSST_ATTN_CHUNKS=32 timeout 300 python3 tools/stamps_bwd.py 2>&1 | tail -17
echo "== bwd_attn, 2 WG/CU =="; timeout 300 python3 tools/stamps_bwd.py 2>&1 | tail -17
python3 -m maskedsst_amd.build --force > /dev/null 2>&1
