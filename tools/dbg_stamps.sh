python3 -c "from maskedsst_amd.build import build; build(force=True, extra_flags=('-DMSST_STAMPS',))" > /dev/null 2>&1
HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 python3 -X faulthandler tools/stamps_bwd4.py 2>&1 | tail -30
python3 -c "from maskedsst_amd.build import build; build(force=True)" > /dev/null 2>&1
