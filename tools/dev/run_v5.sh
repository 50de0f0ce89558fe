timeout 200 python tools/dev/et5_check.py 64 128 nolast 2>&1 | tail -9
timeout 200 python tools/dev/et5_check.py 64 144 ragged nolast 2>&1 | tail -9
timeout 100 python tools/dev/et5_check.py 3 48 ragged nolast 2>&1 | tail -9 | head -5
PF_LIB_PATH=pepflowww_amd/lib/variants/libpf_prof.so timeout 120 python tools/dev/et5_prof.py 2>&1 | tail -8
