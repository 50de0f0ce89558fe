for f in pepflowww_amd/lib/variants/libpf_prof.so pepflowww_amd/lib/variants/libpf_wi_*.so; do
  echo "== $(basename $f)"
  PF_LIB_PATH=$f timeout 100 python tools/dev/et5_prof.py 2>&1 | tail -8 | awk '{print $0}' | cut -c1-60,100-140
done
