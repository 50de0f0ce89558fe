for r in 1 2; do for lib in new sleep sleep3; do
  if [ $lib = new ]; then unset PF_LIB_PATH; else export PF_LIB_PATH=$PWD/pepflowww_amd/lib/variants/libpf_$lib.so; fi
  python bench.py --workload cfg4 --no-modes --no-per-call --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib', 'ms_per_step %.4f' % d['ms_per_step'], 'ET launch %.1f us' % d['roofline']['avg_launch_us'], 'sclk', d['roofline'].get('sclk_mhz'))"
done; done
