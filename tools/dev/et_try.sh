python -m pytest tests/test_gpu_parity.py -q -x -k "edge or transition" 2>&1 | tail -1
for p in fp32 f16; do python bench.py --no-cpu-baseline --no-secondary --precision $p 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline_other']['avg_launch_us'])"; done
for p in f16 fp32; do PB=64 PL=128 PPREC=$p bash tools/phase_profile.sh 2>&1 | tail -1; done
