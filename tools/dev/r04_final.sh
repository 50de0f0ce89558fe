# final checks at HEAD: smoke(), the driver's default bench line, kernel statistics
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-r04z}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > gpurun_out/${TAG}_smoke.log
python bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_default.json
bash tools/kernel_stats.sh $TAG fp32 > /dev/null 2>&1
bash tools/kernel_stats.sh $TAG f16 > /dev/null 2>&1
cat gpurun_out/${TAG}_smoke.log; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_default.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['hbm_roofline']['frac'], d.get('clocks_under_load'), (d.get('whole_step_traffic') or {}).get('ratio'))
print(d['secondary']['ms_per_step'], {k:(v.get('ms_per_step'), v.get('error')) for k,v in d['modes'].items()})"
