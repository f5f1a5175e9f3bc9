python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
T0=$(date +%s); python bench.py > gpurun_out/final_bench.log 2> gpurun_out/final_bench.err; echo "Elapsed $(( $(date +%s) - T0 )) s"

tail -1 gpurun_out/final_bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['value_tolerance_compliant']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['variants'].items()}, d['north_star_b256']['ms_per_step'], d['north_star_b256']['trunk_fwd_bwd']['frac'], d['cpu_baseline']['value'])
"
