mkdir -p gpurun_out/r06z
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06z/gputest_full.log 2>&1
echo "gpu suite rc=$?"; tail -4 gpurun_out/r06z/gputest_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06z/smoke.log 2>&1; tail -1 gpurun_out/r06z/smoke.log
/usr/bin/time -v python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06z/bench_driver_cmd.json 2> gpurun_out/r06z/bench_driver_cmd.err
grep -E "Elapsed|Maximum resident" gpurun_out/r06z/bench_driver_cmd.err
python -c "
import json
d=json.loads(open('gpurun_out/r06z/bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['opt_in_f16x2'].get('ms_per_step'), {k:(v.get('ms_per_step'), v.get('error')) for k,v in d['other_configs'].items() if isinstance(v,dict)}, d['other_configs']['total_wall_s'])
"
