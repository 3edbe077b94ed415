cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06z
python bench.py --workload c1 --steps 200 --warmup 20 --cpu-seconds 5 2>/dev/null | tail -1 > gpurun_out/r06z/bench_c1.json
python3 -c "import json; d=json.loads(open('gpurun_out/r06z/bench_c1.json').read()); print('c1', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['steps_per_event_pair'])"
python bench.py --workload c5 --steps 1 --warmup 0 --cpu-seconds 10 2> gpurun_out/r06z/bench_c5_full_definition.err | tail -1 > gpurun_out/r06z/bench_c5_full_definition.json
python3 -c "import json; d=json.loads(open('gpurun_out/r06z/bench_c5_full_definition.json').read()); print('c5 full', d['value'], d['ms_per_step'], {k: round(v['avg_ms'],2) for k,v in d['roofline']['kernels'].items()}, d['roofline']['frac'])"
