for rep in 1 2; do
  for t in abtree .; do
    (cd $t && echo "== $t" && python bench.py --workload c2 --steps 10 --warmup 2 --cpu-seconds 0 --no-other-mode 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2', round(d['ms_per_step'],3), {k:round(v['avg_ms'],3) for k,v in d['roofline']['kernels'].items()})")
  done
done
