for v in 200000 1000000 4000000; do for e in "A=1" "AMX_NO_CHUNK_ORDER=1" "A=1" "AMX_NO_CHUNK_ORDER=1"; do
  echo "== $v $e"
  env $e python bench.py --steps 5 --warmup 2 --voxels $v --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.2f M voxels/s  %.2f ms  stages %s seeds %s' % (d['value']/1e6, d['ms_per_step'], [round(v,2) for v in r['stage_ms']], [round(v,2) for v in r['seed_ms']]))"
done; done
python tools/seed_check.py 1000000 2>&1 | tail -4
