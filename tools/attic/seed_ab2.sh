# env-variant A/B of the NODDI headline: bash tools/seed_ab2.sh "NAME=VALUE ..." ...   (each argument = one environment)
for e in "$@"; do
  echo "== $e"
  env $e python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.2f M voxels/s  %.2f ms  stages %s seeds %s' % (d['value']/1e6, d['ms_per_step'], [round(v,2) for v in r['stage_ms']], [round(v,2) for v in r['seed_ms']]))"
done
