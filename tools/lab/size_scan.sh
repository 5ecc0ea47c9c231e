# NODDI fit rate over the call size (inputs in HBM): bash tools/lab/size_scan.sh
for v in 50000 100000 200000 500000 1000000 2000000 4000000; do
  python bench.py --steps 5 --warmup 2 --voxels $v --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v voxels: %.2f M voxels/s  %.2f ms' % (d['value']/1e6, d['ms_per_step']))"
done
