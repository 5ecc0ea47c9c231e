# A/B of k_sandi_rows builds (variants/<name>/libamico_amd.so): bash tools/lab/ab_sandi.sh name ...   ("default" = the in-tree library)
for v in "$@"; do for r in 1 2; do
  echo "== $v"
  if [ $v = default ]; then L=; else L="AMICO_AMD_LIB=variants/$v/libamico_amd.so"; fi
  env $L python bench.py --model sandi --voxels 1000000 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.3f G voxels/s  %.3f ms  kernel %.3f ms  parity %s' % (d['value']/1e9, d['ms_per_step'], r['kernel_ms'], d['parity']))"
done; done
