for c in 512 1024 2048 4096; do
  echo "chunk $c: $(AMX_REFILL_CHUNK=$c timeout 200 python bench.py --model freewater --voxels 2000000 --steps 5 --warmup 2 | grep -o '"kernel_ms": [0-9.]*')"
done
