# A/B of two builds of the library on one box: benchmarks/bin/libA.so, libB.so (alternating)
for i in 1 2 3; do
  for v in A B; do
    cp benchmarks/bin/lib$v.so openseize_amd/lib/libosz_hip.so
    python bench.py --steps 40 --warmup 15 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'])"
  done
done
cp benchmarks/bin/libB.so openseize_amd/lib/libosz_hip.so
timeout -k 10 300 python -m pytest tests/test_gpu_zp.py tests/test_gpu_nonfinite.py -q -x 2>&1 | tail -2
