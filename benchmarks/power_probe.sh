#!/bin/bash
# Shader clock and socket power while the headline step runs (is the kernel held by the power cap?)
#   bash benchmarks/power_probe.sh   (on the GPU box)
rocm-smi --showclocks --showpower --showmaxpower 2>&1 | grep -i "sclk\|power\|mclk" | head -8
echo "--- under load"
PYTHONPATH=. python bench.py --steps 4000 --warmup 5 --no-cpu > /tmp/pp_bench.json 2>/tmp/pp_bench.err &
BP=$!
sleep 25
for k in 1 2 3 4; do
  rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|Power" | head -4
  sleep 1
done
wait $BP
cut -c1-200 /tmp/pp_bench.json
