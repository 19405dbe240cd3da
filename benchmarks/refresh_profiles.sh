#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun
# from the repo root):
#   bash benchmarks/refresh_profiles.sh
# then, back in the build container:
#   python benchmarks/summarise_profiles.py gpurun_out/prof rNN
#   python benchmarks/summarise_pmc.py gpurun_out/pmc rNN
# The timing pass has no counters; every counter pass is a separate run with
# nothing but --pmc (MI355X_MICROARCH.md), the program directly after `--`.
export TMPDIR=/tmp
P=gpurun_out/prof
rm -rf $P gpurun_out/pmc && mkdir -p $P
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-steady > $P/bench_stats.json 2> $P/stats.log || exit 1
timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -- python3 bench.py --steps 5 --warmup 3 --no-cpu --no-steady > $P/bench_fetch.json 2> $P/fetch.log || exit 1
timeout -k 10 420 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -- python3 bench.py --steps 5 --warmup 3 --no-cpu --no-steady > $P/bench_write.json 2> $P/write.log || exit 1
# the round-2 step (fused forward kernel beside the backward pass), for comparison
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_2k -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-steady --two-kernel > $P/bench_stats_2k.json 2> $P/stats_2k.log || exit 1
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_sec -- python3 benchmarks/bench_kernels.py --pmc-subset > $P/sec_stats.jsonl 2> $P/stats_sec.log || exit 1
# keep only the summaries (the traces are large)
find $P -name '*kernel_trace.csv' -delete
bash benchmarks/collect_pmc.sh all
ls -R $P | head -40
