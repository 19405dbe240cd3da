#!/bin/bash
# Collect the rocprofv3 evidence for the bench command on the GPU box
# (run through gpurun from the repo root):
#   bash benchmarks/refresh_profiles.sh
# then, back in the build container:
#   python benchmarks/summarise_profiles.py gpurun_out/prof rNN
# The counter passes are separate runs with nothing but --pmc (one counter
# each), as MI355X_MICROARCH.md prescribes; the timing pass has no counters.
set -e
export TMPDIR=/tmp
P=gpurun_out/prof
rm -rf $P && mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu > $P/bench_stats.json 2> $P/stats.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -- python3 bench.py --steps 5 --warmup 3 --no-cpu > $P/bench_fetch.json 2> $P/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -- python3 bench.py --steps 5 --warmup 3 --no-cpu > $P/bench_write.json 2> $P/write.log
# keep only the summaries (the traces are large)
find $P -name '*kernel_trace.csv' -delete
ls -R $P | head -40
