# the bench lines of a round (with cpu_baseline) on one box -> gpurun_out/r4bench/
O=gpurun_out/r4bench; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/e1.log
python bench.py > $O/bench_default.json 2> $O/e2.log
python bench.py --workload fir > $O/bench_fir.json 2> $O/e4.log
python bench.py --workload welch > $O/bench_welch.json 2> $O/e5.log
python bench.py --workload stft > $O/bench_stft.json 2> $O/e6.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -2 $O/smoke.txt
