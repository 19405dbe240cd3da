# every bench line and sweep of a round in one run on the GPU box -> gpurun_out/r5lines/
set -x
O=gpurun_out/r5lines; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/e1.log
echo driver-flags line done
python bench.py --no-cpu > $O/bench_default.json 2> $O/e2.log
python bench.py --full-stream --no-cpu > $O/bench_full_stream.json 2> $O/e3.log
python bench.py --workload fir > $O/bench_fir.json 2> $O/e4.log
python bench.py --workload welch > $O/bench_welch.json 2> $O/e5.log
python bench.py --workload stft > $O/bench_stft.json 2> $O/e6.log
python bench.py --two-kernel --no-cpu > $O/bench_two_kernel.json 2> $O/e7.log
python bench.py --steps 24 --warmup 8 --no-cpu --scaling strong --shard-of 8 > $O/bench_shard_of_8.json 2> $O/e8.log
python bench.py --steps 32 --warmup 16 --no-cpu --scaling strong --shard-of 16 > $O/bench_shard_of_16.json 2> $O/e8b.log
python bench.py --steps 20 --warmup 5 --no-cpu --zp-tol 1e-12 > $O/bench_tol_1e-12.json 2> $O/e8c.log
python benchmarks/sweep_chain.py > $O/channel_sweep.jsonl 2> $O/e9.log
python benchmarks/zp_coverage.py > $O/zp_coverage.jsonl 2> $O/e10.log
python benchmarks/tutorial_shape.py > $O/tutorial_shape.jsonl 2> $O/e11.log
python benchmarks/bench_kernels.py > $O/secondary_kernels.jsonl 2> $O/e12.log
python benchmarks/dc_probe.py > $O/dc_probe.txt 2> $O/e13.log
python benchmarks/large_nfft.py > $O/large_nfft.jsonl 2> $O/e14.log
python tests/fuzz_gpu.py 4200 505 > $O/fuzz.txt 2>&1
tail -3 $O/fuzz.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -2 $O/smoke.txt
