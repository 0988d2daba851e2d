# the numbers of profiles/r02 (v3): default bench line, config 3 on one GPU, config 4, --refine timing; outputs under gpurun_out/final
O=gpurun_out/final
rm -rf $O; mkdir -p $O
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --scaling strong --total-reads 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-real-reads --no-host-to-host > $O/bench_config3.json 2> $O/bench_config3.err
python bench.py --read-len 10000 --reads-per-gpu 100000 --steps 2 --warmup 1 > $O/bench_config4.json 2> $O/bench_config4.err
python tools/refine_timing.py > $O/refine_timing.log 2>&1
tail -2 $O/*.err
