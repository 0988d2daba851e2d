#!/bin/bash
# The measurements kept under profiles/r03 (run on a GPU box from the repo root).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/measure_r03
mkdir -p $O
cd $R
python bench.py > $O/bench_default_10M.json 2> $O/bench_default_10M.err
python bench.py --total-reads 1000000 --no-cpu-baseline > $O/bench_1M.json 2> $O/bench_1M.err
python bench.py --scaling weak --reads-per-gpu 100000 --read-len 10000 --steps 3 --warmup 1 > $O/bench_config4_long_reads.json 2> $O/bench_config4.err
tail -c 300 $O/*.err
