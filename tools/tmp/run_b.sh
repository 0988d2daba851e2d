set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_real
rm -rf $O && mkdir -p $O
cd $R
PMX_VERBOSE=1 python3 tools/real_reads.py 2 > $O/verbose.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/real_reads.py 3 > $O/stats.log 2>&1
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/stats
PMX_BENCH_HOST_TIMES=1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-real-reads --no-host-to-host > $O/host_times.log 2>&1
