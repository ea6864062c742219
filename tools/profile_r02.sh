#!/bin/bash
# Runs on the GPU box (through gpurun): for every configuration, a kernel trace and the two HBM traffic counter
# passes of tools/profile_op.py, each in its own rocprofv3 run; then the same three passes over bench.py.
# Output under gpurun_out/<tag>_<cfg>_{kt,fetch,write}; condense with tools/summarize_r02.py <tag>.
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for cfg in cfg1 cfg2 cfg3 cfg4 cfg5sin cfg5mix; do
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_${cfg}_kt -o kt --output-format csv -- python3 $R/tools/profile_op.py $cfg 10 > $R/gpurun_out/${TAG}_${cfg}_kt.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_${cfg}_fetch -o pmc --output-format csv -- python3 $R/tools/profile_op.py $cfg 4 > $R/gpurun_out/${TAG}_${cfg}_fetch.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_${cfg}_write -o pmc --output-format csv -- python3 $R/tools/profile_op.py $cfg 4 > $R/gpurun_out/${TAG}_${cfg}_write.log 2>&1
  echo "done $cfg"
done
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_bench_kt -o kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra > $R/gpurun_out/${TAG}_bench_kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_bench_fetch -o pmc --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 10 --prewarm 10 --warmup 5 > $R/gpurun_out/${TAG}_bench_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_bench_write -o pmc --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 10 --prewarm 10 --warmup 5 > $R/gpurun_out/${TAG}_bench_write.log 2>&1
grep -h '"metric"' $R/gpurun_out/${TAG}_bench_kt.log | tail -1 | cut -c1-400
