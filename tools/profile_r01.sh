#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + the two HBM traffic counter passes of bench.py,
# each in its own rocprofv3 run.  Output under gpurun_out/<tag>_{kt,fetch,write}; condense with
# tools/summarize_profile.py <tag>.
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt -o kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra > $R/gpurun_out/${TAG}_kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch -o pmc --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 10 > $R/gpurun_out/${TAG}_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write -o pmc --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 10 > $R/gpurun_out/${TAG}_write.log 2>&1
grep -h '"metric"' $R/gpurun_out/${TAG}_kt.log | tail -1
