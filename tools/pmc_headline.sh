#!/bin/bash
# Runs on the GPU box: SQ / LDS / L2 counter passes over the headline filter pass (tools/path_probe.py cfg5sin, a
# few steps), each counter set in its own rocprofv3 run.  usage: tools/pmc_headline.sh <tag> [cfg]; output under
# gpurun_out/<tag>_*; tools/pmc_summary.py <tag> prints the per-kernel averages.
set -e
TAG=$1
CFG=${2:-cfg5sin}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export PROBE_WARM=3 PROBE_STEPS=5
run() { # name, counters
  timeout -s KILL 120 rocprofv3 --pmc $2 -d $R/gpurun_out/${TAG}_$1 -o pmc --output-format csv -- python3 $R/tools/path_probe.py $CFG slots ${OP:-filter} > $R/gpurun_out/${TAG}_$1.log 2>&1
}
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"
run sq2 "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT"
