#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- bash tools/collect_issue_counters.sh r03): SQ issue counters (two passes of eight) and a kernel trace of one
# 50-step episode of each workload of tools/traffic_run.py; tools/issue_counters.py reduces them to <tag>_issue_counters.txt
set -o pipefail
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/counters_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WORKLOADS="wildfire cybersecurity rideshare wildfire_grid_8x8 wildfire_grid_16x16"
for w in $WORKLOADS; do
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_${w}_a -o pmc -- python3 $GRAFT_REPO_ROOT/tools/traffic_run.py $w > $OUT/pmc_${w}_a.log 2>&1 || echo "pass a $w failed"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_${w}_b -o pmc -- python3 $GRAFT_REPO_ROOT/tools/traffic_run.py $w > $OUT/pmc_${w}_b.log 2>&1 || echo "pass b $w failed"
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_${w} -o trace -- python3 $GRAFT_REPO_ROOT/tools/traffic_run.py $w > $OUT/trace_${w}.log 2>&1 || echo "trace $w failed"
done
python3 $GRAFT_REPO_ROOT/tools/issue_counters.py $OUT $TAG $WORKLOADS > $OUT/${TAG}_issue_counters.txt 2> $OUT/issue_counters.err || echo "summary failed"
tail -40 $OUT/${TAG}_issue_counters.txt
