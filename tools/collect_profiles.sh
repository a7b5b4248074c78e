#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- bash tools/collect_profiles.sh r02): the round's measurement artefacts under gpurun_out/profiles_<tag>/,
# all from ONE box in one call (the boxes of the pool differ by ~10 %), to be copied into profiles/ afterwards (tools/adopt_profiles.sh).
# Order: the counter passes first (rocprofv3 gets the program itself after `--`, one counter per pass, no trace domains) and their
# summary — profiles/hbm_traffic.json carries the fingerprint of the kernel sources — so that the bench lines that follow report
# `roofline.traffic` from this very build; then the bench lines; then the kernel traces of the same commands.
set -o pipefail
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for domain in wildfire wildfire20 cybersecurity rideshare wildfire_grid_8x8 wildfire_grid_16x16; do
  for counter in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $counter --output-format csv -d $OUT/pmc_${domain}_${counter} -o pmc -- python3 $GRAFT_REPO_ROOT/tools/traffic_run.py $domain > $OUT/pmc_${domain}_${counter}.log 2>&1 || echo "pmc $domain $counter failed"
  done
done
# A/B of the cybersecurity task-row stores (round 4): the same episode with the library that stores them lane by lane
if [ -f $GRAFT_REPO_ROOT/free-range-zoo_amd/csrc/libfrz_hip_cy_direct.so ]; then
  for counter in FETCH_SIZE WRITE_SIZE; do
    FRZ_HIP_LIB=$GRAFT_REPO_ROOT/free-range-zoo_amd/csrc/libfrz_hip_cy_direct.so rocprofv3 --pmc $counter --output-format csv -d $OUT/pmc_cybersecurity_direct_${counter} -o pmc -- python3 $GRAFT_REPO_ROOT/tools/traffic_run.py cybersecurity > $OUT/pmc_cybersecurity_direct_${counter}.log 2>&1 || echo "pmc cybersecurity_direct $counter failed"
  done
fi
python3 $GRAFT_REPO_ROOT/tools/traffic_summarise.py $OUT $TAG > $OUT/traffic_summary.json 2> $OUT/traffic_summary.err || echo "traffic summary failed"
cp $GRAFT_REPO_ROOT/profiles/hbm_traffic.json $GRAFT_REPO_ROOT/profiles/${TAG}_pmc_*_step.csv $OUT/ 2>/dev/null
python3 $GRAFT_REPO_ROOT/bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err || echo "bench default failed"
python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_steps20.json 2> $OUT/bench20.err || echo "bench 20 failed"
python3 $GRAFT_REPO_ROOT/bench.py --rng mt19937 --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_mt19937.json 2> $OUT/bench_mt.err || echo "bench mt19937 failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/trace_bench.log 2>&1 || echo "trace failed"
cp $OUT/trace_bench/bench_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bench20 -o bench20 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-episode-probe > $OUT/trace_bench20.log 2>&1 || echo "trace 20 failed"
cp $OUT/trace_bench20/bench20_kernel_stats.csv $OUT/${TAG}_bench_steps20_kernel_stats.csv 2>/dev/null
ls $OUT
