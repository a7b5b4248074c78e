#!/bin/bash
# After `gpurun -- bash tools/collect_profiles.sh <tag>`: copy the judged summaries from gpurun_out/profiles_<tag>/ into profiles/.
TAG=${1:-r04}
SRC=gpurun_out/profiles_$TAG
cp $SRC/hbm_traffic.json $SRC/${TAG}_pmc_*_step.csv $SRC/${TAG}_bench_default.json $SRC/${TAG}_bench_steps20.json $SRC/${TAG}_bench_mt19937.json \
   $SRC/${TAG}_bench_kernel_stats.csv $SRC/${TAG}_bench_steps20_kernel_stats.csv profiles/
