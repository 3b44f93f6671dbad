#!/bin/bash
# copy the outputs of scripts_gpu_profile.sh / scripts_gpu_pmc.sh / scripts_gpu_static.sh (gpurun_out/<tag>*) into
# profiles/ under the names profiles/README.md lists.   usage: tools/collect_profiles.sh r02_v5
set -e
tag=$1
g=gpurun_out
cp $g/$tag/bench.json profiles/${tag}_bench.json
cp $g/$tag/bench_under_rocprof.json profiles/${tag}_bench_under_rocprof.json
cp $g/$tag/kernel_stats.csv profiles/${tag}_kernel_stats.csv
cp $g/$tag/bench_c3.json profiles/${tag}_bench_c3.json
cp $g/${tag}_static/kernel_stats.csv profiles/${tag}_kernel_stats_static.csv
mkdir -p profiles/${tag}_pmc
cp $g/${tag}_pmc/p*_per_kernel_avg.csv profiles/${tag}_pmc/
cp $g/${tag}_pmc/traffic.json profiles/r02_traffic.json
[ -f $g/tile_clocks.txt ] && cp $g/tile_clocks.txt profiles/${tag}_tile_clocks.txt
ls profiles | grep $tag
