#!/bin/bash
# copy the outputs of tools/gpu/r3_campaign.sh (gpurun_out/<tag>/) into profiles/ under the names profiles/README.md lists,
# replacing the files of the tag given as second argument.   usage: tools/collect_campaign.sh r03_v5 [r03_v4]
set -e
tag=$1; old=$2
g=gpurun_out/$tag
r=${tag%%_*}
if [ -n "$old" ]; then git rm -q -r --ignore-unmatch profiles/${old}_bench*.json profiles/${old}_kernel_stats*.csv profiles/${old}_pmc profiles/${old}_soak_*.log; fi
for n in "" _poly _c3 _c5; do [ -f $g/bench$n.json ] && cp $g/bench$n.json profiles/${tag}_bench$n.json; done
cp $g/kernel_stats_bench.csv profiles/${tag}_kernel_stats.csv
cp $g/kernel_stats_bench_static.csv profiles/${tag}_kernel_stats_static.csv
for n in poly c3 c5; do cp $g/kernel_stats_bench_$n.csv profiles/${tag}_kernel_stats_$n.csv; done
for n in c4 poly c3 c5; do mkdir -p profiles/${tag}_pmc/$n; cp $g/pmc_$n/p*_per_kernel_avg.csv profiles/${tag}_pmc/$n/; done
cp $g/pmc_c4/traffic.json profiles/${r}_traffic.json
cp $g/pmc_poly/traffic.json profiles/${r}_traffic_poly.json
cp $g/pmc_poly/traffic_points.json profiles/${r}_traffic_poly_points.json
cp $g/pmc_poly/traffic_tile.json profiles/${r}_traffic_poly_tile.json
cp $g/pmc_c3/traffic.json profiles/${r}_traffic_c3.json
cp $g/pmc_c3/traffic_steer.json profiles/${r}_traffic_c3_steer.json
cp $g/pmc_c5/traffic.json profiles/${r}_traffic_c5.json
for n in 2rank_weak 2rank_strong 2rank_obstacles 4rank_grid2x2; do cp $g/bench_$n.json profiles/${r}_bench_${n}_rehearsal.json; done
for s in lattice:400 dubins:200 polygons:300 cull:100; do n=${s%%:*}; k=${s#*:}; [ -f $g/soak_$n.log ] && cp $g/soak_$n.log profiles/${tag}_soak_${n}_$k.log; done
ls profiles | grep $tag
