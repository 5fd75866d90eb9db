#!/bin/bash
# Copies the summaries of gpurun_out/refresh (tools/refresh_profiles.sh) into profiles/ under the given tag.
# usage: tools/store_profiles.sh <bench-json-name> <kernel-stats-name> [round tag, default r02]
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/refresh
T=${3:-r03}
cp $O/bench.json profiles/$1
ls -t $O/stats/runc/*_kernel_stats.csv | head -1 | xargs -I{} cp {} profiles/$2
mkdir -p profiles/${T}_pmc_final
ls -t $O/pmc_FETCH_SIZE/runc/*_counter_collection.csv | head -1 | xargs -I{} cp {} profiles/${T}_pmc_final/pmc_fetch_size_bench.csv
ls -t $O/pmc_WRITE_SIZE/runc/*_counter_collection.csv | head -1 | xargs -I{} cp {} profiles/${T}_pmc_final/pmc_write_size_bench.csv
ls -t $O/pmc_sq/runc/*_counter_collection.csv | head -1 | xargs -I{} cp {} profiles/${T}_pmc_final/pmc_sq_bench.csv
cp $O/bench_2rank_one_gpu.json profiles/${T}_bench_2rank_one_gpu_rehearsal.json
python tools/pmc_traffic.py profiles/${T}_pmc_final/pmc_fetch_size_bench.csv profiles/${T}_pmc_final/pmc_write_size_bench.csv profiles/${T}_traffic_bench.json > /dev/null
python tools/pmc_valu.py profiles/${T}_pmc_final/pmc_sq_bench.csv profiles/${T}_valu_bench.json > /dev/null
rm -rf $O
