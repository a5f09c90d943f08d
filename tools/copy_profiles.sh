#!/bin/bash
# Copy a collection of tools/collect_profiles.sh from gpurun_out/<dir> into profiles/ with a round prefix.
# usage: bash tools/copy_profiles.sh <dir under gpurun_out> <prefix, e.g. r05>
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/gpurun_out/$1; P=$2
for f in bench_config2p_kernel_stats.csv bench_config2p_frames1_kernel_stats.csv config3_kernel_stats.csv config5_kernel_stats.csv \
         pmc_traffic_config2p.json pmc_traffic_config3.json pmc_traffic_config5.json pmc_sq_config2p.json fetch_calib.json \
         bench_config3.json bench_config5.json bench_1scene.json bench_strong_n1.json bench_frames_n1.json sharded_bench.json history_probe_config2.json history_probe_config3.json int_split_config3.txt \
         run_config3_history.json run_config2_history.json; do
  [ -f $S/$f ] && cp $S/$f $R/profiles/${P}_$f
done
cp $S/bench.json $R/profiles/${P}_bench_config2p.json
cp $S/bench_driver_cmd.json $R/profiles/${P}_bench_driver_cmd.json
cp $S/run_config3.json $R/profiles/${P}_run_config3.json
cp $S/run_config5.json $R/profiles/${P}_run_config5.json
ls $R/profiles | grep "^${P}_" | wc -l
