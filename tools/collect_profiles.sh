#!/bin/bash
# Collect the round's judged measurements on the GPU box into gpurun_out/<tag>/ (copy into profiles/ afterwards).
# usage (inside gpurun): bash tools/collect_profiles.sh r01
R=/root/repo
OUT=$R/gpurun_out/${1:-prof}
mkdir -p $OUT
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python3 $R/bench.py --cpu-sample 0 > /dev/null 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -o ks1 -- python3 $R/bench.py --cpu-sample 0 --frames 1 --single-frames 0 > /dev/null 2>&1
cp $(find /tmp/ks1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_frames1.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pf --output-format csv -- python3 $R/bench.py --frames 1 --steps 5 --warmup 2 --cpu-sample 0 --single-frames 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o pw --output-format csv -- python3 $R/bench.py --frames 1 --steps 5 --warmup 2 --cpu-sample 0 --single-frames 0 > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py /tmp/pf /tmp/pw $OUT/pmc_traffic.json > /dev/null
cut -c1-400 $OUT/bench.json
head -3 $OUT/kernel_stats.csv | cut -c1-120
head -3 $OUT/kernel_stats_frames1.csv | cut -c1-120
