#!/bin/bash
# Collect the round's judged measurements on the GPU box into gpurun_out/<tag>/ (copy into profiles/ afterwards).
# usage (inside gpurun): bash tools/collect_profiles.sh <out dir under gpurun_out> [profiles/ prefix, default r06]
# The program goes directly after `--` (no env / bash -c hop under rocprofv3); the queue count bench.py asks for is
# exported here because under rocprofv3 the GPU is initialised before python starts.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${1:-prof}
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
cd /tmp
stats() {  # stats <name> <program args...>: rocprofv3 --kernel-trace --stats summary of a command
  local name=$1; shift
  rm -rf /tmp/ks_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$name -o ks -- "$@" > /dev/null 2>&1
  cp "$(find /tmp/ks_$name -name '*kernel_stats.csv' | head -1)" $OUT/${name}_kernel_stats.csv
}
pmc() {  # pmc <name> <program args...>: HBM bytes per kernel launch (two passes)
  local name=$1; shift
  rm -rf /tmp/pf_$name /tmp/pw_$name
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf_$name -o pf --output-format csv -- "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw_$name -o pw --output-format csv -- "$@" > /dev/null 2>&1
  python3 $R/tools/pmc_traffic.py /tmp/pf_$name /tmp/pw_$name $OUT/pmc_traffic_$name.json "$*" > /dev/null
}
stats bench_config2p python3 $R/bench.py --cpu-sample 0 --api-frames 0
stats bench_config2p_frames1 python3 $R/bench.py --cpu-sample 0 --frames 1 --single-frames 0 --api-frames 0
pmc config2p python3 $R/bench.py --frames 1 --steps 8 --warmup 4 --cpu-sample 0 --single-frames 0 --api-frames 0
# bench.py reports `roofline.traffic` from profiles/*pmc_traffic*config2p*.json when that file was measured on the running
# library: put the fresh one there (this copy of the repository is scratch) before the judged bench lines are taken
cp $OUT/pmc_traffic_config2p.json $R/profiles/${2:-r06}_pmc_traffic_config2p.json
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2>> $OUT/bench.err
for cfg in 3 5; do
  python3 $R/tools/run_config.py --config $cfg --check none --reps 7 --search --templates 200 > $OUT/run_config$cfg.json 2>> $OUT/bench.err
  stats config$cfg python3 $R/tools/run_config.py --config $cfg --check none --reps 7 --search --templates 200
  pmc config$cfg python3 $R/tools/run_config.py --config $cfg --check none --reps 3
  cp $OUT/pmc_traffic_config$cfg.json $R/profiles/${2:-r06}_pmc_traffic_config$cfg.json
done
# round 5: the driver-format lines of the other single-GPU configurations (value, roofline with this library's traffic,
# roofline_search, cpu_baseline with its core count, parity gate over all of the rank's templates); the CPU build of config 5
# takes half a minute on 256 threads, so one timed run without a warm-up there
python3 $R/bench.py --config 3 --steps 100 --warmup 5 --cpu-reps 1 > $OUT/bench_config3.json 2>> $OUT/bench.err
python3 $R/bench.py --config 5 --steps 40 --warmup 4 --cpu-reps 1 --cpu-warmup 0 > $OUT/bench_config5.json 2>> $OUT/bench.err
# round 3: first build / same scene / changed scene (the L2 sweep's launch order comes from the handle's history),
# the FETCH_SIZE calibration, the sharded engine with frames in flight, the multi-scene and strong-scaling bench lines
python3 $R/tools/run_config.py --config 3 --check none --reps 9 --perturb > $OUT/run_config3_history.json 2>> $OUT/bench.err
python3 $R/tools/run_config.py --config 2 --check none --reps 9 --perturb > $OUT/run_config2_history.json 2>> $OUT/bench.err
bash $R/tools/fetch_calib.sh > /dev/null 2>&1 && cp $R/gpurun_out/fetch_calib.json $OUT/fetch_calib.json
python3 $R/tools/sharded_bench.py --frames 4 --steps 200 2>> $OUT/bench.err | grep '^{' | tail -1 > $OUT/sharded_bench.json
# round 6: bench.py cycles four scenes by default (timed region and blocking frames); the one-scene line beside it, the
# frame-sharded schedule at N = 1, the per-stage times by scene, and the line integral's traffic by class of slice
python3 $R/bench.py --scenes 1 --steps 200 --warmup 10 > $OUT/bench_1scene.json 2>> $OUT/bench.err
python3 $R/bench.py --scaling strong --steps 200 --warmup 10 --cpu-sample 0 --api-frames 0 > $OUT/bench_strong_n1.json 2>> $OUT/bench.err
python3 $R/bench.py --scaling frames --steps 200 --warmup 10 --cpu-reps 1 --api-frames 0 > $OUT/bench_frames_n1.json 2>> $OUT/bench.err
python3 $R/tools/history_probe.py 2 > $OUT/history_probe_config2.json 2>> $OUT/bench.err
python3 $R/tools/history_probe.py 3 > $OUT/history_probe_config3.json 2>> $OUT/bench.err
bash $R/tools/int_split.sh 2>> $OUT/bench.err | grep "^k_integral" > $OUT/int_split_config3.txt
bash $R/tools/pmc_sq.sh ${1:-prof}/pmc_sq > /dev/null 2>&1 && cp $OUT/pmc_sq/summary.json $OUT/pmc_sq_config2p.json
cut -c1-300 $OUT/bench.json
for f in $OUT/*_kernel_stats.csv; do echo $f; head -4 $f | cut -c1-110; done
