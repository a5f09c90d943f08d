cd /root/repo
export TMPDIR=/tmp
for i in 1 2 3 4; do
  echo "== config 3 run $i"; timeout 600 python tools/run_config.py --config 3 --check none --reps 15 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', {k: round(v,4) for k,v in d['stage_ms'].items()}, 'kern %.3f' % d['kernels_ms'])"
done
mkdir -p gpurun_out/prof_h; rm -rf gpurun_out/prof_h/*
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_h -o ks -- python3 tools/run_config.py --config 3 --check none --reps 20 > /dev/null 2>&1
f=$(find gpurun_out/prof_h -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-160
