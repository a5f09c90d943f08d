cd /root/repo
export TMPDIR=/tmp
for cfg in 2 3 5; do
  echo "== config $cfg default"; timeout 600 python tools/run_config.py --config $cfg --check none --reps 9 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', {k: round(v,4) for k,v in d['stage_ms'].items()})"
  echo "== config $cfg PROP_EXPERIMENT"; FDCM_PROP_EXPERIMENT=1 timeout 600 python tools/run_config.py --config $cfg --check none --reps 9 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', {k: round(v,4) for k,v in d['stage_ms'].items()})"
done
echo "== bench"; python bench.py --steps 100 --warmup 10 --cpu-sample 0 --single-frames 20 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   value %.1f M  ms/step %.3f  single-frame build %.3f frac %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
