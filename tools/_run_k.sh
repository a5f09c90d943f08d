cd /root/repo
export TMPDIR=/tmp
(time python -m pytest tests -x -q -m gpu 2>&1 | tail -4) 2>&1 | tail -7
for cfg in 2 3 5; do for i in 1 2; do
  echo "== config $cfg"; timeout 900 python tools/run_config.py --config $cfg --check none --reps 9 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', {k: round(v,4) for k,v in d['stage_ms'].items()}, 'frac', round(d['frac_of_8TBps'],3))"
done; done
echo "== bench"; python bench.py --steps 100 --warmup 10 --cpu-sample 0 --single-frames 20 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   value %.1f M  ms/step %.3f  single-frame build %.3f frac %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
