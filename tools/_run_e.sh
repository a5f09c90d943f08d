cd /root/repo
export TMPDIR=/tmp
for s in 4 5 6 7; do
  echo "== config 2 S=$s"; FDCM_K2_SEGMENTS=$s timeout 300 python tools/run_config.py --config 2 --check none --reps 15 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   pass2 %.3f total %.3f' % (d['stage_ms']['pass2_ms'], d['kernels_ms']))"
  echo "== bench S=$s"; FDCM_K2_SEGMENTS=$s python bench.py --steps 100 --warmup 10 --cpu-sample 0 --single-frames 10 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   value %.1f M  ms/step %.3f  single-frame build %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
