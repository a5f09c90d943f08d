cd /root/repo
export TMPDIR=/tmp
echo "== fuzz"; timeout 600 python tools/fuzz_parity.py 80 91 2>&1 | tail -1
echo "== fuzz redo"; FDCM_K2_SEGMENTS=7 FDCM_K2_FORCE_REDO=3 timeout 600 python tools/fuzz_parity.py 40 92 2>&1 | tail -1
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "staged or kat or precision or config2_build or rebuilds or device_volume" 2>&1 | tail -4
for cfg in 2 3; do
  echo "== config $cfg"; timeout 600 python tools/run_config.py --config $cfg --check full --reps 9 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', {k: round(v,4) for k,v in d['stage_ms'].items()}, 'diff', d['voxels_differing'])"
done
echo "== bench"; python bench.py --steps 100 --warmup 10 --cpu-sample 0 --single-frames 20 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   value %.1f M  ms/step %.3f  single-frame build %.3f frac %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
