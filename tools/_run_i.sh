cd /root/repo
export TMPDIR=/tmp
echo "== fuzz"; timeout 600 python tools/fuzz_parity.py 80 93 2>&1 | tail -1
(time python -m pytest tests -x -q -m gpu 2>&1 | tail -5) 2>&1 | tail -8
for cfg in 5; do
  echo "== config $cfg"; timeout 900 python tools/run_config.py --config $cfg --check sample --reps 7 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', {k: round(v,4) for k,v in d['stage_ms'].items()}, 'diff', d['voxels_differing'], 'frac', round(d['frac_of_8TBps'],3))"
done
for st in 1 0; do for i in 1 2; do
  echo "== config 3 INT_STRIDE=$st"; FDCM_INT_STRIDE=$st timeout 600 python tools/run_config.py --config 3 --check none --reps 15 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', {k: round(v,4) for k,v in d['stage_ms'].items()})"
done; done
for st in 1 0; do
  echo "== config 2 INT_STRIDE=$st"; FDCM_INT_STRIDE=$st timeout 600 python tools/run_config.py --config 2 --check none --reps 15 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', {k: round(v,4) for k,v in d['stage_ms'].items()})"
  echo "== config 5 INT_STRIDE=$st"; FDCM_INT_STRIDE=$st timeout 600 python tools/run_config.py --config 5 --check none --reps 5 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', {k: round(v,4) for k,v in d['stage_ms'].items()})"
done
