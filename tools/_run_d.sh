cd /root/repo
export TMPDIR=/tmp
echo "== fuzz"; timeout 600 python tools/fuzz_parity.py 60 81 2>&1 | tail -1
for cfg in 3 2; do
for s in 4 6 8; do
  echo "== config $cfg S=$s"; FDCM_K2_SEGMENTS=$s timeout 300 python tools/run_config.py --config $cfg --check none --reps 9 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   pass2 %.3f total %.3f' % (d['stage_ms']['pass2_ms'], d['kernels_ms']))"
done; done
echo "== config 3 full check"; timeout 600 python tools/run_config.py --config 3 --check full --reps 5 | cut -c1-420
echo "== config 2 full check"; timeout 600 python tools/run_config.py --config 2 --check full --reps 5 | cut -c1-420
