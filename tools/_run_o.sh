cd /root/repo
export TMPDIR=/tmp
echo "== fuzz"; timeout 600 python tools/fuzz_parity.py 80 101 2>&1 | tail -1
echo "== fuzz flat"; FDCM_SEARCH_FLAT=1 timeout 600 python tools/fuzz_parity.py 30 102 2>&1 | tail -1
python -m pytest tests -x -q -m gpu -k "search or config2p or config3 or config4 or config5 or seam or topk or pipeline" 2>&1 | tail -3
for flat in 0 1; do
  echo "== config 2p search FLAT=$flat"; if [ $flat = 1 ]; then export FDCM_SEARCH_FLAT=1; else unset FDCM_SEARCH_FLAT; fi
  python tools/run_config.py --config 2p --check none --reps 15 --search | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', d['search'])"
  python tools/run_config.py --config 3 --check none --reps 9 --search | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', d['search'])"
  python bench.py --steps 200 --warmup 10 --cpu-sample 0 --single-frames 10 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   value %.1f M  ms/step %.3f single frame %.3f' % (d['value']/1e6, d['ms_per_step'], d['single_frame_ms']))"
done
