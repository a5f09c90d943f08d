cd /root/repo
export TMPDIR=/tmp
for lpt in 1 0; do for s in 4 8; do
echo "== config 3 UNFUSED S=$s LPT=$lpt"; FDCM_K2_UNFUSED=1 FDCM_K2_SEGMENTS=$s FDCM_K2_LPT=$lpt timeout 300 python tools/run_config.py --config 3 --check none --reps 9 | cut -c1-260
done; done
echo "== config 2 UNFUSED"; FDCM_K2_UNFUSED=1 timeout 300 python tools/run_config.py --config 2 --check none --reps 9 | cut -c1-260
(time python -m pytest tests -x -q -m gpu 2>&1 | tail -8) 2>&1
