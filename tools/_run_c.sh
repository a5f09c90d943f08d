cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "randomised or force_dist" 2>&1 | tail -6
echo "== bench default (20 steps)"; python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-1500
echo "== bench strong"; python bench.py --steps 20 --warmup 5 --scaling strong --cpu-sample 0 2>/dev/null | tail -1 | cut -c1-400
echo "== config 3 perturb"; python tools/run_config.py --config 3 --check none --reps 7 --perturb | cut -c1-900
echo "== config 2 perturb"; python tools/run_config.py --config 2 --check none --reps 7 --perturb | cut -c1-900
