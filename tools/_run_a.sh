cd /root/repo
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sharded or pipeline or force_dist" 2>&1 | tail -15
bash tools/fetch_calib.sh 2>&1 | tail -60
