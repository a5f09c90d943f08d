cd /root/repo
export TMPDIR=/tmp
python tools/_dbg5.py 5 180
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config5 or config3 or staged" 2>&1 | tail -3
