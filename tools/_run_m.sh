cd /root/repo
export TMPDIR=/tmp
echo "== fuzz"; timeout 600 python tools/fuzz_parity.py 40 95 2>&1 | tail -1
echo "== fuzz LPT=1"; FDCM_K2_LPT=1 timeout 600 python tools/fuzz_parity.py 40 96 2>&1 | tail -1
echo "== config 3 perturb"; python tools/run_config.py --config 3 --check none --reps 7 --perturb | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', d['history'])"
echo "== config 3 perturb LPT=0"; FDCM_K2_LPT=0 python tools/run_config.py --config 3 --check none --reps 7 --perturb | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', d['history'])"
echo "== config 3 first build parity"; python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, '.')
from openfdcm_amd import synthetic
from openfdcm_amd.engine import DeviceFeatureMap
from oracle import oracle as O
cfg = synthetic.CONFIGS["3"]
sc = synthetic.scene(cfg["S"], cfg["scene_lines"], 5)
dev = DeviceFeatureMap.build(sc, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"])
orc = O.build(sc, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"], nthreads=os.cpu_count())
bad = sum(int(np.sum(dev.slice(k).view(np.uint32) != orc.slice(k).view(np.uint32))) for k in range(0, 60, 7))
print("   first build (proxy order), scene seed 5: differing voxels in 9 slices:", bad)
PY
