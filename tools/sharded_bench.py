#!/usr/bin/env python3
"""Throughput of fdcm_sharded_* (template shards from ONE process, include/fdcm.h) with frames in flight, beside the
blocking calls: config 2' on the devices of this node (one device + FDCM_SHARDED_ALWAYS_COLLECTIVE on a one-GPU box,
so that every frame goes through the grouped RCCL send/recv).  Every collected frame is compared with the single-device
search of its scene.  Run on the GPU box: python tools/sharded_bench.py [--frames 4] [--steps 200] [--devices N]"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--devices", type=int, default=1)
    ap.add_argument("--scenes", type=int, default=4)
    args = ap.parse_args()
    from openfdcm_amd import synthetic, _capi
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, ShardedEngine, search_raw
    cfg = dict(synthetic.CONFIGS["2p"])
    tmpls = synthetic.templates(cfg["T"] * args.devices, cfg["n"], cfg["S"], 2)
    scenes = [synthetic.scene(cfg["S"], cfg["scene_lines"], 1 + i) for i in range(args.scenes)]
    recs = [_capi.as_records(s) for s in scenes]
    eng = ShardedEngine(tmpls, n_devices=args.devices, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"],
                        always_collective=True)
    fm = DeviceFeatureMap.build(scenes[0], depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"])
    tset = DeviceTemplates(tmpls)
    want = []
    for sc in scenes:
        fm.rebuild(sc)
        want.append(np.array(search_raw(fm, tset, sc, 4, 4, _capi.BATCH_OPTIMIZE, 10), copy=True).tobytes())
    out = {"devices": args.devices, "templates": len(tmpls), "scenes": args.scenes}
    for frames in (1, args.frames):
        eng.set_frames_in_flight(frames)
        ok = True
        for timed in (False, True):
            n = args.steps if timed else 4 * frames
            pend, total = [], 0
            t0 = time.perf_counter()
            for i in range(n):
                if len(pend) == frames:
                    t, si = pend.pop(0)
                    got = eng.wait(t)
                    total += len(got)
                    ok = ok and got.tobytes() == want[si]
                si = i % args.scenes
                pend.append((eng.submit(recs[si], 4, 4, _capi.BATCH_OPTIMIZE, 10, prepared=True), si))
            for t, si in pend:
                got = eng.wait(t)
                total += len(got)
                ok = ok and got.tobytes() == want[si]
            dt = time.perf_counter() - t0
        out[f"frames_in_flight_{frames}"] = {"matches_per_s": total / dt, "ms_per_frame": dt / args.steps * 1e3, "every_frame_identical": ok}
    info = eng.info()
    out["collectives"], out["bytes_moved"] = info["collectives"], info["bytes_moved"]
    eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
