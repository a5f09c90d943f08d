#!/usr/bin/env python3
"""Build (and optionally search) one BASELINE config on the GPU, print stage times, and check the
result against the CPU oracle.  Diagnostic tool (imports oracle/ as the checker only)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="3")
    ap.add_argument("--templates", type=int, default=None)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", default="full", choices=["none", "full", "sample"])
    ap.add_argument("--search", action="store_true")
    ap.add_argument("--perturb", action="store_true",
                    help="also report the build time of a handle's FIRST build, of rebuilds with the SAME scene and of rebuilds "
                         "with a CHANGED scene every time (alternately: every line jittered by up to 3 pixels, and a new seed): "
                         "the L2 sweep takes its launch order from the handle's previous build")
    args = ap.parse_args()
    from openfdcm_amd import synthetic, _capi
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, search_raw
    cfg = dict(synthetic.CONFIGS[args.config])
    T = args.templates or cfg["T"]
    scene = synthetic.scene(cfg["S"], cfg["scene_lines"], 1)
    dev = DeviceFeatureMap.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"])
    stages = []
    for _ in range(args.reps):
        dev.rebuild(scene)
        stages.append(dev.build_timing())
    avg = {k: float(np.median([s[k] for s in stages])) for k in stages[0]}
    V = 4.0 * cfg["depth"] * cfg["S"] ** 2
    stages = ("seeds_ms", "pass1_ms", "pass2_ms", "propagate_ms", "integral_ms")
    kern = sum(avg[k] for k in stages)  # (with an event between the stages: 3 - 5 us each; span_ms is first to last event)
    out = {"config": args.config, "V_MB": V / 1e6, "stage_ms": avg, "kernels_ms": kern,
           "GBps_7V": 7 * V / (kern * 1e-3) / 1e9, "frac_of_8TBps": 7 * V / (kern * 1e-3) / 8e12}
    if args.perturb:
        def kern(t):
            return sum(t[k] for k in stages)
        firsts = []
        for i in range(5):  # fresh handles: no history
            d2 = DeviceFeatureMap.build(synthetic.scene(cfg["S"], cfg["scene_lines"], 1 + i), depth=cfg["depth"], coeff=5.0,
                                        padding=1.0, distance=cfg["distance"])
            firsts.append(kern(d2.build_timing()))
            d2.close()
        same, jitter, reseed = [], [], []
        rng = np.random.default_rng(3)
        base = np.array(scene, dtype=np.float32)
        for i in range(args.reps):
            dev.rebuild(scene); dev.rebuild(scene)
            same.append(kern(dev.build_timing()))
            j = base.copy()
            j[:, 2:] += rng.uniform(-3, 3, size=j[:, 2:].shape).astype(np.float32)  # the two anchor lines keep the size
            j = np.clip(j, 0, cfg["S"] - 1)
            dev.rebuild(j)
            jitter.append(kern(dev.build_timing()))
            dev.rebuild(scene)
            dev.rebuild(synthetic.scene(cfg["S"], cfg["scene_lines"], 100 + i))
            reseed.append(kern(dev.build_timing()))
        out["history"] = {"first_build_ms": float(np.median(firsts)), "same_scene_ms": float(np.median(same)),
                          "jittered_scene_ms": float(np.median(jitter)), "new_scene_ms": float(np.median(reseed)),
                          "note": "kernel spans; jittered = every line moved by up to 3 px since the previous build, new = another random scene of the same size"}
    if args.search:
        tmpls = synthetic.templates(T, cfg["n"], cfg["S"], 2)
        tset = DeviceTemplates(tmpls)
        m = search_raw(dev, tset, scene, 4, 4, _capi.BATCH_OPTIMIZE, 10)
        ts = []
        for _ in range(args.reps):
            m = search_raw(dev, tset, scene, 4, 4, _capi.BATCH_OPTIMIZE, 10)
            ts.append(dev.search_timing())
        out["search"] = {"matches": int(len(m)), "total_ms": float(np.median([t["total_ms"] for t in ts])),
                         "kernel_ms": float(np.median([t["kernel_ms"] for t in ts])),
                         "evaluations": int(ts[-1]["evaluations"]), "candidates": int(ts[-1]["candidates"])}
    if args.check != "none":
        from oracle import oracle as O
        t0 = time.time()
        orc = O.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"], nthreads=os.cpu_count())
        out["oracle_build_s"] = time.time() - t0
        ks = range(orc.depth) if args.check == "full" else sorted(set(np.linspace(0, orc.depth - 1, 7).astype(int)))
        bad = 0
        for k in ks:
            a, b = dev.slice(int(k)), orc.slice(int(k))
            bad += int(np.sum(a.view(np.uint32) != b.view(np.uint32)))
        out["slices_checked"] = len(list(ks))
        out["voxels_differing"] = bad
        if args.search:
            t0 = time.time()
            want = O.search(orc, tmpls, scene, 4, 4, kind=O.BATCH_OPTIMIZE, batch=10, nthreads=os.cpu_count())
            out["oracle_search_s"] = time.time() - t0
            out["match_parity"] = bool(len(want) == len(m) and np.array_equal(want["tmpl_idx"], m["tmpl_idx"])
                                       and np.array_equal(want["score"].view(np.uint32), m["score"].view(np.uint32)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
