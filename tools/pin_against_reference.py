#!/usr/bin/env python3
"""Pin the CPU oracle (and, with a GPU, the HIP path) against the REAL reference, where it is installed.

This image has no Eigen, so the reference could not be built here and the oracle is pinned by the reference's own known-answer
tests plus a second, independent restatement: that leaves large-image float behaviour (Eigen's
LinSpaced in the rasteriser, the order of VectorXf::sum() in evaluate) resting on two restatements that agree.  Anyone with
the reference's wheel (`pip install openfdcm`, v0.10.x) closes that gap with this script: it runs the reference's
build_cpu_featuremap / search on the synthetic BASELINE scenes and on the shipped obj_04 assets and compares, bit for bit,
  * the DT3 volumes (get_dt3_map) with the oracle's, and with the HIP path's when a device is present,
  * the raw match lists (tmpl_idx, score, transform) in positional order, for DefaultSearch(4,4) + BatchOptimize(10) and
    DefaultOptimize,
  * penalize(ExponentialPenalty(1.5)) + sort_matches.
NOT RUN AGAINST THE REFERENCE IN THIS IMAGE (it cannot be installed here: no network); it only uses the reference's documented
Python API (modules/python/src/{core,matching}.cpp), and `--stand-in` runs every line of it with
openfdcm_amd in the reference's place (tests/test_gpu_seam.py).  Exit code 0 = everything identical, 1 = a difference
(printed), 2 = the reference is not importable.

usage: pin_against_reference.py [--configs 1,2] [--templates 200] [--no-gpu]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1,2")
    ap.add_argument("--templates", type=int, default=200)
    ap.add_argument("--no-gpu", action="store_true")
    ap.add_argument("--stand-in", action="store_true",
                    help="exercise this script without the reference: openfdcm_amd (the same API names) plays its part")
    args = ap.parse_args()
    try:
        if args.stand_in:
            import openfdcm_amd as ref
        else:
            import openfdcm as ref  # the reference's pybind11 module
    except ImportError:
        print("the reference's Python module `openfdcm` is not installed: pip install openfdcm (v0.10.x), then run this again")
        return 2
    from openfdcm_amd import lineio, synthetic
    from oracle import oracle as O
    gpu = None
    if not args.no_gpu:
        try:
            import openfdcm_amd as amd
            from openfdcm_amd import _capi
            import ctypes
            n = ctypes.c_int()
            if _capi.lib().fdcm_device_count(ctypes.byref(n)) == 0 and n.value > 0:
                gpu = amd
        except Exception as e:  # no device / no library: the oracle alone is pinned
            print("no HIP device or library:", e)
    bad = 0

    def check(what, ok):
        nonlocal bad
        print(("ok   " if ok else "DIFF ") + what)
        bad += 0 if ok else 1

    def records_of_ref(matches):
        rec = np.zeros(len(matches), dtype=O.MATCH_DTYPE)
        for i, m in enumerate(matches):
            rec[i] = (m.tmpl_idx, m.score, np.asarray(m.transform, dtype=np.float32).reshape(6))
        return rec

    for name in args.configs.split(","):
        if name == "1":  # the notebook's scene and templates (tests/golden/obj_04: the reference's own line files)
            d = os.path.join(ROOT, "tests", "golden", "obj_04")
            scene = lineio.read(os.path.join(d, "scene_0.scene"))
            tmpls = [lineio.read(os.path.join(d, f"template_{i}.tmpl")) for i in range(122)]
            depth, dist_names = 30, ("L2",)
        else:
            cfg = dict(synthetic.CONFIGS[name])
            scene = synthetic.scene(cfg["S"], cfg["scene_lines"], 1)
            tmpls = synthetic.templates(min(args.templates, cfg["T"]), cfg["n"], cfg["S"], 2)
            depth, dist_names = cfg["depth"], ("L2", "L2_SQUARED", "L1")
        for dname in dist_names:
            params = ref.Dt3CpuParameters(depth=depth, dt3Coeff=5.0, padding=1.0, distance=getattr(ref.distance, dname))
            rfm = ref.build_cpu_featuremap(scene, params)
            orc = O.build(scene, depth=depth, coeff=5.0, padding=1.0, distance=getattr(O, dname), nthreads=os.cpu_count())
            rmap = rfm.get_dt3_map()
            keys = sorted(rmap)
            check(f"config {name} {dname}: feature size and scene translation",
                  tuple(int(v) for v in rfm.get_feature_size()) == (orc.W, orc.H)
                  and np.array_equal(np.asarray(rfm.get_scene_translation(), dtype=np.float32), orc.translation))
            same = len(keys) == orc.depth and all(
                np.array_equal(np.asarray(rmap[k], dtype=np.float32).view(np.uint32), orc.slice(i).view(np.uint32)) for i, k in enumerate(keys))
            check(f"config {name} {dname}: DT3 volume, reference vs oracle ({orc.depth} slices of {orc.W} x {orc.H})", same)
            gfm = None
            if gpu is not None:
                gfm = gpu.build_cpu_featuremap(scene, gpu.Dt3CpuParameters(depth=depth, dt3Coeff=5.0, padding=1.0, distance=getattr(gpu.distance, dname)))
                gmap = gfm.get_dt3_map()
                gkeys = sorted(gmap)
                check(f"config {name} {dname}: DT3 volume, reference vs HIP",
                      len(gkeys) == len(keys) and all(np.array_equal(np.asarray(rmap[k], dtype=np.float32).view(np.uint32), gmap[g].view(np.uint32))
                                                      for k, g in zip(keys, gkeys)))
            for oname, okind, make in (("BatchOptimize(10)", O.BATCH_OPTIMIZE, lambda m: m.BatchOptimize(10)),
                                       ("DefaultOptimize", O.DEFAULT_OPTIMIZE, lambda m: m.DefaultOptimize())):
                rm = ref.search(ref.DefaultMatch(), ref.DefaultSearch(4, 4), make(ref), rfm, tmpls, scene)
                rrec = records_of_ref(rm)
                orec = np.asarray(O.search(orc, tmpls, scene, 4, 4, kind=okind, batch=10 if okind == O.BATCH_OPTIMIZE else 1,
                                           nthreads=os.cpu_count()), dtype=O.MATCH_DTYPE)
                check(f"config {name} {dname} {oname}: raw match list, reference vs oracle ({len(rrec)} matches)", rrec.tobytes() == orec.tobytes())
                if gfm is not None:
                    grec = gpu.search(gpu.DefaultMatch(), gpu.DefaultSearch(4, 4), make(gpu), gfm, tmpls, scene).records()
                    check(f"config {name} {dname} {oname}: raw match list, reference vs HIP", rrec.tobytes() == np.ascontiguousarray(grec).tobytes())
            lens = ref.get_template_lengths(tmpls)
            best = ref.sort_matches(ref.penalize(ref.ExponentialPenalty(1.5), rm, lens))
            want = O.sort_matches(O.penalize(rrec, np.asarray(lens, dtype=np.float32), 1.5))
            check(f"config {name} {dname}: sort_matches(penalize(ExponentialPenalty(1.5))), reference vs oracle", records_of_ref(best).tobytes() == want.tobytes())
    print("reference pin:", "all identical" if bad == 0 else f"{bad} difference(s)")
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
