#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

Workload (config 2' of BASELINE.md): 1024x1024 synthetic scene (200 lines, seed 1), depth 30, L2,
coeff 5, padding 1.0; 1000 templates x 32 lines (seed 2) PER GPU; DefaultSearch(4,4),
BatchOptimize(10), DefaultMatch.  A step = one DT3 feature-map build + one search over the rank's
template shard (+ one RCCL gather of the match records to rank 0 when N > 1).  Inputs are resident
before the timed region (templates in HBM; the 3.2 KB scene is handed over as the C ABI's host
pointer and uploaded inside the step).  value = raw matches produced by all ranks / second.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the slowest build kernel: algorithmic bytes / its HIP-event time vs 8 TB/s HBM
  cpu_baseline -- the CPU oracle (a port; the reference cannot be built here) on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0
# Algorithmic bytes per build stage, in units of V = 4*m*W*H (7V in total):
# distance transform = pass 1 writes V + pass 2 reads V and writes V (one fused kernel here, which
# actually moves ~V + V/16); propagation reads V and writes V; line integral reads V and writes V.
STAGE_BYTES_V = {"pass2_ms": 3.0, "propagate_ms": 2.0, "integral_ms": 2.0}
STAGE_KERNEL = {"pass2_ms": "k_pass2_l2", "propagate_ms": "k_propagate_reg", "integral_ms": "k_integral_shallow+steep"}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary (profiles/*pmc_traffic*.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, FETCH doubled per
    MI355X_MICROARCH.md).  Counters cannot be read from inside the timed run, hence the file."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")))
    if not files:
        return None
    ks = json.load(open(files[-1]))["kernels"]
    tot = [v["hbm_bytes_per_launch"] for k, v in ks.items() if any(part in k for part in kernel.split("+"))
           or kernel.split("+")[0].replace("shallow", "") in k]
    return float(sum(tot)) if tot else None


def cpu_baseline(cfg, scene, tmpls, sample_templates):
    """Oracle on the host cores: one build + search over the first `sample_templates` templates."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    fm = O.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"], nthreads=cores)
    t_build = time.perf_counter() - t0
    sub = tmpls[:sample_templates]
    t0 = time.perf_counter()
    m = O.search(fm, sub, scene, 4, 4, kind=O.BATCH_OPTIMIZE, batch=10, nthreads=cores)
    t_search = time.perf_counter() - t0
    scale = len(tmpls) / len(sub)
    frame = t_build + t_search * scale
    return {
        "value": len(m) * scale / frame, "unit": "matches/s", "cores": cores, "kind": "port",
        "sample": f"1 DT3 build ({t_build * 1e3:.0f} ms) + search of the first {len(sub)} of {len(tmpls)} templates "
                  f"({t_search * 1e3:.0f} ms, scaled x{scale:g}) with {cores} threads",
        "dt3_build_ms": t_build * 1e3, "search_matches_per_s": len(m) / t_search,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="2p")
    ap.add_argument("--templates", type=int, default=None, help="templates per GPU (default: the config's)")
    ap.add_argument("--cpu-sample", type=int, default=100, help="templates in the CPU baseline sample (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from openfdcm_amd import synthetic
    from openfdcm_amd import _capi
    from openfdcm_amd.dist import ShardedSearcher
    from openfdcm_amd.engine import DeviceFeatureMap

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    _capi.check(_capi.lib().fdcm_set_device(local_rank))
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    cfg = dict(synthetic.CONFIGS[args.config])
    per_gpu = args.templates or cfg["T"]
    scene = synthetic.scene(cfg["S"], cfg["scene_lines"], 1)
    # weak scaling: every rank owns `per_gpu` templates of a global list of world * per_gpu
    all_templates = synthetic.templates(per_gpu * world, cfg["n"], cfg["S"], 2)
    searcher = ShardedSearcher(all_templates, rank, world, device)
    fm = DeviceFeatureMap.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"])

    def step():
        fm.rebuild(scene)
        return searcher.search(fm, scene, 4, 4, _capi.BATCH_OPTIMIZE, 10)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    stage_ms = {k: 0.0 for k in ("seeds_ms", "pass1_ms", "pass2_ms", "propagate_ms", "integral_ms", "total_ms")}
    search_kernel_ms = search_total_ms = 0.0
    n_matches = 0
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
        bt, stt = fm.build_timing(), fm.search_timing()
        for k in stage_ms:
            stage_ms[k] += bt[k]
        search_kernel_ms += stt["kernel_ms"]
        search_total_ms += stt["total_ms"]
        if res is not None:
            n_matches = len(res)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        K = args.steps
        V = 4.0 * cfg["depth"] * cfg["S"] * cfg["S"]
        avg = {k: v / K for k, v in stage_ms.items()}
        dom = max(STAGE_BYTES_V, key=lambda k: avg[k])
        achieved = STAGE_BYTES_V[dom] * V / (avg[dom] * 1e-3) / 1e9
        kernels_ms = sum(avg[k] for k in ("seeds_ms", "pass1_ms", "pass2_ms", "propagate_ms", "integral_ms"))
        out = {
            "metric": "template matches/sec (DT3 build + DefaultMatch/BatchOptimize search per frame)",
            "value": n_matches * K / elapsed, "unit": "matches/s", "n_gpus": world, "steps": K,
            "warmup": args.warmup, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE config 2': {cfg['S']}x{cfg['S']} scene, {cfg['scene_lines']} lines, "
                                   f"depth {cfg['depth']}, L2, {per_gpu} templates x {cfg['n']} lines per GPU, "
                                   "DefaultSearch(4,4), BatchOptimize(10)",
                       "templates_total": per_gpu * world, "matches_per_step": n_matches,
                       "parallelism": f"template shards x{world}, DT3 replicated, 1 RCCL gather"},
            "dt3_build_ms": avg["total_ms"], "dt3_build_kernels_ms": kernels_ms,
            "dt3_build_GBps_7V": 7.0 * V / (kernels_ms * 1e-3) / 1e9,
            "search_ms": search_total_ms / K, "search_kernel_ms": search_kernel_ms / K,
            "search_matches_per_s": n_matches / (search_total_ms / K * 1e-3) if search_total_ms else None,
            "stage_ms": {k: round(v, 4) for k, v in avg.items()},
            "roofline": {"bound": "hbm", "kernel": STAGE_KERNEL[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(STAGE_KERNEL[dom]),
                         "algorithmic_bytes_per_launch": STAGE_BYTES_V[dom] * V, "avg_launch_ms": avg[dom]},
        }
        if args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(cfg, scene, all_templates[:per_gpu], min(args.cpu_sample, per_gpu))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
