#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

Workload (config 2' of BASELINE.md): 1024x1024 synthetic scene (200 lines, seed 1), depth 30, L2,
coeff 5, padding 1.0; 1000 templates x 32 lines (seed 2) PER GPU; DefaultSearch(4,4),
BatchOptimize(10), DefaultMatch.  A step = one frame: one DT3 feature-map build + one search over
the rank's template shard (+ one RCCL gather of the match records to rank 0 when N > 1), with the
matches delivered to the host.  Inputs are resident before the timed region (templates in HBM; the
3.2 KB scene is handed over as the C ABI's host pointer and uploaded inside the step).
value = raw matches produced by all ranks / second.

One frame at this size is latency bound (a sequential envelope per image row, dependent gathers
per candidate), so by default --frames 4 frames are in flight through the library's frame pipeline
(include/fdcm.h: each slot has its own feature map, HIP stream and host worker).  The K timed steps
are K frames submitted and collected, in order, inside the timed region (the pipeline starts and
ends empty).  --frames 1 is the blocking rebuild -> search sequence; its per-frame time is also
measured (untimed extra, "single_frame_ms") so that both modes are on record.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the slowest build kernel: algorithmic bytes / its HIP-event time vs 8 TB/s HBM
  cpu_baseline -- the CPU oracle (a port; the reference cannot be built here) on a bounded sample
"""
import argparse
import json
import os
import sys
import time

# One HIP stream per pipeline slot (plus torch's): ask the runtime for enough hardware queues that
# they do not share one (read at HIP initialisation, i.e. before torch is imported).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0
# Algorithmic bytes per build stage, in units of V = 4*m*W*H (7V in total):
# distance transform = pass 1 writes V + pass 2 reads V and writes V (one fused kernel here, which
# actually moves ~V + V/16); propagation reads V and writes V; line integral reads V and writes V.
STAGE_BYTES_V = {"pass2_ms": 3.0, "propagate_ms": 2.0, "integral_ms": 2.0}
STAGE_KERNEL = {"pass2_ms": "k_pass2_l2", "propagate_ms": "k_propagate_reg", "integral_ms": "k_integral"}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary (profiles/*pmc_traffic*.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, FETCH doubled per
    MI355X_MICROARCH.md).  Counters cannot be read from inside the timed run, hence the file."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")))
    if not files:
        return None
    ks = json.load(open(files[-1]))["kernels"]
    tot = [v["hbm_bytes_per_launch"] for k, v in ks.items() if kernel in k]
    return float(sum(tot)) if tot else None


def cpu_baseline(cfg, scene, tmpls, sample_templates):
    """Oracle on the host cores: one build + search over the first `sample_templates` templates."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    sub = tmpls[:sample_templates]
    builds, searches = [], []
    for _ in range(3):  # median of three: the build is a fraction of a second on a many-core host
        t0 = time.perf_counter()
        fm = O.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"], nthreads=cores)
        builds.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        m = O.search(fm, sub, scene, 4, 4, kind=O.BATCH_OPTIMIZE, batch=10, nthreads=cores)
        searches.append(time.perf_counter() - t0)
    t_build, t_search = sorted(builds)[1], sorted(searches)[1]
    scale = len(tmpls) / len(sub)
    frame = t_build + t_search * scale
    return {
        "value": len(m) * scale / frame, "unit": "matches/s", "cores": cores, "kind": "port",
        "sample": f"1 DT3 build ({t_build * 1e3:.0f} ms) + search of the first {len(sub)} of {len(tmpls)} templates "
                  f"({t_search * 1e3:.0f} ms, scaled x{scale:g}) with {cores} threads, median of 3 runs "
                  f"(~{(t_build + t_search) * cores:.0f} core-seconds each)",
        "dt3_build_ms": t_build * 1e3, "search_matches_per_s": len(m) / t_search,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="2p")
    ap.add_argument("--frames", type=int, default=4, help="frames in flight (1 = blocking rebuild -> search)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 code path (process group, device record buffers, gather) even with one rank")
    ap.add_argument("--single-frames", type=int, default=20,
                    help="frames of the untimed blocking-sequence measurement (0 = skip)")
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
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if "MASTER_ADDR" not in os.environ:  # --force-dist without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=device)

    cfg = dict(synthetic.CONFIGS[args.config])
    per_gpu = args.templates or cfg["T"]
    scene = synthetic.scene(cfg["S"], cfg["scene_lines"], 1)
    # weak scaling: every rank owns `per_gpu` templates of a global list of world * per_gpu
    all_templates = synthetic.templates(per_gpu * world, cfg["n"], cfg["S"], 2)
    searcher = ShardedSearcher(all_templates, rank, world, device)
    from openfdcm_amd.dist import ShardedPipeline
    rec = _capi.as_records(scene)
    F = max(1, args.frames)
    pipe = ShardedPipeline.create(searcher, rec.shape[0], cfg["depth"], 5.0, 1.0, cfg["distance"], 4, 4,
                           _capi.BATCH_OPTIMIZE, 10, slots=F, gather=use_dist)
    stage_ms = {k: 0.0 for k in ("seeds_ms", "pass1_ms", "pass2_ms", "propagate_ms", "integral_ms", "total_ms")}
    acc = {"search_kernel_ms": 0.0, "search_total_ms": 0.0, "frames": 0, "n_matches": 0}

    def run_frames(n, record):
        """n frames through the pipeline: at most F in flight, collected in submission order."""
        for _ in range(n):
            if len(pipe.pending) == F:
                collect(record)
            pipe.submit(rec)
        while pipe.pending:
            collect(record)

    stamps = []

    def collect(record):
        res = pipe.collect()
        stamps.append(time.perf_counter())
        if record:
            bt, stt = pipe.pipe.last_build_timing, pipe.pipe.last_search_timing
            for k in stage_ms:
                stage_ms[k] += bt[k]
            acc["search_kernel_ms"] += stt["kernel_ms"]
            acc["search_total_ms"] += stt["total_ms"]
            acc["frames"] += 1
            if res is not None:
                acc["n_matches"] = len(res)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    fence()               # set-up: torch's lazy HIP initialisation happens here, not next to the timed region
    run_frames(F, False)  # set-up: every slot allocates its feature map and workspaces on its first frame
    run_frames(args.warmup, False)
    fence()
    t0 = time.perf_counter()
    run_frames(args.steps, True)
    fence()
    elapsed = time.perf_counter() - t0
    if os.environ.get("BENCH_DEBUG"):
        d = np.diff(np.array([t0] + stamps[-args.steps:])) * 1e3
        print("collect intervals ms:", np.round(d, 2).tolist(), file=sys.stderr)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_matches = acc["n_matches"]
    search_kernel_ms, search_total_ms = acc["search_kernel_ms"], acc["search_total_ms"]

    # untimed extra: the blocking sequence (one frame in flight) on this rank's shard
    single_frame_ms = None
    if rank == 0 and world == 1 and args.single_frames > 0:
        fm = DeviceFeatureMap.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"])
        from openfdcm_amd.engine import search_raw
        for _ in range(3):
            fm.rebuild(scene)
            search_raw(fm, searcher.tset, scene, 4, 4, _capi.BATCH_OPTIMIZE, 10, searcher.begin)
        single_stage = {k: 0.0 for k in STAGE_BYTES_V}
        t1 = time.perf_counter()
        for _ in range(args.single_frames):
            fm.rebuild(scene)
            search_raw(fm, searcher.tset, scene, 4, 4, _capi.BATCH_OPTIMIZE, 10, searcher.begin)
            bt = fm.build_timing()
            for k in single_stage:
                single_stage[k] += bt[k] / args.single_frames
        single_frame_ms = (time.perf_counter() - t1) / args.single_frames * 1e3
        fm.close()

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
                       "frames_in_flight": F,
                       "parallelism": f"template shards x{world}, DT3 replicated, 1 RCCL gather per frame"},
            "single_frame_ms": single_frame_ms,
            "single_frame_matches_per_s": n_matches / world / (single_frame_ms * 1e-3) if single_frame_ms else None,
            "dt3_build_ms": avg["total_ms"], "dt3_build_kernels_ms": kernels_ms,
            "dt3_build_GBps_7V": 7.0 * V / (kernels_ms * 1e-3) / 1e9,
            "search_ms": search_total_ms / K, "search_kernel_ms": search_kernel_ms / K,
            "search_matches_per_s": n_matches / (search_total_ms / K * 1e-3) if search_total_ms else None,
            "templates_per_s": per_gpu * world * K / elapsed,
            "stage_ms": {k: round(v, 4) for k, v in avg.items()},
            "roofline": {"bound": "hbm", "kernel": STAGE_KERNEL[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(STAGE_KERNEL[dom]),
                         "algorithmic_bytes_per_launch": STAGE_BYTES_V[dom] * V, "avg_launch_ms": avg[dom],
                         "frames_in_flight": F},
        }
        if single_frame_ms:
            # the same kernel with the GPU to itself (untimed extra): launches of concurrent frames share the
            # CUs, so the per-launch duration inside the timed region is longer than this one
            a1 = STAGE_BYTES_V[dom] * V / (single_stage[dom] * 1e-3) / 1e9
            out["roofline_single_frame"] = {"kernel": STAGE_KERNEL[dom], "achieved": a1, "peak": HBM_PEAK_GBS,
                                            "unit": "GB/s", "frac": a1 / HBM_PEAK_GBS,
                                            "avg_launch_ms": single_stage[dom]}
        if args.cpu_sample > 0 and world == 1:  # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(cfg, scene, all_templates[:per_gpu], min(args.cpu_sample, per_gpu))
    pipe.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio, which is flushed at exit when stdout is a pipe:
        # flush it now so that the JSON line is the last line of the output.
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
