#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

Workload (default: config 2' of BASELINE.md, the configuration BASELINE.json:metric is quoted on): 1024x1024
synthetic scene (200 lines, seed 1), depth 30, L2, coeff 5, padding 1.0; 1000 templates x 32 lines (seed 2) PER GPU;
DefaultSearch(4,4), BatchOptimize(10), DefaultMatch.  A step = one frame: one DT3 feature-map build + one search
over the rank's template shard (+ one RCCL gather of the match records to rank 0 when N > 1), with the matches
delivered to the host.  Inputs are resident before the timed region (templates in HBM; the 3.2 KB scene is handed
over as the C ABI's host pointer and uploaded inside the step).  value = raw matches of all ranks / second.

One frame at this size is latency bound (a sequential envelope per image row, dependent gathers per candidate), so
by default --frames 4 frames are in flight through the library's frame pipeline (include/fdcm.h: each slot has its
own feature map, HIP stream and host worker).  The K timed steps are K frames submitted and collected, in order,
inside the timed region (the pipeline starts and ends empty).  --frames 1 is the blocking rebuild -> search sequence.

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline         the DT3 build: 7 V algorithmic bytes / the span of its kernels with the GPU to itself (blocking
                   frames measured right after the timed region, HIP events on the handle's own stream), against
                   8 TB/s; per-kernel table beside it, and the same stages as timed inside the (overlapped) timed region
  roofline_search  the search kernels: 8 B x translations scored by the reference rule x lines per template
  parity_gate      the GPU's match records against the CPU oracle's, bit for bit -- at N > 1 the GATHERED list of all ranks
                   (rank 0 runs the oracle on the whole job's template list)
  cpu_baseline     the CPU oracle (a port; the reference cannot be built here) on the same job, on rank 0's host cores
  scaling_bounds   what N GPUs can do to a blocking frame (the build is replicated): strong and weak
A failed parity gate makes the run exit non-zero.
"""
import argparse
import hashlib
import json
import os
import subprocess
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
# distance transform = pass 1 writes V + pass 2 reads V and writes V (fused here: the sweep recomputes pass 1 from
# V/16 of column descriptors and never materialises it); propagation reads V and writes V; line integral likewise.
STAGE_BYTES_V = {"pass2_ms": 3.0, "propagate_ms": 2.0, "integral_ms": 2.0}
STAGE_KERNELS = {"seeds_ms": "k_seeds (feature sizes above 4096 px only; below, k_coldesc_tile draws the seeds itself)", "pass1_ms": "k_coldesc_tile",
                 "pass2_ms": "L2 / L2^2: k_sweep_balanced (k_pass2_l2 above 2896 px); L1: k_l1_word_mins + k_l1_carries + k_l1_word",
                 "propagate_ms": "k_propagate_reg", "integral_ms": "k_integral"}
DIST_NAMES = {0: "L2", 1: "L2_SQUARED", 2: "L1"}
# templates per GPU of the BASELINE configs (4 and 5 are sharded over 8 GPUs)
PER_GPU = {"2": 100, "2p": 1000, "3": 1000, "4": 1000, "5": 2000}


def so_hash():
    p = os.path.join(ROOT, "openfdcm_amd", "libfdcm_hip.so")
    return hashlib.sha256(open(p, "rb").read()).hexdigest()[:16]


def device_code_hash(path):
    """sha256[:16] of the library's .hip_fatbin section (the gfx950 code objects): what the GPU runs.  Host-only changes of the
    library leave it alone, so a PMC summary stays valid for them."""
    import struct
    data = open(path, "rb").read()
    if data[:4] != b"\x7fELF" or data[4] != 2:
        return None
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    def sh(i):
        name, typ, flags, addr, off, size = struct.unpack_from("<IIQQQQ", data, shoff + i * shentsize)
        return name, off, size
    _, stroff, strsize = sh(shstrndx)
    for i in range(shnum):
        name, off, size = sh(i)
        end = data.index(b"\0", stroff + name)
        if data[stroff + name:end] == b".hip_fatbin":
            return hashlib.sha256(data[off:off + size]).hexdigest()[:16]
    return None


def pmc_is_of_this_library(doc):
    """A PMC summary counts for the running library when it was measured on the same file, or on the same device code."""
    so = os.path.join(ROOT, "openfdcm_amd", "libfdcm_hip.so")
    return doc.get("so_sha256_16") == so_hash() or (doc.get("device_sha256_16") is not None and doc.get("device_sha256_16") == device_code_hash(so))


def pmc_traffic(config):
    """HBM bytes per build (all build kernels) from a committed PMC summary (profiles/*pmc_traffic*<config>*.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per MI355X_MICROARCH.md).  Counters cannot
    be read from inside the timed run; the value is only reported when the summary was taken on this very library
    (it records the .so's hash), otherwise null with the reason."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_traffic*config{config}*.json")))
    if not files:
        return None, "no PMC summary for this config under profiles/"
    doc = json.load(open(files[-1]))
    if not pmc_is_of_this_library(doc):
        return None, f"{os.path.basename(files[-1])} was measured on another build of libfdcm_hip.so"
    build = [v["hbm_bytes_per_launch"] * v.get("launches_per_frame", 1.0) for k, v in doc["kernels"].items()
             if any(t in k for t in ("k_seeds", "k_coldesc", "k_sweep", "k_pass2", "k_l1", "k_propagate", "k_integral"))]
    return float(sum(build)), os.path.basename(files[-1])


def pmc_search_traffic(config):
    """HBM bytes per launch of k_search from the same summary: (lower, upper) -- its 4-byte gathers produce 64-B and
    128-B read requests that FETCH_SIZE tallies alike (tools/pmc_traffic.py, profiles/r03_fetch_calib.json), so the
    truth lies between the raw counter and twice it; `traffic` reports the upper bound."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_traffic*config{config}*.json")))
    if not files:
        return None, None, "no PMC summary for this config under profiles/"
    doc = json.load(open(files[-1]))
    if not pmc_is_of_this_library(doc):
        return None, None, f"{os.path.basename(files[-1])} was measured on another build of libfdcm_hip.so"
    for k, v in doc["kernels"].items():
        if "k_search" in k:
            lo = v.get("hbm_bytes_per_launch_lower", v["fetch_raw_bytes"] + v["write_bytes"])
            return float(lo), float(v["hbm_bytes_per_launch"]), os.path.basename(files[-1])
    return None, None, "k_search is not in the summary"


def cpu_baseline(cfg, scene, tmpls, sample_templates, reps, warmup=True):
    """Oracle on the host cores: one build + search over the first `sample_templates` templates (by default all of
    them: nothing is scaled), median of `reps`.  The oracle's workers are a long-lived pool, as the reference's
    BS::thread_pool is, so thread start-up is not what is timed.  Returns (json object, the oracle's match records)."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    sub = tmpls[:sample_templates]
    builds, searches = [], []
    if warmup:
        fm = O.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"], nthreads=cores)
    for _ in range(reps):
        t0 = time.perf_counter()
        fm = O.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"], nthreads=cores)
        builds.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        m = O.search(fm, sub, scene, 4, 4, kind=O.BATCH_OPTIMIZE, batch=10, nthreads=cores)
        searches.append(time.perf_counter() - t0)
    t_build, t_search = float(np.median(builds)), float(np.median(searches))
    scale = len(tmpls) / len(sub)
    frame = t_build + t_search * scale
    return {
        "value": len(m) * scale / frame, "unit": "matches/s", "cores": cores, "kind": "port",
        "sample": f"1 DT3 build ({t_build * 1e3:.0f} ms) + search of " + (f"all {len(tmpls)} templates ({t_search * 1e3:.0f} ms)" if scale == 1
                  else f"the first {len(sub)} of {len(tmpls)} templates ({t_search * 1e3:.0f} ms, scaled x{scale:g})") +
                  f" with {cores} threads (a long-lived pool), {1 if warmup else 0} warm-up + median of {reps} run(s) "
                  f"(~{(t_build + t_search) * cores * (reps + (1 if warmup else 0)):.0f} core-seconds in all); the oracle is a restatement "
                  "(the reference cannot be built here) that omits the reference's two O(V) deep copies",
        "dt3_build_ms": t_build * 1e3, "search_matches_per_s": len(m) / t_search,
    }, m


def relaunch_under_torchrun(args):
    """--gpus N without a launcher: start the N ranks as children (before anything touches the GPU) and exit with
    their code, so that a plain `python bench.py --gpus 8` reports an 8-GPU number instead of a 1-GPU one."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: WORLD_SIZE is not set; launching " + " ".join(cmd[1:]), file=sys.stderr)
    sys.exit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="2p", choices=sorted(PER_GPU))
    ap.add_argument("--frames", type=int, default=4, help="frames in flight (1 = blocking rebuild -> search)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 code path (process group, device record buffers, gather) even with one rank")
    ap.add_argument("--single-frames", type=int, default=30,
                    help="blocking frames measured after the timed region for the roofline objects (0 = skip)")
    ap.add_argument("--templates", type=int, default=None, help="templates per GPU (default: the config's)")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="templates in the CPU baseline / parity gate (default -1 = all of the rank's; 0 = skip both)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong", "frames"],
                    help="weak: the config's per-GPU template count on every rank; strong: the config's TOTAL template count "
                         "(BASELINE.md section 5: 1/2/4/8 GPUs on config 2' itself) cut into N shards; frames: every rank holds the "
                         "config's whole template list and takes every N-th frame of the stream, no collective on the data path "
                         "(a step = one frame per rank; the JSON says scaling weak: per-GPU work is fixed)")
    ap.add_argument("--scenes", type=int, default=4,
                    help="distinct scenes cycled through the frames, in the timed region and in the blocking frames the roofline "
                         "is measured on (seeds 1..n; default 4: a caller builds a new scene every frame, matching.cpp:116-130).  "
                         "With more than one, EVERY collected frame of the timed region is checked against the oracle's records of "
                         "its own scene after the run (a frame/slot mix-up in the pipeline or the gather would show)")
    ap.add_argument("--api-frames", type=int, default=20,
                    help="frames of the reference's Python call sequence (build_cpu_featuremap -> search -> get_template_lengths -> "
                         "penalize -> sort_matches through `import openfdcm_amd as openfdcm` only) timed after the run (0 = skip)")
    ap.add_argument("--cpu-reps", type=int, default=5, help="CPU baseline runs (median)")
    ap.add_argument("--cpu-warmup", type=int, default=1, help="0: no warm-up build of the CPU baseline (the large configs: a build takes a minute)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args)

    import torch
    import torch.distributed as dist
    from openfdcm_amd import synthetic
    from openfdcm_amd import _capi
    from openfdcm_amd.dist import FrameShards, ShardedPipeline, ShardedSearcher
    from openfdcm_amd.engine import DeviceFeatureMap, search_raw

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
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
    per_gpu = args.templates or PER_GPU[args.config]
    n_scenes = max(1, args.scenes)
    scenes = [synthetic.scene(cfg["S"], cfg["scene_lines"], 1 + i) for i in range(n_scenes)]
    scene = scenes[0]
    # weak scaling: every rank owns `per_gpu` templates of a global list of world * per_gpu;
    # strong scaling: the global list is the config's own (per_gpu templates in all), cut into `world` contiguous shards
    by_frames = args.scaling == "frames"
    fshards = FrameShards(rank, world)
    total_templates = per_gpu * world if args.scaling == "weak" else per_gpu
    all_templates = synthetic.templates(total_templates, cfg["n"], cfg["S"], 2)
    # template shards: rank r of `world`; frame shards: every rank is "rank 0 of 1" (the whole list, no gather)
    searcher = ShardedSearcher(all_templates, 0 if by_frames else rank, 1 if by_frames else world, device)
    recs = [_capi.as_records(sc) for sc in scenes]
    rec = recs[0]
    frame_no = [0]        # frames submitted so far (selects the scene)
    kept = []             # (scene index, matches) of every timed frame when several scenes are cycled
    F = max(1, args.frames)
    pipe = ShardedPipeline.create(searcher, rec.shape[0], cfg["depth"], 5.0, 1.0, cfg["distance"], 4, 4,
                                  _capi.BATCH_OPTIMIZE, 10, slots=F, gather=use_dist and not by_frames)
    stage_keys = ("seeds_ms", "pass1_ms", "pass2_ms", "propagate_ms", "integral_ms", "total_ms")
    stage_ms = {k: 0.0 for k in stage_keys}
    acc = {"search_kernel_ms": 0.0, "search_total_ms": 0.0, "frames": 0, "n_matches": 0, "evaluations": 0, "last": None}
    submit_t, latency, frame_log = [], [], []  # frame_log: (latency, build span, search span, search kernels) in ms

    def run_frames(n, record):
        """n frames through the pipeline: at most F in flight, collected in submission order."""
        for _ in range(n):
            if len(pipe.pending) == F:
                collect(record)
            submit_t.append(time.perf_counter())
            # (frame shards: this rank's frames are rank, rank + N, ... of the stream)
            scene_q.append((fshards.frame_of(frame_no[0]) if by_frames else frame_no[0]) % n_scenes)
            pipe.submit(recs[scene_q[-1]])
            frame_no[0] += 1
        while pipe.pending:
            collect(record)

    scene_q = []

    def collect(record):
        res = pipe.collect()
        t_sub = submit_t.pop(0)
        si = scene_q.pop(0)
        if record and (n_scenes > 1 or by_frames) and res is not None:
            kept.append((si, res))
        if record:
            latency.append(time.perf_counter() - t_sub)
            bt, stt = pipe.pipe.last_build_timing, pipe.pipe.last_search_timing
            frame_log.append((latency[-1] * 1e3, bt["total_ms"], stt["total_ms"], stt["kernel_ms"]))
            for k in stage_ms:
                stage_ms[k] += bt[k]
            acc["search_kernel_ms"] += stt["kernel_ms"]
            acc["search_total_ms"] += stt["total_ms"]
            acc["evaluations"] = stt["evaluations"]
            acc["frames"] += 1
            if res is not None:
                acc["n_matches"] += len(res)
                acc["last"] = res
                acc["last_scene"] = si

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up (untimed, before the W warm-up steps): torch's lazy HIP initialisation, then every slot allocates its
    # feature map and workspaces on its first frame and runs a few more, so that the pipeline's streams, pinned
    # result buffers and clocks are in the state a running service has them in.
    fence()
    run_frames(int(os.environ.get("BENCH_SETUP_FRAMES", 4 * F)), False)
    run_frames(args.warmup, False)
    # no collector pauses inside the timed region (a generation-2 pass of this process takes milliseconds: several frames)
    import gc
    gc.collect()
    if os.environ.get("BENCH_GC") != "1":
        gc.disable()
    fence()
    t0 = time.perf_counter()
    run_frames(args.steps, True)
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_matches = acc["n_matches"]            # of all K timed frames (they differ when scenes are cycled)
    frame_digests = None
    if by_frames and use_dist:
        # every rank delivered its own frames: the job's matches are the sum, and rank 0 gets every frame's digest for the gate
        total_matches = fshards.total(total_matches, device)
        sample_n = total_templates if args.cpu_sample < 0 else min(args.cpu_sample, total_templates)
        frame_digests = fshards.gather_digests(kept, below_template=sample_n)
    n_matches = total_matches / max(1, args.steps * (world if by_frames else 1))
    gpu_last = None if acc["last"] is None else np.array(acc["last"], copy=True)

    # untimed extra: blocking frames (one in flight, the GPU to itself) for the roofline objects.  The frames cycle the
    # same scenes as the timed region, so every build is a NEW scene for its handle (the launch order and the priority
    # of heavy workgroups come from the handle's previous build, i.e. from another scene); the same-scene figure, where
    # that history fits, is kept beside it.
    single = None
    if rank == 0 and args.single_frames > 0:  # (N > 1: the other ranks wait at the closing barrier meanwhile)
        fm = DeviceFeatureMap.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"])
        count = [0]

        def blocking_frame(cycle=True):
            sc = scenes[count[0] % n_scenes] if cycle else scene
            count[0] += 1
            fm.rebuild(sc)
            return search_raw(fm, searcher.tset, sc, 4, 4, _capi.BATCH_OPTIMIZE, 10, searcher.begin)

        for _ in range(5):
            blocking_frame()
        st = {k: [] for k in stage_keys}
        sk, wall = [], []
        for _ in range(args.single_frames):  # per-stage times: an event between every two kernels of the build
            blocking_frame()
            bt = fm.build_timing()
            for k in st:
                st[k].append(bt[k])
            sk.append(fm.search_timing()["kernel_ms"])
        fm.stage_timing(2)  # the build's span without the events between its stages (3 - 5 us each): first and last event only
        spans, totals, same = [], [], []
        for _ in range(args.single_frames):
            blocking_frame()
            bt = fm.build_timing()
            spans.append(bt["span_ms"])
            totals.append(bt["total_ms"])
        for i in range(args.single_frames + 3):  # the same scene over and over (3 frames to settle its history)
            blocking_frame(cycle=False)
            if i >= 3:
                same.append(fm.build_timing()["span_ms"])
        fm.stage_timing(1)
        same_pass2 = []
        for i in range(max(3, args.single_frames // 3)):  # .. and its dominant kernel's time (an event between the stages)
            blocking_frame(cycle=False)
            same_pass2.append(fm.build_timing()["pass2_ms"])
        fm.stage_timing(False)  # the frame as a caller runs it: no events at all, wall clock around the two calls
        for _ in range(args.single_frames + 3):
            t1 = time.perf_counter()
            blocking_frame()
            wall.append(time.perf_counter() - t1)
        single = {"stage_ms": {k: float(np.mean(v)) for k, v in st.items()}, "search_kernel_ms": float(np.mean(sk)),
                  "span_ms": float(np.mean(spans)), "same_scene_span_ms": float(np.mean(same)), "same_scene_pass2_ms": float(np.mean(same_pass2)),
                  "build_total_ms": float(np.mean(totals)),
                  "frame_ms": float(np.mean(wall[3:])) * 1e3}
        fm.close()

    # untimed extra: the reference's Python call sequence, through `import openfdcm_amd as openfdcm` only
    # (python/src/matching.cpp:116-130,279-307): what a drop-in caller gets per blocking frame
    api = None
    if rank == 0 and args.api_frames > 0:
        import openfdcm_amd as openfdcm
        my_templates = list(all_templates[searcher.begin:searcher.end])
        params = openfdcm.Dt3CpuParameters(depth=cfg["depth"], dt3Coeff=5.0, padding=1.0, distance=openfdcm.distance(cfg["distance"]))
        matcher, strategy, optimizer = openfdcm.DefaultMatch(), openfdcm.DefaultSearch(4, 4), openfdcm.BatchOptimize(10)
        penalty = openfdcm.ExponentialPenalty(tau=1.5)
        parts = {k: [] for k in ("build_cpu_featuremap", "search", "get_template_lengths", "penalize", "sort_matches")}
        api_wall, api_n, api_last = [], 0, None
        for i in range(args.api_frames + 3):
            sc = scenes[i % n_scenes]
            t = [time.perf_counter()]
            featuremap = openfdcm.build_cpu_featuremap(sc, params)
            t.append(time.perf_counter())
            matches = openfdcm.search(matcher, strategy, optimizer, featuremap, my_templates, sc)
            t.append(time.perf_counter())
            lengths = openfdcm.get_template_lengths(my_templates)
            t.append(time.perf_counter())
            penalised = openfdcm.penalize(penalty, matches, lengths)
            t.append(time.perf_counter())
            best = openfdcm.sort_matches(penalised)
            t.append(time.perf_counter())
            if i >= 3:
                api_wall.append(t[-1] - t[0])
                api_n += len(matches)
                for k, (x, y) in zip(parts, zip(t, t[1:])):
                    parts[k].append(y - x)
            api_last = (i % n_scenes, matches, best)
        t1 = time.perf_counter()
        checksum = 0.0
        for m in api_last[2]:
            checksum += m.score
        iterate_ms = (time.perf_counter() - t1) * 1e3
        api = {"frame_ms": float(np.mean(api_wall)) * 1e3, "matches_per_s": api_n / float(np.sum(api_wall)),
               "calls_ms": {k: round(float(np.mean(v)) * 1e3, 4) for k, v in parts.items()},
               "iterate_all_matches_ms": iterate_ms, "frames": args.api_frames, "last": api_last}
        del featuremap
        openfdcm.clear_featuremap_pool()
        openfdcm.clear_template_cache()

    out, gate_failed = None, False
    if rank == 0:
        K = args.steps
        V = 4.0 * cfg["depth"] * cfg["S"] * cfg["S"]
        avg = {k: v / K for k, v in stage_ms.items()}
        kernels_ms = sum(avg[k] for k in ("seeds_ms", "pass1_ms", "pass2_ms", "propagate_ms", "integral_ms"))
        lat = np.array(latency) * 1e3
        dname = DIST_NAMES[cfg["distance"]]
        out = {
            "metric": "template matches/sec (DT3 build + DefaultMatch/BatchOptimize search per frame)",
            "value": total_matches / elapsed, "unit": "matches/s", "n_gpus": world, "steps": K,
            "warmup": args.warmup, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "scaling": "weak" if by_frames else args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE config {args.config.replace('p', chr(39))}: {cfg['S']}x{cfg['S']} scene, "
                                   f"{cfg['scene_lines']} lines, depth {cfg['depth']}, {dname}, " +
                                   (f"{per_gpu} templates x {cfg['n']} lines per GPU" if args.scaling == "weak" else
                                    f"{per_gpu} templates x {cfg['n']} lines on every GPU, {world} frame(s) per step (one per GPU)" if by_frames else
                                    f"{per_gpu} templates x {cfg['n']} lines in all, cut into {world} shard(s)") +
                                   ", DefaultSearch(4,4), BatchOptimize(10)",
                       "templates_total": total_templates, "matches_per_step": n_matches, "frames_in_flight": F,
                       "distinct_scenes": n_scenes,
                       "sharding": "frames" if by_frames else "templates",
                       "parallelism": (f"frames round-robin over {world} GPU(s), the whole template list on each, no collective on the data path"
                                       if by_frames else f"template shards x{world}, DT3 replicated, 1 RCCL gather per frame"),
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "orientation_bins": "host libm" if _capi.lib().fdcm_orientation_bins_mode() else "device atanf"},
            "frame_latency_ms": {"p50": float(np.percentile(lat, 50)), "p95": float(np.percentile(lat, 95)),
                                 "max": float(lat.max()), "note": "submit -> matches on the host, F frames in flight",
                                 # the slowest frame: its position and what its GPU spans were (a stalled host shows
                                 # normal spans, a stalled GPU a long one)
                                 "slowest": (lambda i: {"frame": int(i), "latency": frame_log[i][0], "build_span": frame_log[i][1],
                                                        "search_span": frame_log[i][2], "search_kernels": frame_log[i][3]})(int(np.argmax(lat)))},
            "templates_per_s": total_templates * K * (world if by_frames else 1) / elapsed,
            # BASELINE.json's "DT3 build ms": one blocking build with the GPU to itself, host preparation included (set below from
            # the blocking frames; the span of a build inside the timed region, where F frames share the CUs, is
            # in_timed_region.dt3_build_span_ms)
            "dt3_build_ms": None,
            "in_timed_region": {"note": f"per-launch HIP-event times with {F} frames in flight: launches of concurrent "
                                        "frames share the CUs, so these exceed ms_per_step and the blocking figures",
                                "stage_ms": {k: round(v, 4) for k, v in avg.items()}, "build_kernels_ms": kernels_ms,
                                "dt3_build_span_ms": avg["total_ms"], "search_span_ms": acc["search_total_ms"] / K,
                                "search_kernel_ms": acc["search_kernel_ms"] / K},
        }
        if single:
            s_ms = single["stage_ms"]
            stage_sum = sum(s_ms[k] for k in ("seeds_ms", "pass1_ms", "pass2_ms", "propagate_ms", "integral_ms"))
            span = single["span_ms"]  # first to last kernel of the build, no events in between
            achieved = 7.0 * V / (span * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic(args.config)
            table = {}
            for k, name in STAGE_KERNELS.items():
                b = STAGE_BYTES_V.get(k, 0.0) * V
                table[k[:-3]] = {"kernels": name, "ms": round(s_ms[k], 4), "algorithmic_bytes": b,
                                 "GBps": b / (s_ms[k] * 1e-3) / 1e9 if b and s_ms[k] > 0 else None,
                                 "frac": b / (s_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS if b and s_ms[k] > 0 else None}
            out["roofline"] = {"bound": "hbm", "kernel": "DT3 build (k_seeds .. k_integral)", "achieved": achieved,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                               "traffic": traffic, "traffic_source": traffic_src,
                               "algorithmic_bytes_per_launch": 7.0 * V, "avg_launch_ms": span,
                               "same_scene_ms": single["same_scene_span_ms"], "distinct_scenes": n_scenes,
                               "measured": f"{args.single_frames} blocking frames after the timed region (GPU to itself) cycling the {n_scenes} "
                                           "scenes of the timed region, so each build is a new scene for its handle (same_scene_ms: one scene "
                                           "rebuilt over and over, where the launch order taken from the previous build fits); HIP events on the "
                                           "feature map's own stream: one before the build's first kernel and one behind its last; the stage "
                                           f"table comes from {args.single_frames} more frames with an event between the stages, which cost 3 - 5 us "
                                           f"each (their sum: {stage_sum:.4f} ms)",
                               # the kernel the build's time hangs on, by itself (its 3 V are pass 1's V written + pass 2's V read and V written)
                               "dominant_kernel": {"kernel": STAGE_KERNELS["pass2_ms"], "algorithmic_bytes_per_launch": STAGE_BYTES_V["pass2_ms"] * V,
                                                   "avg_launch_ms": s_ms["pass2_ms"],
                                                   "frac": STAGE_BYTES_V["pass2_ms"] * V / (s_ms["pass2_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if s_ms["pass2_ms"] > 0 else None,
                                                   "same_scene_ms": single["same_scene_pass2_ms"],
                                                   "same_scene_frac": STAGE_BYTES_V["pass2_ms"] * V / (single["same_scene_pass2_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if single["same_scene_pass2_ms"] > 0 else None,
                                                   "note": f"averaged over the {n_scenes} scenes cycled; same_scene = scene seed 1 rebuilt over and over (what rounds 1 - 5 reported); "
                                                           "its time is its slowest workgroup's, which is the scene's (profiles/NOTES.md section 12)"},
                               "stages": table}
            reads = 8.0 * acc["evaluations"] * cfg["n"]
            s_lo, s_hi, s_src = pmc_search_traffic(args.config)
            ska = reads / (single["search_kernel_ms"] * 1e-3) / 1e9
            out["roofline_search"] = {"bound": "hbm", "kernel": "k_pairs + k_worklist + k_search + compaction",
                                      "achieved": ska, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ska / HBM_PEAK_GBS,
                                      "traffic": s_hi, "traffic_lower_bound": s_lo, "traffic_source": s_src,
                                      "algorithmic_bytes_per_launch": reads,
                                      "avg_launch_ms": single["search_kernel_ms"],
                                      "note": "8 B x translations scored by the reference rule x lines per template; "
                                              "random 4-byte gathers move a 64-byte sector each, informational"}
            # what N GPUs can do to one blocking frame: every GPU rebuilds the volume (b), the search (s) is what shards
            b_ms = span
            s_ms_ = max(single["frame_ms"] - span, single["search_kernel_ms"])
            out["scaling_bounds"] = {
                "note": "from this run's blocking frame: build b (replicated on every GPU) and the rest of the frame s (sharded); "
                        "strong = (b + s) / (b + s / N) on the config's own template count, weak = N (per-GPU work fixed; one "
                        "32-byte-record gather per frame is the only shared step)",
                "b_ms": b_ms, "s_ms": s_ms_,
                "strong_speedup_bound": {str(n): (b_ms + s_ms_) / (b_ms + s_ms_ / n) for n in (1, 2, 4, 8)},
                "weak_speedup_bound": {str(n): float(n) for n in (1, 2, 4, 8)},
                # --scaling frames / fdcm_sharded_set_mode(FDCM_SHARD_FRAMES): a stream of frames dealt round-robin, every GPU the whole
                # job of its frames, nothing shared: N x the one-GPU rate on the config's own template count (where template
                # shards are bounded by the replicated build)
                "frames": {str(n): float(n) for n in (1, 2, 4, 8)}}
            # BASELINE.json's "DT3 build ms": host preparation (plan, staging, launches) + the kernels' span of a blocking
            # build of a new scene, as fdcm_featuremap_last_timing reports it; the kernels alone beside it
            out["dt3_build_ms"] = single["build_total_ms"]
            out["dt3_build_kernels_ms"] = span
            out["search_kernels_ms"] = single["search_kernel_ms"]
            out["single_frame_ms"] = single["frame_ms"]
            out["single_frame_matches_per_s"] = n_matches / (single["frame_ms"] * 1e-3)
        if api:
            out["api_frame_ms"] = api["frame_ms"]
            out["api_matches_per_s"] = api["matches_per_s"]
            out["api"] = {"note": "the reference's Python call sequence per blocking frame, through `import openfdcm_amd as openfdcm` only: "
                                  "build_cpu_featuremap -> search -> get_template_lengths -> penalize(ExponentialPenalty(1.5)) -> sort_matches "
                                  f"(python/src/matching.cpp:116-130,279-307), {n_scenes} scenes cycled, mean of {api['frames']} frames; compare "
                                  "with single_frame_ms (rebuild + search through the engine layer, raw records)",
                          "calls_ms": api["calls_ms"], "iterate_all_matches_ms": api["iterate_all_matches_ms"]}
        if out["dt3_build_ms"] is None:  # --single-frames 0: only the contended span is known
            out["dt3_build_ms"] = avg["total_ms"]
        if args.cpu_sample != 0:
            # Rank 0 at every N: the oracle runs the WHOLE job's template list (all shards), so at N > 1 the gate compares the
            # GATHERED list -- every rank's records after the RCCL exchange, in the reference's positional order
            # (defaultmatch.cpp:76-86) -- and the baseline is timed on the same job the N GPUs ran.
            sample = total_templates if args.cpu_sample < 0 else min(args.cpu_sample, total_templates)
            out["cpu_baseline"], want0 = cpu_baseline(cfg, scene, all_templates[:total_templates], sample, args.cpu_reps, bool(args.cpu_warmup))

            def same_as_oracle(got_all, want):
                got = got_all[got_all["tmpl_idx"] < sample]
                return len(got) == len(want) and got.tobytes() == np.asarray(want, dtype=_capi.MATCH_DTYPE).tobytes()

            from oracle import oracle as O
            wants = [want0]
            if n_scenes == 1 and not by_frames:
                same, checked = same_as_oracle(gpu_last, want0), 1
                n_rec = len(want0)
            else:  # every timed frame against the oracle's records of ITS scene
                cores = os.cpu_count() or 1
                for sc in scenes[1:]:
                    ofm = O.build(sc, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"], nthreads=cores)
                    wants.append(O.search(ofm, all_templates[:sample], sc, 4, 4, kind=O.BATCH_OPTIMIZE, batch=10, nthreads=cores))
                same = len(kept) == K and all(same_as_oracle(np.asarray(res), wants[si]) for si, res in kept)
                checked, n_rec = len(kept), sum(len(wants[si]) for si, _ in kept)
                if frame_digests is not None:  # frame shards: every frame of every rank, by digest (rank 0's own were compared above)
                    want_dig = [FrameShards.digest(np.asarray(w, dtype=_capi.MATCH_DTYPE)) for w in wants]
                    for r, frames in enumerate(frame_digests):
                        same = same and len(frames) == K and all((n, h) == want_dig[si] for si, n, h in frames)
                    checked, n_rec = sum(len(f) for f in frame_digests), sum(n for f in frame_digests for _, n, _ in f)
            out["parity_gate"] = "ok" if same else "FAILED"
            out["parity_gate_detail"] = (("every rank's own frames (digests gathered to rank 0): " if frame_digests is not None else
                                          "the gathered list of all ranks: " if world > 1 else "") +
                                         f"match records of the first {sample} templates of "
                                         f"{'the last timed frame' if n_scenes == 1 and not by_frames else f'all {checked} timed frames ({n_scenes} scenes cycled)'}"
                                         f" ({n_rec} records) against the CPU oracle, bit for bit")
            gate_failed = not same
            if api:  # the API leg's last frame: the raw list and the penalised, sorted list against the oracle's
                si, matches, best = api["last"]
                n_own = min(sample, searcher.end)
                want = np.asarray(wants[si], dtype=_capi.MATCH_DTYPE)
                want = want[want["tmpl_idx"] < n_own]
                got = matches.records()
                got = got[got["tmpl_idx"] < n_own]
                lens = np.array([O.eigen_sum(np.array([O.line_props(np.ascontiguousarray(t[:, i]))[1] for i in range(t.shape[1])],
                                                      dtype=np.float32)) for t in all_templates[:n_own]], dtype=np.float32)
                want_best = O.sort_matches(O.penalize(want, lens, 1.5))
                got_best = best.records()
                ok_api = (got.tobytes() == want.tobytes() and
                          (n_own < searcher.end or np.ascontiguousarray(got_best).tobytes() == want_best.tobytes()))
                out["api"]["parity_gate"] = "ok" if ok_api else "FAILED"
                out["api"]["parity_gate_detail"] = (f"search(): {len(want)} records of the last API frame; sort_matches(penalize()): the same list "
                                                    "penalised and sorted by the oracle (std::sort, ties included), bit for bit")
                gate_failed = gate_failed or not ok_api
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
        if gate_failed:
            sys.exit(3)


if __name__ == "__main__":
    main()
