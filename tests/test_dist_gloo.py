"""World-size-2 test of the template sharding + match gather on CPU (gloo backend).

The per-rank search is stood in by the CPU oracle on the rank's contiguous template shard (with
tmpl_idx offset by the shard's first template, like fdcm_search's tmpl_index_base); what is under
test is openfdcm_amd.dist: shard ranges, the count exchange, the padded gather and that
concatenating shards in rank order reproduces the single-process positional match list.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT  # noqa: F401
from openfdcm_amd import synthetic
from openfdcm_amd._capi import MATCH_DTYPE
from openfdcm_amd.dist import RECORD_BYTES, gather_matches, shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_templates, out_path):
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S = 128
        scene = synthetic.scene(S, 30, 5)
        tmpls = synthetic.templates(n_templates, 9, S, 6)
        fm = O.build(scene, depth=12, coeff=5.0, padding=1.0)  # every rank builds the DT3 volume itself
        b, e = shard_range(len(tmpls), rank, world)
        local = O.search(fm, tmpls[b:e], scene, 3, 3, kind=O.BATCH_OPTIMIZE, batch=10).astype(MATCH_DTYPE)
        local["tmpl_idx"] += b
        buf = torch.from_numpy(local.view(np.uint8).copy()) if len(local) else torch.zeros(0, dtype=torch.uint8)
        res = gather_matches(buf, torch.device("cpu"))
        if rank == 0:
            full = O.search(fm, tmpls, scene, 3, 3, kind=O.BATCH_OPTIMIZE, batch=10)
            assert res is not None and len(res) == len(full)
            assert np.array_equal(res["tmpl_idx"], full["tmpl_idx"])
            assert np.array_equal(res["score"], full["score"])
            assert np.array_equal(res["transform"], full["transform"])
            np.save(out_path, np.array([len(res)]))
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_templates", [7, 2, 1])
def test_sharded_gather_world2(tmp_path, n_templates):
    out = str(tmp_path / "n.npy")
    mp.spawn(_worker, args=(2, _free_port(), n_templates, out), nprocs=2, join=True)
    assert np.load(out)[0] > 0


def test_shard_ranges_cover_and_are_contiguous():
    for n in (0, 1, 7, 1000, 16001):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, i, w) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1
    assert RECORD_BYTES == 32


class _OraclePipe:
    """Stand-in for engine.FramePipeline on CPU: frames are searched by the oracle at submit time and
    their records written to the caller's buffer; wait() returns the count (device-buffer mode)."""

    def __init__(self, fm_builder, templates, base):
        self.build, self.templates, self.base = fm_builder, templates, base
        self.counts = {}
        self.next = 0

    def submit(self, rec, ptr, prepared=False):
        import ctypes
        from oracle import oracle as O
        assert prepared and ptr
        scene = np.ascontiguousarray(rec.T)
        local = O.search(self.build(scene), self.templates, scene, 3, 3, kind=O.BATCH_OPTIMIZE, batch=10).astype(MATCH_DTYPE)
        local["tmpl_idx"] += self.base
        if len(local):
            ctypes.memmove(ptr, local.ctypes.data, local.nbytes)
        self.counts[self.next] = len(local)
        self.next += 1
        return self.next - 1

    def wait(self, t):
        return self.counts.pop(t)

    def close(self):
        pass


def _pipe_worker(rank, world, port, out_path):
    from oracle import oracle as O
    from openfdcm_amd.dist import ShardedPipeline
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S = 128
        scenes = [synthetic.scene(S, 20 + 3 * i, 7 + i) for i in range(5)]  # frames differ: order must be kept
        tmpls = synthetic.templates(5, 9, S, 6)
        build = lambda sc: O.build(sc, depth=12, coeff=5.0, padding=1.0)  # noqa: E731
        b, e = shard_range(len(tmpls), rank, world)
        slots = 2
        cap = 2 * 3 * 3 * len(tmpls) * 9
        sp = ShardedPipeline(_OraclePipe(build, tmpls[b:e], b), world, torch.device("cpu"), cap, slots)
        got = []
        keep = lambda r: None if r is None else np.array(r, copy=True)  # noqa: E731  (results are views into a ring)
        for sc in scenes:
            if len(sp.pending) == slots:
                got.append(keep(sp.collect()))
            sp.submit(np.ascontiguousarray(np.asarray(sc, dtype=np.float32).T))
        while sp.pending:
            got.append(keep(sp.collect()))
        sp.close()
        if rank == 0:
            assert len(got) == len(scenes)
            for sc, res in zip(scenes, got):
                full = O.search(build(sc), tmpls, sc, 3, 3, kind=O.BATCH_OPTIMIZE, batch=10)
                assert res is not None and res.tobytes() == full.astype(MATCH_DTYPE).tobytes()
            np.save(out_path, np.array([len(got)]))
        else:
            assert all(r is None for r in got)
    finally:
        dist.destroy_process_group()


def test_sharded_pipeline_world2(tmp_path):
    """Frames submitted through ShardedPipeline come back in order, each the full positional list."""
    out = str(tmp_path / "n.npy")
    mp.spawn(_pipe_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert int(np.load(out)[0]) == 5


def _topk_worker(rank, world, port, out_path):
    from oracle import oracle as O
    from openfdcm_amd.dist import gather_topk
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S = 128
        scene = synthetic.scene(S, 30, 5)
        tmpls = synthetic.templates(7, 9, S, 6)
        fm = O.build(scene, depth=12, coeff=5.0, padding=1.0)
        b, e = shard_range(len(tmpls), rank, world)
        local = O.search(fm, tmpls[b:e], scene, 3, 3, kind=O.BATCH_OPTIMIZE, batch=10).astype(MATCH_DTYPE)
        local["tmpl_idx"] += b
        local["score"] = np.round(local["score"], -1)  # force ties across ranks
        k = 9
        mine = local[np.argsort(local["score"], kind="stable")[:k]]  # what fdcm_topk returns per rank
        res = gather_topk(mine, k, torch.device("cpu"))
        if rank == 0:
            full = O.search(fm, tmpls, scene, 3, 3, kind=O.BATCH_OPTIMIZE, batch=10).astype(MATCH_DTYPE)
            full["score"] = np.round(full["score"], -1)
            want = full[np.argsort(full["score"], kind="stable")[:k]]
            assert res.tobytes() == want.tobytes()
            np.save(out_path, np.array([len(res)]))
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


def test_sharded_topk_world2(tmp_path):
    """k best per rank, gathered and merged, equal the k best of the whole list (ties in positional order)."""
    out = str(tmp_path / "n.npy")
    mp.spawn(_topk_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert int(np.load(out)[0]) == 9


def _frames_worker(rank, world, port, out_path):
    """Frame shards (dist.FrameShards, what bench.py --scaling frames and fdcm_sharded_set_mode(FRAMES) do): frame f of the
    stream whole on rank f % world, nothing exchanged on the data path; the count and the digests reach rank 0."""
    from oracle import oracle as O
    from openfdcm_amd.dist import FrameShards
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S = 128
        scenes = [synthetic.scene(S, 24 + 3 * i, 40 + i) for i in range(3)]
        tmpls = synthetic.templates(6, 9, S, 6)
        fs = FrameShards(rank, world)
        n_frames = 7
        assert list(fs.mine(n_frames)) == [f for f in range(n_frames) if f % world == rank]
        assert [fs.frame_of(i) for i in range(len(fs.mine(n_frames)))] == list(fs.mine(n_frames))
        ran, count = [], 0
        for f in fs.mine(n_frames):
            sc = scenes[f % len(scenes)]
            rec = O.search(O.build(sc, depth=12, coeff=5.0, padding=1.0), tmpls, sc, 3, 3, kind=O.BATCH_OPTIMIZE, batch=10).astype(MATCH_DTYPE)
            ran.append((f, rec))
            count += len(rec)
        total = fs.total(count, torch.device("cpu"))
        dig = fs.gather_digests(ran, below_template=4)
        if rank == 0:
            want = []
            for sc in scenes:
                r = O.search(O.build(sc, depth=12, coeff=5.0, padding=1.0), tmpls, sc, 3, 3, kind=O.BATCH_OPTIMIZE, batch=10).astype(MATCH_DTYPE)
                want.append(r)
            assert total == sum(len(want[f % 3]) for f in range(n_frames))
            assert len(dig) == world and sorted(t for d in dig for t, _, _ in d) == list(range(n_frames))
            for r, frames in enumerate(dig):
                assert [t % world for t, _, _ in frames] == [r] * len(frames)
                for t, n, h in frames:
                    assert (n, h) == FrameShards.digest(want[t % 3], 4), (r, t)
            np.save(out_path, np.array([total]))
        else:
            assert dig is None
    finally:
        dist.destroy_process_group()


def test_frame_shards_world2(tmp_path):
    out = str(tmp_path / "frames.npy")
    mp.spawn(_frames_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert np.load(out)[0] > 0
