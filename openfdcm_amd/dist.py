"""Template sharding across the GPUs of a node (one process per GPU, torch.distributed).

Candidates of different templates are independent given the DT3 volume, so the template list is
split into contiguous index ranges, every rank builds the (small-input) DT3 volume itself, and the
only exchange is one gather of the 32-byte match records to rank 0 (per frame: fixed-capacity blocks
with the count in a trailing record, FrameGatherer).  Concatenating the shards in
rank order reproduces the reference's positional order (defaultmatch.cpp:51-86) because the
ranges are contiguous.  Backend "nccl" is RCCL on ROCm; "gloo" runs the same code on CPU tensors
(used by the CPU tests with a stand-in search function).
"""
import sys

import numpy as np

from . import _capi
from ._capi import MATCH_DTYPE

if _capi.loaded() and "torch" not in sys.modules:
    # torch bundles its own HIP runtime; imported after libfdcm_hip.so (which brought /opt/rocm's) it finds no device
    raise ImportError("import torch (or openfdcm_amd.dist) before the first openfdcm_amd call that loads libfdcm_hip.so: "
                      "with the library loaded first, torch's HIP runtime reports no GPU")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

RECORD_BYTES = MATCH_DTYPE.itemsize  # 32


def shard_range(n_templates, rank, world_size):
    """Contiguous [begin, end) template range of `rank`."""
    begin = (n_templates * rank) // world_size
    end = (n_templates * (rank + 1)) // world_size
    return begin, end


def gather_matches(local_records, device, group=None, dst=0):
    """Gather per-rank match records (uint8 tensor of n*32 bytes on `device`) to rank `dst`.

    One all_gather of the counts (8 bytes per rank) sizes the buffers, then one gather of the
    padded record blocks.  Returns a structured numpy array on `dst`, None elsewhere.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = local_records.numel() // RECORD_BYTES
    mine = torch.tensor([n_local], dtype=torch.int64, device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, mine, group=group)
    counts_h = [int(c.item()) for c in counts]
    cap = max(1, max(counts_h)) * RECORD_BYTES
    send = torch.zeros(cap, dtype=torch.uint8, device=device)
    send[: n_local * RECORD_BYTES] = local_records[: n_local * RECORD_BYTES]
    if rank == dst:
        recv = [torch.empty(cap, dtype=torch.uint8, device=device) for _ in range(world)]
        dist.gather(send, recv, dst=dst, group=group)
        parts = [r[: c * RECORD_BYTES].cpu().numpy().view(MATCH_DTYPE) for r, c in zip(recv, counts_h)]
        return np.concatenate(parts) if parts else np.zeros(0, dtype=MATCH_DTYPE)
    dist.gather(send, None, dst=dst, group=group)
    return None


class ShardedSearcher:
    """Per-rank state of a sharded search: the rank's template shard resident in HBM plus a
    device buffer for its match records."""

    def __init__(self, templates, rank, world_size, device):
        from .engine import DeviceTemplates
        self.rank, self.world = rank, world_size
        self.begin, self.end = shard_range(len(templates), rank, world_size)
        self.tset = DeviceTemplates(list(templates[self.begin:self.end]))
        self.device = device
        self._buf = None

    def search_local(self, fm, scene, max_tmpl_lines, max_scene_lines, optimizer, batch_size):
        from .engine import search_into
        n_scene = np.asarray(scene).reshape(4, -1).shape[1]
        cap = max(1, self.tset.capacity(n_scene, max_tmpl_lines, max_scene_lines))
        if self._buf is None or self._buf.numel() < cap * RECORD_BYTES:
            self._buf = torch.empty(cap * RECORD_BYTES, dtype=torch.uint8, device=self.device)
        n = search_into(fm, self.tset, scene, max_tmpl_lines, max_scene_lines, optimizer, batch_size, self.begin,
                        self._buf.data_ptr())
        return self._buf[: n * RECORD_BYTES]

    def search_topk(self, fm, scene, max_tmpl_lines, max_scene_lines, optimizer, batch_size, k, penalty=None, tau=1.0,
                    group=None):
        """Sharded search + device tail: every rank penalises, sorts and keeps its k best in HBM
        (fdcm_topk), and only those k records per rank are gathered.  Returns the global k best on rank 0."""
        from .engine import topk
        local = self.search_local(fm, scene, max_tmpl_lines, max_scene_lines, optimizer, batch_size)
        n = local.numel() // RECORD_BYTES
        best = topk(fm, self.tset, k, penalty, tau, tmpl_index_base=self.begin, device_ptr=self._buf.data_ptr(), n=n)
        if self.world == 1:
            return best
        return gather_topk(best, k, self.device, group=group)

    def search(self, fm, scene, max_tmpl_lines, max_scene_lines, optimizer, batch_size, group=None):
        local = self.search_local(fm, scene, max_tmpl_lines, max_scene_lines, optimizer, batch_size)
        if self.world == 1:
            return local.cpu().numpy().view(MATCH_DTYPE)
        return gather_matches(local, self.device, group=group)


def merge_topk(parts, k):
    """Merge per-rank k-best lists (each ascending by score, ties in positional order) given in rank order:
    a stable sort by score keeps ties in (rank, position) = global positional order."""
    allm = np.concatenate(parts) if parts else np.zeros(0, dtype=MATCH_DTYPE)
    order = np.argsort(allm["score"], kind="stable")
    return allm[order[:k]]


def gather_topk(local_topk, k, device, group=None, dst=0):
    """local_topk: this rank's k best (structured numpy array, <= k records).  Returns the global k best on
    rank `dst` (None elsewhere): k records per rank cross the links instead of every match."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    buf = torch.from_numpy(np.ascontiguousarray(local_topk).view(np.uint8).copy()).to(device)
    res = gather_matches(buf, device, group=group, dst=dst)
    if rank != dst:
        return None
    # gather_matches concatenates the lists in rank order, which is what the stable merge needs
    return merge_topk([res], k)


class FrameShards:
    """Frame sharding across the ranks (the other way to use N GPUs, for a STREAM of frames): frame f of the stream runs
    whole on rank f % world with the whole template list, so nothing is exchanged on the data path and the job's rate is
    N times one GPU's (template shards are bounded by the build every rank repeats).  This class is the bookkeeping only:
    which frames are a rank's, the job's match count, and every frame's digest on rank 0 (for a parity gate)."""

    def __init__(self, rank, world_size, group=None):
        self.rank, self.world, self.group = rank, world_size, group

    def frame_of(self, local_index):
        """Position in the stream of this rank's `local_index`-th frame."""
        return local_index * self.world + self.rank

    def mine(self, n_frames):
        """The frames of a stream of n_frames that this rank runs."""
        return range(self.rank, n_frames, self.world)

    def total(self, local_count, device):
        """Sum of a per-rank count over the ranks (one 8-byte all-reduce, after the timed region)."""
        if self.world == 1 and not dist.is_initialized():
            return int(local_count)
        t = torch.tensor([int(local_count)], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return int(t.item())

    @staticmethod
    def digest(records, below_template=None):
        """(record count, sha256 of the bytes) of a frame's match records, optionally of the templates below an index only."""
        import hashlib
        r = np.asarray(records)
        if below_template is not None:
            r = r[r["tmpl_idx"] < below_template]
        return len(r), hashlib.sha256(np.ascontiguousarray(r).tobytes()).hexdigest()

    def gather_digests(self, frames, below_template=None, dst=0):
        """frames: this rank's [(tag, records)] in the order it ran them.  Returns on rank `dst` a list per rank of
        (tag, count, sha256), None elsewhere."""
        mine = [(tag,) + self.digest(rec, below_template) for tag, rec in frames]
        if not dist.is_initialized():
            return [mine]
        out = [None] * self.world
        dist.all_gather_object(out, mine, group=self.group)  # (all_gather: the object collective every backend has had for longest)
        return out if self.rank == dst else None


class FrameGatherer:
    """Persistent buffers for gathering one frame's match records per call (the per-frame path of the
    pipeline, where allocations, pageable copies and host-side concatenation would cost more than
    the frame).  ONE collective per frame: every rank sends a fixed-capacity block whose trailing
    record carries its record count (the way fdcm_search hands matches and count to the host in one
    copy), so no count exchange precedes the gather.  Rank `dst` then reads the counts out of the
    gathered blocks and delivers the exact list to the host: on a GPU with one library call (fdcm_blocks_to_host: a kernel
    writes the records into pinned memory); on CPU tensors (the gloo tests) with torch copies into a ring of `depth` host
    buffers, where a returned array stays valid until `depth` more frames have been gathered."""

    def __init__(self, capacity_records, device, group=None, dst=0, depth=4):
        self.group, self.dst, self.device = group, dst, device
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.cap = max(1, capacity_records)
        self.blk = (self.cap + 1) * RECORD_BYTES  # records + the trailing count record
        cuda = device.type == "cuda"
        self.counts_host = torch.zeros(self.world, dtype=torch.int64, pin_memory=cuda)
        if self.rank == dst:
            self.recv = torch.empty(self.world * self.blk, dtype=torch.uint8, device=device)
            self.recv_views = list(self.recv.split(self.blk))
            # the count of rank r: first int64 of the trailing record of block r
            self.trailers = self.recv.view(torch.int64).view(self.world, self.blk // 8)[:, self.cap * RECORD_BYTES // 8]
            if not cuda:  # the CPU-tensor path (gloo tests) packs and copies with torch
                self.packed = torch.empty(self.world * self.cap * RECORD_BYTES, dtype=torch.uint8, device=device)
                self.host = [torch.empty(self.world * self.cap * RECORD_BYTES, dtype=torch.uint8) for _ in range(depth)]
                self.turn = 0

    def block_bytes(self):
        return self.blk

    def gather(self, block, n_local):
        """block: this rank's uint8 buffer of block_bytes() bytes whose first n_local records are valid."""
        block[: self.blk].view(torch.int64)[self.cap * RECORD_BYTES // 8] = int(n_local)  # queued on the caller's stream
        dist.gather(block[: self.blk], self.recv_views if self.rank == self.dst else None, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        if self.device.type == "cuda":
            # one library call on torch's stream, behind the gather: a kernel reads the blocks' counts and writes the
            # valid records of all blocks, in rank order, into a pinned host array (no copy commands, one
            # synchronisation); the array is library-owned and stays valid as long as it is referenced
            import ctypes as C
            from .engine import _adopt_matches
            out, n = C.c_void_p(), C.c_int64()
            # the library launches on ITS thread-local device: select the buffers' one (it defaults to 0, and this may be
            # another thread than the one that created the handles)
            _capi.check(_capi.lib().fdcm_set_device(self.device.index if self.device.index is not None else torch.cuda.current_device()))
            _capi.check(_capi.lib().fdcm_blocks_to_host(C.c_void_p(self.recv.data_ptr()), self.world, self.cap,
                                                        C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream),
                                                        C.byref(out), C.byref(n)))
            return _adopt_matches(out, n.value)
        self.counts_host.copy_(self.trailers, non_blocking=True)
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()  # counts on the host (the gather is queued before)
        counts = self.counts_host.tolist()
        total = sum(counts) * RECORD_BYTES
        off = 0
        for r, c in enumerate(counts):  # compaction on the device: one slice per rank, one contiguous list
            nb = c * RECORD_BYTES
            self.packed[off: off + nb].copy_(self.recv_views[r][:nb], non_blocking=True)
            off += nb
        host = self.host[self.turn]
        self.turn = (self.turn + 1) % len(self.host)
        host[:total].copy_(self.packed[:total], non_blocking=True)
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        return host[:total].numpy().view(MATCH_DTYPE)


class ShardedPipeline:
    """Frame pipeline of one rank plus the in-order gather of every frame's match records to rank 0.

    `pipe` is an engine.FramePipeline over the rank's template shard (anything with its
    submit/wait/close works; the CPU tests pass a stand-in).  Worker threads never issue
    collectives: the caller's thread gathers frame by frame, in submission order on every rank,
    while later frames are being computed."""

    def __init__(self, pipe, world_size, device, capacity_records, slots, group=None, gather=None):
        self.pipe, self.world, self.device, self.slots, self.group = pipe, world_size, device, slots, group
        self.bufs = None
        self.gatherer = None
        if world_size > 1 if gather is None else gather:
            # one record buffer per slot (capacity + the trailing count record): ticket t runs on slot t % slots
            self.gatherer = FrameGatherer(capacity_records, device, group=group, depth=slots + 2)
            self.bufs = [torch.empty(self.gatherer.block_bytes(), dtype=torch.uint8, device=device) for _ in range(slots)]
        self.pending = []
        self.submitted = 0
        # per slot: event recorded (on torch's stream) behind the collective that reads the slot's record buffer
        self.sent = [None] * slots

    @classmethod
    def create(cls, searcher, n_scene_lines, depth, coeff, padding, distance, max_tmpl_lines, max_scene_lines,
               optimizer, batch_size, slots, group=None, gather=None):
        from .engine import FramePipeline
        pipe = FramePipeline(searcher.tset, depth=depth, coeff=coeff, padding=padding, distance=distance,
                             max_tmpl_lines=max_tmpl_lines, max_scene_lines=max_scene_lines, optimizer=optimizer,
                             batch_size=batch_size, tmpl_index_base=searcher.begin, slots=slots)
        cap = searcher.tset.capacity(n_scene_lines, max_tmpl_lines, max_scene_lines)
        if searcher.world > 1:
            # the per-frame gather moves equal-sized blocks: every rank uses the largest shard's capacity (shards differ
            # when the template count does not divide by the ranks, or when templates have different line counts)
            c = torch.tensor([cap], dtype=torch.int64, device=searcher.device)
            dist.all_reduce(c, op=dist.ReduceOp.MAX, group=group)
            cap = int(c.item())
        return cls(pipe, searcher.world, searcher.device, cap, slots, group, gather)

    def submit(self, scene_records):
        """Queue one frame (at most `slots` may be uncollected); scene_records: (N, 4) float32."""
        slot = self.submitted % self.slots
        buf = self.bufs[slot] if self.bufs is not None else None
        if self.sent[slot] is not None:
            # The slot's record buffer is the send buffer of the previous frame's gather, which RCCL runs on torch's
            # stream; the library writes the buffer from the slot's own (non-blocking) stream.  On ranks other than
            # the destination nothing else orders the two, so the next frame waits for that send here.
            self.sent[slot].synchronize()
            self.sent[slot] = None
        t = self.pipe.submit(scene_records, buf.data_ptr() if buf is not None else None, prepared=True)
        assert t == self.submitted
        self.submitted += 1
        self.pending.append((t, buf))
        return t

    def collect(self):
        """Wait for the oldest frame; returns its matches on rank 0 (all ranks' shards, rank order).
        With more than one rank the array is a view into a ring of host buffers: it stays valid until
        slots + 2 more frames have been collected (copy it to keep it longer)."""
        t, buf = self.pending.pop(0)
        res = self.pipe.wait(t)
        if buf is None:
            return res
        out = self.gatherer.gather(buf, int(res))
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.sent[t % self.slots] = ev
        return out

    def close(self):
        while self.pending:
            self.collect()
        self.pipe.close()
