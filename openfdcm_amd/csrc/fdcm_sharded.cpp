// fdcm_sharded.cpp -- template shards over the GPUs of one node from ONE process (fdcm_sharded_* in include/fdcm.h).
//
// candidates of different templates are independent given the DT3 volume, so the template
// list is split into contiguous index ranges, one per device; every device rebuilds the volume itself from the scene
// lines (16 B per line; cheaper than moving V over a 153 GB/s xGMI link) and searches its range.  The reference's own
// parallel seam is the per-candidate task loop of optimize<BatchOptimize> (batchoptimize.cpp:102-114); here the seam
// is the template index.  Long-lived host threads, one per device and frame slot, run rebuild -> search (the blocking
// entry points of fdcm.h); fdcm_sharded_submit hands a frame to the slot workers of every device and returns,
// fdcm_sharded_wait collects it: the match records -- or, in top-k mode, the k best of every shard -- travel to the
// first device in ONE grouped RCCL send/recv, issued by the caller's thread while the workers compute later frames.  The counts need no exchange: all shards live in this process, so every transfer has its exact size
// and lands at its final offset; concatenation in shard order is the reference's positional order because the ranges
// are contiguous (defaultmatch.cpp:51-86).
//
// RCCL is bound at run time (dlopen of librccl.so.1) so that the library neither needs it for single-GPU use nor
// brings a second copy into a process that already holds one (PyTorch ships its own).
#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fdcm_internal.h"

namespace {

// ---- the few RCCL entry points this file uses (signatures of rccl.h; ncclResult_t 0 = success)
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
constexpr ncclDataType_t kNcclUint8 = 1;  // ncclUint8 / ncclChar family: rccl.h enum ncclDataType_t {ncclInt8 = 0, ncclUint8 = 1, ...}

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) { r.error = std::string("RCCL not found: ") + dlerror(); return; }
        auto sym = [&](const char* s) { void* p = dlsym(r.handle, s); if (!p) r.error = std::string("RCCL symbol missing: ") + s; return p; };
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return r;
}

void nccl_check(ncclResult_t rc, const char* what) {
    if (rc != 0) throw std::string("RCCL error in ") + what + ": " + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
}

// One frame slot of a shard: its own feature map (volume, workspaces, stream), result buffers and a long-lived host
// thread that runs  rebuild -> search (-> device tail)  for the frames it is handed.  Ticket t of the engine runs on
// slot t % n_slots of EVERY shard, so the slots of one index form a frame; frames on different slots overlap on the
// devices, and the exchange of frame i (caller's thread, in fdcm_sharded_wait) runs while the workers compute i + 1...
struct Job {
    std::vector<float> scene;
    int64_t n_scene = 0, maxT = 0, maxS = 0, batch = 1, k = 0;
    int optimizer = 0, penalty = -1;
    float tau = 1.f;
    bool topk = false;
    bool whole = false;   // frame-sharded mode: this device runs the frame over the WHOLE template list
};

struct FrameSlot {
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    bool has_job = false, done = true, quit = false;
    Job job;
    fdcm_featuremap* fm = nullptr;
    fdcm::DevBuf block;                // this slot's match records (search capacity of the frame)
    fdcm::DevBuf best;                 // top-k mode: the k best
    // per-frame results of the worker
    int rc = FDCM_OK;
    std::string error;
    int64_t n = 0;                     // records to send (matches, or min(k, matches) in top-k mode)
    const fdcm_match* send_from = nullptr;
    fdcm_build_timing bt{};
    fdcm_search_timing st{};
};

struct Shard {
    int device = 0;
    int64_t begin = 0, end = 0;        // template range
    fdcm_templates* tset = nullptr;
    fdcm_templates* full = nullptr;    // frame-sharded mode: the whole template list on this device (made by fdcm_sharded_set_mode)
    hipStream_t stream = nullptr;      // the collective's stream on this device
    std::vector<std::unique_ptr<FrameSlot>> slots;
    fdcm_build_timing bt{};            // of the frame collected last
    fdcm_search_timing st{};
};

// The caller's device (the library's thread-local one and HIP's) is restored when an entry point returns: the engine
// switches devices on the caller's thread for allocations, streams and the exchange.
struct DeviceGuard {
    int lib_dev = 0, hip_dev = 0;
    bool have_hip = false;
    DeviceGuard() {
        (void)fdcm_get_device(&lib_dev);
        have_hip = hipGetDevice(&hip_dev) == hipSuccess;
    }
    ~DeviceGuard() {
        (void)fdcm_set_device(lib_dev);  // sets HIP's current device too
        if (have_hip) (void)hipSetDevice(hip_dev);
    }
};

}  // namespace

struct fdcm_sharded {
    std::vector<Shard> shards;
    std::vector<ncclComm_t> comms;
    bool always_collective = false;
    bool same_device_ok = false;   // FDCM_SHARDED_ALLOW_SAME_DEVICE (test hook)
    int64_t depth = 0;
    float coeff = 0.f, padding = 0.f;
    int distance = 0;
    int n_slots = 0;
    int mode = FDCM_SHARD_TEMPLATES;    // fdcm_sharded_set_mode
    std::vector<float> all_lines;       // host copy of the template list (the frame-sharded mode uploads it whole to every device)
    std::vector<int64_t> all_offsets;
    int64_t next_ticket = 0;
    // Ticket t lives in entry `entry_of(t)`: template shards: slot t % n_slots of EVERY shard (n_slots entries);
    // frame shards: slot (t / N) % n_slots of shard t % N (N * n_slots entries, shard-major).
    std::vector<int64_t> slot_ticket;   // ticket held by every entry (-1 = free)
    std::vector<Job> slot_job;          // what that ticket asked for (the tail of a top-k frame needs k)
    size_t n_dev() const { return shards.size(); }
    size_t entry_of(int64_t t) const {
        return mode == FDCM_SHARD_FRAMES ? (size_t)(t % (int64_t)n_dev()) * (size_t)n_slots + (size_t)((t / (int64_t)n_dev()) % n_slots)
                                         : (size_t)(t % n_slots);
    }
    size_t shard_of(int64_t t) const { return (size_t)(t % (int64_t)n_dev()); }               // frame shards only
    size_t slot_of(int64_t t) const { return mode == FDCM_SHARD_FRAMES ? (size_t)((t / (int64_t)n_dev()) % n_slots) : (size_t)(t % n_slots); }
    size_t n_entries() const { return mode == FDCM_SHARD_FRAMES ? n_dev() * (size_t)n_slots : (size_t)n_slots; }
    fdcm::DevBuf gathered;   // on the first device: all shards' records back to back
    int64_t collectives = 0; // grouped send/recv operations issued so far
    int64_t bytes_moved = 0; // bytes that crossed between devices (or through RCCL) so far
};

namespace fdcm {
const char* last_error_cstr();
}

using namespace fdcm;

namespace {

template <class F>
int guarded_s(F&& f) {
    try {
        f();
        return FDCM_OK;
    } catch (const HipError& e) {
        set_error(std::string("HIP error: ") + hipGetErrorString(e.code) + " in " + e.what);
        return FDCM_EHIP;
    } catch (const std::string& s) {
        set_error(s);
        return FDCM_EINVAL;
    } catch (const std::exception& e) {
        set_error(e.what());
        return FDCM_EINTERNAL;
    } catch (...) {
        set_error("unknown error");
        return FDCM_EINTERNAL;
    }
}

// rebuild -> search on one slot of one shard; top-k mode adds the device tail.  Never throws: whatever goes wrong is
// recorded in the slot (a worker thread has nobody to throw to).
void run_slot_frame(fdcm_sharded* s, Shard& sh, FrameSlot& fs) {
    fs.rc = FDCM_OK; fs.n = 0; fs.send_from = nullptr; fs.error.clear();
    const Job& j = fs.job;
    const int rc = guarded_s([&] {
        auto ok = [&](int r) { if (r != FDCM_OK) throw std::string(fdcm_last_error()); };
        const float* scene = j.scene.data();
        ok(fs.fm ? fdcm_featuremap_rebuild(fs.fm, scene, j.n_scene)
                 : fdcm_featuremap_build(scene, j.n_scene, s->depth, s->coeff, s->padding, s->distance, &fs.fm));
        fs.fm->shares_gpu = s->n_slots > 1;  // (tunes the next builds of this slot)
        const fdcm_templates* tset = j.whole ? sh.full : sh.tset;
        const int32_t base = j.whole ? 0 : (int32_t)sh.begin;
        int64_t cap = 0;
        ok(fdcm_search_capacity(tset, j.n_scene, j.maxT, j.maxS, &cap));
        FDCM_HIP(hipSetDevice(sh.device));
        fs.block.reserve(std::max<size_t>(32, (size_t)cap * sizeof(fdcm_match)));
        int64_t n = 0;
        ok(fdcm_search_device(fs.fm, tset, scene, j.n_scene, j.maxT, j.maxS, j.optimizer, j.batch, base,
                              fs.block.as<fdcm_match>(), &n));
        (void)fdcm_featuremap_last_timing(fs.fm, &fs.bt);
        (void)fdcm_search_last_timing(fs.fm, &fs.st);
        fs.n = n;
        fs.send_from = fs.block.as<fdcm_match>();
        if (j.topk) {
            const int64_t kk = std::min<int64_t>(std::max<int64_t>(j.k, 0), n);
            fs.best.reserve(std::max<size_t>(32, (size_t)kk * sizeof(fdcm_match)));
            fdcm::run_topk_device(fs.fm, tset, fs.block.as<fdcm_match>(), n, base, j.penalty, j.tau, kk,
                                  fs.best.as<fdcm_match>());
            fs.n = kk;
            fs.send_from = fs.best.as<fdcm_match>();
        }
    });
    if (rc != FDCM_OK) { fs.rc = rc; fs.error = fdcm_last_error(); fs.n = 0; fs.send_from = nullptr; }
}

void slot_worker(fdcm_sharded* s, Shard* sh, FrameSlot* fs) {
    (void)fdcm_set_device(sh->device);
    std::unique_lock<std::mutex> lk(fs->mu);
    while (true) {
        fs->cv.wait(lk, [&] { return fs->has_job || fs->quit; });
        if (fs->quit) break;
        fs->has_job = false;
        lk.unlock();
        try {
            run_slot_frame(s, *sh, *fs);
        } catch (...) {  // (run_slot_frame catches everything itself; a worker must never take the process down)
            fs->rc = FDCM_EINTERNAL; fs->error = "unknown error in a shard worker"; fs->n = 0;
        }
        lk.lock();
        fs->done = true;
        fs->cv.notify_all();
    }
    lk.unlock();
    if (fs->fm) (void)fdcm_featuremap_free(fs->fm);
    fs->fm = nullptr;
    (void)hipSetDevice(sh->device);
    fs->block.release();
    fs->best.release();
}

void stop_workers(fdcm_sharded* s) {
    for (auto& sh : s->shards) {
        for (auto& sp : sh.slots) {
            FrameSlot& fs = *sp;
            {
                std::unique_lock<std::mutex> lk(fs.mu);
                fs.cv.wait(lk, [&] { return fs.done; });  // a frame in flight finishes first
                fs.quit = true;
                fs.cv.notify_all();
            }
            if (fs.worker.joinable()) fs.worker.join();
        }
        sh.slots.clear();
    }
    s->n_slots = 0;
    s->slot_ticket.clear();
    s->slot_job.clear();
}

void start_workers(fdcm_sharded* s, int n_slots) {
    stop_workers(s);
    for (auto& sh : s->shards)
        for (int i = 0; i < n_slots; ++i) {
            sh.slots.emplace_back(new FrameSlot());
            FrameSlot* fs = sh.slots.back().get();
            try {
                fs->worker = std::thread(slot_worker, s, &sh, fs);
            } catch (...) {  // thread creation failed: the ones already running are joined by stop_workers
                sh.slots.pop_back();
                s->n_slots = n_slots;
                stop_workers(s);
                throw std::string("could not start a shard worker thread");
            }
        }
    s->n_slots = n_slots;
    s->slot_ticket.assign(s->n_entries(), -1);
    s->slot_job.assign(s->n_entries(), Job{});
}

// Sizes the device memory of slot `si` of one shard for the frame: the record buffers take the frame's search capacity, the
// slot's feature map every buffer its build and its search will ask for (reservations only: nothing is built or queued, and
// a scene that cannot be built still fails where it always did -- in the frame, reported by its wait).
void reserve_slot(fdcm_sharded* s, Shard& sh, size_t si, const Job& job) {
    FrameSlot& fs = *sh.slots[si];
    const fdcm_templates* tset = job.whole ? sh.full : sh.tset;
    int64_t cap = 0;
    if (fdcm_search_capacity(tset, job.n_scene, job.maxT, job.maxS, &cap) != FDCM_OK) throw std::string(fdcm_last_error());
    FDCM_HIP(hipSetDevice(sh.device));
    (void)fdcm_set_device(sh.device);
    fs.block.reserve(std::max<size_t>(32, (size_t)cap * sizeof(fdcm_match)));
    if (job.topk) fs.best.reserve(std::max<size_t>(32, (size_t)std::min<int64_t>(std::max<int64_t>(job.k, 0), cap) * sizeof(fdcm_match)));
    if (!fs.fm && fdcm_featuremap_build(job.scene.data(), 0, s->depth, s->coeff, s->padding, s->distance, &fs.fm) != FDCM_OK)
        throw std::string(fdcm_last_error());  // (an empty handle: no lines, no volume)
    try {
        BuildPlan plan;
        make_build_plan(job.scene.data(), s->depth > 0 ? job.n_scene : 0, s->depth, s->coeff, s->padding, plan);
        run_build(fs.fm, plan, 3, /*reserve_only=*/true);
        reserve_search(fs.fm, tset, job.n_scene, job.maxT, job.maxS);
    } catch (const std::string&) {  // e.g. a feature size the build rejects: the frame reports it
    } catch (const HipError&) {     // an allocation that failed: the frame's own build asks again and reports it
        (void)hipGetLastError();
    }
}

void hand_over(FrameSlot& fs, const Job& job) {
    std::unique_lock<std::mutex> lk(fs.mu);
    fs.job = job;  // (a copy per shard: a few KB of scene lines)
    fs.done = false;
    fs.has_job = true;
    fs.cv.notify_all();
}

int64_t submit_frame(fdcm_sharded* s, Job&& job) {
    DeviceGuard guard;  // the loop below switches devices on the caller's thread (also restored when it throws)
    if (s->n_slots == 0) start_workers(s, 1);
    const int64_t t = s->next_ticket;
    const size_t ei = s->entry_of(t), si = s->slot_of(t);
    if (s->slot_ticket[ei] >= 0) throw std::string("every frame slot holds a frame that has not been waited for");
    // Device memory of the slot is sized HERE, on the caller's thread, before its workers get the frame: an allocation is a
    // device-wide synchronisation, and the caller's thread is also the one that runs the exchange of earlier frames (in
    // wait) -- so an allocation can never race a grouped send/recv in flight.  A later frame with a larger feature size or
    // more scene lines grows the buffers here the same way.
    if (s->mode == FDCM_SHARD_FRAMES) {  // the whole frame on one device, the whole template list
        job.whole = true;
        Shard& sh = s->shards[s->shard_of(t)];
        reserve_slot(s, sh, si, job);
        hand_over(*sh.slots[si], job);
    } else {
        for (auto& sh : s->shards) reserve_slot(s, sh, si, job);
        for (auto& sh : s->shards) hand_over(*sh.slots[si], job);
    }
    s->slot_job[ei] = std::move(job);
    s->slot_job[ei].scene.clear();
    s->slot_ticket[ei] = t;
    return s->next_ticket++;
}

// Waits for every shard's worker of the slot; throws the first failure after all of them have finished.
void wait_slot(fdcm_sharded* s, size_t si) {
    std::string err;
    for (auto& sh : s->shards) {
        FrameSlot& fs = *sh.slots[si];
        std::unique_lock<std::mutex> lk(fs.mu);
        fs.cv.wait(lk, [&] { return fs.done; });
        if (fs.rc != FDCM_OK && err.empty()) err = std::string("shard on device ") + std::to_string(sh.device) + ": " + fs.error;
        sh.bt = fs.bt; sh.st = fs.st;
    }
    if (!err.empty()) throw err;
}

// All shards' records of one frame slot to the first device, back to back in shard order: ONE grouped send/recv with
// exact sizes.  Returns the total record count; the records are at s->gathered on device shards[0].device, complete
// when the first shard's stream has been synchronised (done here).
int64_t gather_to_first(fdcm_sharded* s, size_t si) {
    std::vector<int64_t> off(s->shards.size() + 1, 0);
    for (size_t i = 0; i < s->shards.size(); ++i) off[i + 1] = off[i] + s->shards[i].slots[si]->n;
    const int64_t total = off.back();
    Shard& root = s->shards[0];
    FDCM_HIP(hipSetDevice(root.device));
    s->gathered.reserve(std::max<size_t>(32, (size_t)total * sizeof(fdcm_match)));
    fdcm_match* dst = s->gathered.as<fdcm_match>();
    if (s->comms.empty()) {  // nothing to exchange between devices: every shard's records already are on the first device
        for (size_t i = 0; i < s->shards.size(); ++i) {
            const FrameSlot& fs = *s->shards[i].slots[si];
            if (fs.n) FDCM_HIP(hipMemcpyAsync(dst + off[i], fs.send_from, (size_t)fs.n * sizeof(fdcm_match), hipMemcpyDeviceToDevice, root.stream));
        }
        FDCM_HIP(hipStreamSynchronize(root.stream));
        return total;
    }
    Rccl& R = rccl();
    bool any = false;
    nccl_check(R.GroupStart(), "ncclGroupStart");
    for (size_t i = 0; i < s->shards.size(); ++i) {
        Shard& sh = s->shards[i];
        const FrameSlot& fs = *sh.slots[si];
        if (fs.n == 0) continue;
        const size_t bytes = (size_t)fs.n * sizeof(fdcm_match);
        nccl_check(R.Send(fs.send_from, bytes, kNcclUint8, 0, s->comms[i], sh.stream), "ncclSend");
        nccl_check(R.Recv(dst + off[i], bytes, kNcclUint8, (int)i, s->comms[0], root.stream), "ncclRecv");
        s->bytes_moved += (int64_t)bytes;
        any = true;
    }
    nccl_check(R.GroupEnd(), "ncclGroupEnd");
    if (any) ++s->collectives;
    for (auto& sh : s->shards) {  // the slot's send buffers are reused by its next frame
        FDCM_HIP(hipSetDevice(sh.device));
        FDCM_HIP(hipStreamSynchronize(sh.stream));
    }
    return total;
}

fdcm_match* download(fdcm_sharded* s, int64_t total) {
    Shard& root = s->shards[0];
    fdcm_match* out = result_acquire(std::max<size_t>(1, (size_t)total) * sizeof(fdcm_match));
    try {
        FDCM_HIP(hipSetDevice(root.device));
        if (total) {
            records_to_host(root.stream, s->gathered.as<fdcm_match>(), total, out);
            FDCM_HIP(hipStreamSynchronize(root.stream));
        }
    } catch (...) {
        result_release(out);
        throw;
    }
    return out;
}

void collect_frame(fdcm_sharded* s, int64_t ticket, fdcm_match** out, int64_t* n_out) {
    *out = nullptr; *n_out = 0;
    if (ticket < 0 || s->n_slots == 0) throw std::string("unknown ticket");
    const size_t ei = s->entry_of(ticket), si = s->slot_of(ticket);
    if (s->slot_ticket[ei] != ticket) throw std::string("unknown or already collected ticket");
    s->slot_ticket[ei] = -1;  // whatever happens below, the slot is free again
    const Job& job = s->slot_job[ei];
    if (s->mode == FDCM_SHARD_FRAMES) {
        // the frame ran whole on one device: its records (or its k best, already in order) go to the host from there, no exchange
        Shard& sh = s->shards[s->shard_of(ticket)];
        FrameSlot& fs = *sh.slots[si];
        {
            std::unique_lock<std::mutex> lk(fs.mu);
            fs.cv.wait(lk, [&] { return fs.done; });
        }
        sh.bt = fs.bt; sh.st = fs.st;
        if (fs.rc != FDCM_OK) throw std::string("shard on device ") + std::to_string(sh.device) + ": " + fs.error;
        fdcm_match* res = result_acquire(std::max<size_t>(1, (size_t)fs.n) * sizeof(fdcm_match));
        try {
            FDCM_HIP(hipSetDevice(sh.device));
            if (fs.n) {
                records_to_host(sh.stream, fs.send_from, fs.n, res);
                FDCM_HIP(hipStreamSynchronize(sh.stream));
            }
        } catch (...) {
            result_release(res);
            throw;
        }
        *out = res; *n_out = fs.n;
        return;
    }
    wait_slot(s, si);
    const int64_t total = gather_to_first(s, si);  // top-k mode: at most k records per shard cross the links
    fdcm_match* all = download(s, total);
    if (job.topk) {
        // every shard's list is ascending by score with ties in positional order, and the lists arrive in shard order:
        // a stable sort by score keeps ties in (shard, position) = global positional order
        std::stable_sort(all, all + total, [](const fdcm_match& a, const fdcm_match& b) {
            return fdcm::ordered_key_host(a.score) < fdcm::ordered_key_host(b.score);
        });
        *n_out = std::min<int64_t>(job.k, total);
    } else {
        *n_out = total;
    }
    *out = all;
}

Job make_job(const float* scene, int64_t n_scene, int64_t maxT, int64_t maxS, int optimizer, int64_t batch, bool topk, int penalty,
             float tau, int64_t k) {
    if (n_scene < 0 || (n_scene > 0 && !scene)) throw std::string("bad scene_lines");
    if (maxT < 0 || maxS < 0) throw std::string("negative search window");
    if (optimizer < FDCM_DEFAULT_OPTIMIZE || optimizer > FDCM_INDULGENT_OPTIMIZE) throw std::string("unknown optimizer");
    if (topk && (penalty < -1 || penalty > FDCM_EXPONENTIAL_PENALTY)) throw std::string("unknown penalty");
    if (topk && k < 0) throw std::string("k must be >= 0");
    Job j;
    j.scene.assign(scene, scene + 4 * n_scene);
    j.n_scene = n_scene; j.maxT = maxT; j.maxS = maxS; j.optimizer = optimizer; j.batch = batch;
    j.topk = topk; j.penalty = penalty; j.tau = tau; j.k = k;
    return j;
}

void destroy_sharded(fdcm_sharded* s) {
    if (!s) return;
    stop_workers(s);
    for (size_t i = 0; i < s->comms.size(); ++i)
        if (s->comms[i] && rccl().CommDestroy) (void)rccl().CommDestroy(s->comms[i]);
    for (auto& sh : s->shards) {
        (void)hipSetDevice(sh.device);
        if (sh.stream) { (void)hipStreamSynchronize(sh.stream); (void)hipStreamDestroy(sh.stream); }
        if (sh.tset) (void)fdcm_templates_free(sh.tset);
        if (sh.full) (void)fdcm_templates_free(sh.full);
    }
    if (!s->shards.empty()) (void)hipSetDevice(s->shards[0].device);
    s->gathered.release();
    delete s;
}

}  // namespace

extern "C" {

int fdcm_sharded_create(const int* devices, int n_devices, const float* tmpl_lines, const int64_t* offsets, int64_t n_templates,
                        int64_t depth, float dt3_coeff, float padding, int distance, int flags, fdcm_sharded** out) {
    fdcm_sharded* s = nullptr;
    DeviceGuard guard;
    int rc = guarded_s([&] {
        if (!out) throw std::string("out is null");
        *out = nullptr;
        if (n_devices < 1 || n_devices > 64) throw std::string("n_devices must be 1..64");
        if (n_templates < 0 || !offsets || (n_templates > 0 && offsets[n_templates] > 0 && !tmpl_lines)) throw std::string("bad templates");
        if (depth < 0 || distance < FDCM_L2 || distance > FDCM_L1) throw std::string("bad feature-map parameters");
        int have = 0;
        FDCM_HIP(hipGetDeviceCount(&have));
        s = new fdcm_sharded();
        s->depth = depth; s->coeff = dt3_coeff; s->padding = padding; s->distance = distance;
        s->always_collective = (flags & FDCM_SHARDED_ALWAYS_COLLECTIVE) != 0;
        s->same_device_ok = (flags & FDCM_SHARDED_ALLOW_SAME_DEVICE) != 0;
        if (s->same_device_ok && s->always_collective) throw std::string("ALLOW_SAME_DEVICE and ALWAYS_COLLECTIVE exclude each other");
        s->all_offsets.assign(offsets, offsets + n_templates + 1);
        if (s->all_offsets.back() > 0) s->all_lines.assign(tmpl_lines, tmpl_lines + 4 * s->all_offsets.back());
        s->shards.resize((size_t)n_devices);
        std::vector<int> devs((size_t)n_devices);
        for (int i = 0; i < n_devices; ++i) {
            devs[i] = devices ? devices[i] : i;
            if (devs[i] < 0 || devs[i] >= have) throw std::string("device ") + std::to_string(devs[i]) + " does not exist";
            for (int j = 0; j < i; ++j)
                if (devs[j] == devs[i] && !s->same_device_ok) throw std::string("a device is listed twice");
        }
        for (int i = 0; i < n_devices; ++i) {
            Shard& sh = s->shards[(size_t)i];
            sh.device = devs[i];
            // contiguous ranges: shard r gets [r T / n, (r + 1) T / n)  
            sh.begin = n_templates * i / n_devices;
            sh.end = n_templates * (i + 1) / n_devices;
            int r = fdcm_set_device(sh.device);
            if (r != FDCM_OK) throw std::string(fdcm_last_error());
            std::vector<int64_t> off((size_t)(sh.end - sh.begin) + 1);
            for (int64_t t = sh.begin; t <= sh.end; ++t) off[(size_t)(t - sh.begin)] = offsets[t] - offsets[sh.begin];
            r = fdcm_templates_create(tmpl_lines ? tmpl_lines + 4 * offsets[sh.begin] : nullptr, off.data(), sh.end - sh.begin, &sh.tset);
            if (r != FDCM_OK) throw std::string(fdcm_last_error());
            FDCM_HIP(hipStreamCreateWithFlags(&sh.stream, hipStreamNonBlocking));
        }
        bool all_on_root = true;
        for (int i = 1; i < n_devices; ++i) all_on_root = all_on_root && devs[i] == devs[0];
        if ((n_devices > 1 && !all_on_root) || s->always_collective) {
            if (s->same_device_ok) throw std::string("ALLOW_SAME_DEVICE: every shard must be on the first shard's device");
            Rccl& R = rccl();
            if (!R.error.empty()) throw R.error;
            s->comms.assign((size_t)n_devices, nullptr);
            nccl_check(R.CommInitAll(s->comms.data(), n_devices, devs.data()), "ncclCommInitAll");
        }
        start_workers(s, 1);
        *out = s;
    });
    if (rc != FDCM_OK) destroy_sharded(s);
    return rc;
}

int fdcm_sharded_free(fdcm_sharded* s) {
    DeviceGuard guard;
    return guarded_s([&] { destroy_sharded(s); });
}

int fdcm_sharded_set_frames_in_flight(fdcm_sharded* s, int n_frames) {
    DeviceGuard guard;
    return guarded_s([&] {
        if (!s) throw std::string("null handle");
        if (n_frames < 1 || n_frames > 16) throw std::string("n_frames must be 1..16");
        for (int64_t t : s->slot_ticket)
            if (t >= 0) throw std::string("frames are in flight: collect them with fdcm_sharded_wait first");
        if (n_frames != s->n_slots) start_workers(s, n_frames);
    });
}

int fdcm_sharded_set_mode(fdcm_sharded* s, int mode) {
    DeviceGuard guard;
    return guarded_s([&] {
        if (!s) throw std::string("null handle");
        if (mode != FDCM_SHARD_TEMPLATES && mode != FDCM_SHARD_FRAMES) throw std::string("unknown sharding mode");
        for (int64_t t : s->slot_ticket)
            if (t >= 0) throw std::string("frames are in flight: collect them with fdcm_sharded_wait first");
        if (mode == FDCM_SHARD_FRAMES)
            for (auto& sh : s->shards) {  // the whole list on every device, once
                if (sh.full) continue;
                if (fdcm_set_device(sh.device) != FDCM_OK) throw std::string(fdcm_last_error());
                if (fdcm_templates_create(s->all_lines.empty() ? nullptr : s->all_lines.data(), s->all_offsets.data(),
                                          (int64_t)s->all_offsets.size() - 1, &sh.full) != FDCM_OK)
                    throw std::string(fdcm_last_error());
            }
        if (mode != s->mode) {
            const int n = std::max(1, s->n_slots);
            s->mode = mode;
            s->next_ticket = 0;  // tickets restart: which device and slot a ticket runs on depends on the mode
            start_workers(s, n);
        }
    });
}

int fdcm_sharded_info(const fdcm_sharded* s, int* n_devices, int* devices, int64_t* shard_begin, int64_t* collectives,
                      int64_t* bytes_moved) {
    return guarded_s([&] {
        if (!s) throw std::string("null handle");
        const int n = (int)s->shards.size();
        if (n_devices) *n_devices = n;
        for (int i = 0; i < n; ++i) {
            if (devices) devices[i] = s->shards[(size_t)i].device;
            if (shard_begin) shard_begin[i] = s->shards[(size_t)i].begin;
        }
        if (shard_begin) shard_begin[n] = s->shards.back().end;
        if (collectives) *collectives = s->collectives;
        if (bytes_moved) *bytes_moved = s->bytes_moved;
    });
}

int fdcm_sharded_submit(fdcm_sharded* s, const float* scene_lines, int64_t n_scene_lines, int64_t max_tmpl_lines,
                        int64_t max_scene_lines, int optimizer, int64_t batch_size, int64_t* ticket) {
    return guarded_s([&] {
        if (!s || !ticket) throw std::string("null argument");
        *ticket = submit_frame(s, make_job(scene_lines, n_scene_lines, max_tmpl_lines, max_scene_lines, optimizer, batch_size, false,
                                           -1, 1.f, 0));
    });
}

int fdcm_sharded_submit_topk(fdcm_sharded* s, const float* scene_lines, int64_t n_scene_lines, int64_t max_tmpl_lines,
                             int64_t max_scene_lines, int optimizer, int64_t batch_size, int penalty, float tau, int64_t k,
                             int64_t* ticket) {
    return guarded_s([&] {
        if (!s || !ticket) throw std::string("null argument");
        *ticket = submit_frame(s, make_job(scene_lines, n_scene_lines, max_tmpl_lines, max_scene_lines, optimizer, batch_size, true,
                                           penalty, tau, k));
    });
}

int fdcm_sharded_wait(fdcm_sharded* s, int64_t ticket, fdcm_match** out, int64_t* n_out) {
    DeviceGuard guard;
    return guarded_s([&] {
        if (!s || !out || !n_out) throw std::string("null argument");
        collect_frame(s, ticket, out, n_out);
    });
}

int fdcm_sharded_search(fdcm_sharded* s, const float* scene_lines, int64_t n_scene_lines, int64_t max_tmpl_lines,
                        int64_t max_scene_lines, int optimizer, int64_t batch_size, fdcm_match** out, int64_t* n_out) {
    DeviceGuard guard;
    return guarded_s([&] {
        if (!s || !out || !n_out) throw std::string("null argument");
        *out = nullptr; *n_out = 0;
        const int64_t t = submit_frame(s, make_job(scene_lines, n_scene_lines, max_tmpl_lines, max_scene_lines, optimizer, batch_size,
                                                   false, -1, 1.f, 0));
        collect_frame(s, t, out, n_out);
    });
}

int fdcm_sharded_search_topk(fdcm_sharded* s, const float* scene_lines, int64_t n_scene_lines, int64_t max_tmpl_lines,
                             int64_t max_scene_lines, int optimizer, int64_t batch_size, int penalty, float tau, int64_t k,
                             fdcm_match** out, int64_t* n_out) {
    DeviceGuard guard;
    return guarded_s([&] {
        if (!s || !out || !n_out) throw std::string("null argument");
        *out = nullptr; *n_out = 0;
        const int64_t t = submit_frame(s, make_job(scene_lines, n_scene_lines, max_tmpl_lines, max_scene_lines, optimizer, batch_size,
                                                   true, penalty, tau, k));
        collect_frame(s, t, out, n_out);
    });
}

int fdcm_sharded_last_timing(const fdcm_sharded* s, int shard, fdcm_build_timing* bt, fdcm_search_timing* st) {
    return guarded_s([&] {
        if (!s || shard < 0 || shard >= (int)s->shards.size()) throw std::string("bad shard index");
        if (bt) *bt = s->shards[(size_t)shard].bt;
        if (st) *st = s->shards[(size_t)shard].st;
    });
}

}  // extern "C"
