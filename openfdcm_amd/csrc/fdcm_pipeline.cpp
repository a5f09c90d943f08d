// fdcm_pipeline.cpp -- frame pipeline over the blocking entry points of include/fdcm.h.
//
// The DT3 build (a sequential envelope per image row) and the search (dependent gathers per
// candidate) are latency bound at the reference's frame sizes: one frame leaves most of the 1024
// SIMDs waiting.  Frames of a stream are independent of each other, so a pipeline keeps several in
// flight: each slot owns a feature map (its own HBM volume, workspaces and HIP stream) and a host
// worker thread that runs  rebuild -> search  for the frames it is handed; kernels of different
// slots overlap on the device.  Results are the ones the blocking calls return (the workers call
// them), delivered per ticket, so a caller that waits in submission order sees frame order.
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fdcm_internal.h"

namespace {

struct Slot {
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    // job (guarded by mu)
    bool has_job = false, done = true, quit = false;
    int64_t ticket = -1;
    std::vector<float> scene;
    int64_t n_lines = 0;
    fdcm_match* out_device = nullptr;
    // result
    int rc = FDCM_OK;
    std::string error;
    fdcm_match* out_host = nullptr;
    int64_t n_out = 0;
    fdcm_build_timing bt{};
    fdcm_search_timing st{};
    fdcm_featuremap* fm = nullptr;
};

}  // namespace

struct fdcm_pipeline {
    int device = 0;
    int64_t depth = 0;
    float coeff = 0.f, padding = 0.f;
    int distance = 0;
    const fdcm_templates* templates = nullptr;
    int64_t maxT = 0, maxS = 0, batch = 1;
    int optimizer = 0;
    int32_t base = 0;
    std::vector<std::unique_ptr<Slot>> slots;
    int64_t next_ticket = 0;
};

namespace {

void run_frame(fdcm_pipeline* p, Slot& s) {
    s.out_host = nullptr;
    s.n_out = 0;
    int rc = s.fm ? fdcm_featuremap_rebuild(s.fm, s.scene.data(), s.n_lines)
                  : fdcm_featuremap_build(s.scene.data(), s.n_lines, p->depth, p->coeff, p->padding, p->distance, &s.fm);
    if (rc == FDCM_OK && s.fm) s.fm->shares_gpu = p->slots.size() > 1;  // (tunes the next builds of this slot)
    if (rc == FDCM_OK) {
        rc = s.out_device
                 ? fdcm_search_device(s.fm, p->templates, s.scene.data(), s.n_lines, p->maxT, p->maxS, p->optimizer,
                                      p->batch, p->base, s.out_device, &s.n_out)
                 : fdcm_search(s.fm, p->templates, s.scene.data(), s.n_lines, p->maxT, p->maxS, p->optimizer, p->batch,
                               p->base, &s.out_host, &s.n_out);
    }
    if (rc == FDCM_OK) {
        (void)fdcm_featuremap_last_timing(s.fm, &s.bt);
        (void)fdcm_search_last_timing(s.fm, &s.st);
    } else {
        s.error = fdcm_last_error();
    }
    s.rc = rc;
}

void worker_main(fdcm_pipeline* p, Slot* s) {
    (void)fdcm_set_device(p->device);
    std::unique_lock<std::mutex> lk(s->mu);
    while (true) {
        s->cv.wait(lk, [&] { return s->has_job || s->quit; });
        if (s->quit) break;
        s->has_job = false;
        lk.unlock();
        run_frame(p, *s);
        lk.lock();
        s->done = true;
        s->cv.notify_all();
    }
    lk.unlock();
    if (s->fm) (void)fdcm_featuremap_free(s->fm);
    s->fm = nullptr;
}

}  // namespace

extern "C" {

int fdcm_pipeline_create(int64_t depth, float dt3_coeff, float padding, int distance, const fdcm_templates* templates,
                         int64_t max_tmpl_lines, int64_t max_scene_lines, int optimizer, int64_t batch_size,
                         int32_t tmpl_index_base, int n_slots, fdcm_pipeline** out) {
    if (!out || !templates || n_slots < 1 || n_slots > 64 || depth < 0 || distance < FDCM_L2 || distance > FDCM_L1 ||
        max_tmpl_lines < 0 || max_scene_lines < 0 ||
        optimizer < FDCM_DEFAULT_OPTIMIZE || optimizer > FDCM_INDULGENT_OPTIMIZE ||
        (optimizer == FDCM_BATCH_OPTIMIZE && batch_size < 1)) {
        fdcm::set_error("fdcm_pipeline_create: bad argument");
        if (out) *out = nullptr;
        return FDCM_EINVAL;
    }
    auto* p = new fdcm_pipeline();
    p->device = templates->device;
    p->depth = depth; p->coeff = dt3_coeff; p->padding = padding; p->distance = distance;
    p->templates = templates;
    p->maxT = max_tmpl_lines; p->maxS = max_scene_lines; p->optimizer = optimizer; p->batch = batch_size;
    p->base = tmpl_index_base;
    for (int i = 0; i < n_slots; ++i) {
        p->slots.emplace_back(new Slot());
        Slot* s = p->slots.back().get();
        s->worker = std::thread(worker_main, p, s);
    }
    *out = p;
    return FDCM_OK;
}

int fdcm_pipeline_submit(fdcm_pipeline* p, const float* scene_lines, int64_t n_lines, fdcm_match* out_device,
                         int64_t* ticket) {
    if (!p || !ticket || n_lines < 0 || (n_lines > 0 && !scene_lines)) {
        fdcm::set_error("fdcm_pipeline_submit: bad argument");
        return FDCM_EINVAL;
    }
    Slot& s = *p->slots[(size_t)(p->next_ticket % (int64_t)p->slots.size())];
    std::unique_lock<std::mutex> lk(s.mu);
    if (s.ticket >= 0) {
        fdcm::set_error("fdcm_pipeline_submit: every slot holds a frame that has not been waited for");
        return FDCM_EINVAL;
    }
    s.scene.assign(scene_lines, scene_lines + 4 * n_lines);
    s.n_lines = n_lines;
    s.out_device = out_device;
    s.ticket = p->next_ticket;
    s.done = false;
    s.has_job = true;
    *ticket = p->next_ticket++;
    s.cv.notify_all();
    return FDCM_OK;
}

int fdcm_pipeline_wait(fdcm_pipeline* p, int64_t ticket, fdcm_match** out, int64_t* n_out, fdcm_build_timing* bt,
                       fdcm_search_timing* st) {
    if (!p || ticket < 0 || !n_out) {
        fdcm::set_error("fdcm_pipeline_wait: bad argument");
        return FDCM_EINVAL;
    }
    Slot& s = *p->slots[(size_t)(ticket % (int64_t)p->slots.size())];
    std::unique_lock<std::mutex> lk(s.mu);
    if (s.ticket != ticket) {
        fdcm::set_error("fdcm_pipeline_wait: unknown or already collected ticket");
        return FDCM_EINVAL;
    }
    s.cv.wait(lk, [&] { return s.done; });
    s.ticket = -1;
    const int rc = s.rc;
    if (rc != FDCM_OK) {
        fdcm::set_error(s.error);
        if (s.out_host) fdcm_matches_free(s.out_host);
        s.out_host = nullptr;
        if (out) *out = nullptr;
        *n_out = 0;
        return rc;
    }
    *n_out = s.n_out;
    if (out) *out = s.out_host; else if (s.out_host) fdcm_matches_free(s.out_host);
    s.out_host = nullptr;
    if (bt) *bt = s.bt;
    if (st) *st = s.st;
    return FDCM_OK;
}

int fdcm_pipeline_slots(const fdcm_pipeline* p, int* n_slots) {
    if (!p || !n_slots) {
        fdcm::set_error("fdcm_pipeline_slots: null argument");
        return FDCM_EINVAL;
    }
    *n_slots = (int)p->slots.size();
    return FDCM_OK;
}

int fdcm_pipeline_free(fdcm_pipeline* p) {
    if (!p) return FDCM_OK;
    for (auto& sp : p->slots) {
        Slot& s = *sp;
        {
            std::unique_lock<std::mutex> lk(s.mu);
            s.cv.wait(lk, [&] { return s.done; });  // a frame in flight finishes first
            s.quit = true;
            s.cv.notify_all();
        }
        if (s.worker.joinable()) s.worker.join();
        if (s.out_host) fdcm_matches_free(s.out_host);
    }
    delete p;
    return FDCM_OK;
}

}  // extern "C"
