// fdcm_host.cpp -- host side of the DT3 build: everything that involves libm (atanf, cosf, sinf),
// the scene bounding box, line classification and Cohen-Sutherland clipping.  O(scene lines)
// work; the O(volume) work is in fdcm_build.hip.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <unordered_map>

#include "fdcm_internal.h"

namespace fdcm {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
const char* last_error_cstr() { return g_last_error.c_str(); }

void DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return;
    release();
    FDCM_HIP(hipMalloc(&p, bytes));
    cap = bytes;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}
void PinnedBuf::reserve(size_t bytes) {
    if (bytes <= cap) return;
    release();
    FDCM_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    cap = bytes;
}
void PinnedBuf::release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
}

// ---- clipLines with the box [0, W-1] x [0, H-1] (drawing.cpp:29-112, deleteOob = true) ----
static inline int out_code(float x, float y, float xmax, float ymax) {
    int code = 0;
    if (x < 0.0f) code |= 1; else if (x > xmax) code |= 2;
    if (y < 0.0f) code |= 4; else if (y > ymax) code |= 8;
    return code;
}
static inline void clip_y(float* a, const float* b, float yc) {  // drawing.cpp:53-56
    a[0] = a[0] + (b[0] - a[0]) * (yc - a[1]) / (b[1] - a[1]);
    a[1] = yc;
}
static inline void clip_x(float* a, const float* b, float xc) {  // drawing.cpp:58-61
    a[1] = a[1] + (b[1] - a[1]) * (xc - a[0]) / (b[0] - a[0]);
    a[0] = xc;
}
static bool clip_line(float* l, float xmax, float ymax) {
    float* a = l;
    float* b = l + 2;
    int ca = out_code(a[0], a[1], xmax, ymax), cb = out_code(b[0], b[1], xmax, ymax);
    for (int guard = 0; guard < 1000; ++guard) {
        if (ca == 0 && cb == 0) return true;
        if (ca & cb) return false;
        if (ca != 0) {
            if (ca & 8) clip_y(a, b, ymax);
            else if (ca & 4) clip_y(a, b, 0.0f);
            else if (ca & 2) clip_x(a, b, xmax);
            else if (ca & 1) clip_x(a, b, 0.0f);
            ca = out_code(a[0], a[1], xmax, ymax);
            continue;
        }
        if (cb & 8) clip_y(b, a, ymax);
        else if (cb & 4) clip_y(b, a, 0.0f);
        else if (cb & 2) clip_x(b, a, xmax);
        else if (cb & 1) clip_x(b, a, 0.0f);
        cb = out_code(b[0], b[1], xmax, ymax);
    }
    return false;
}

// One LinSpaced axis of rasterizeLine (drawing.h:74-102).
static inline void lin_axis(int n, float lo, float hi, float& low, float& high, float& step, int32_t& mode) {
    if (n == 1) lo = hi;  // linspaced_op: low := high for a single step
    low = lo;
    high = hi;
    step = (n == 1) ? 0.0f : (hi - lo) / (float)(n - 1);
    mode = (std::fabs(hi) < std::fabs(lo)) ? 2 : 1;
}

// rasterizeLine (drawing.h:74-102) reduced to a descriptor; the device evaluates the points.
static RasterLine raster_descriptor(const float* l, int slice) {
    RasterLine r{};
    r.slice = slice;
    const float p1x = l[0], p1y = l[1], p2x = l[2], p2y = l[3];
    if (all_close2(p2x, p2y, p1x, p1y)) {
        r.n = 1;
        r.xmode = r.ymode = 0;
        r.xlow = p1x;
        r.ylow = p1y;
        return r;
    }
    const float vx = p2x - p1x, vy = p2y - p1y;
    float rx, ry;
    rasterize_vector(vx, vy, rx, ry);
    if (relatively_equal(rx, 0.0f)) {
        r.n = int(vy / ry) + 1;
        r.xmode = 0;
        r.xlow = p1x;
        lin_axis(r.n, p1y, p2y, r.ylow, r.yhigh, r.ystep, r.ymode);
    } else if (relatively_equal(ry, 0.0f)) {
        r.n = int(vx / rx) + 1;
        lin_axis(r.n, p1x, p2x, r.xlow, r.xhigh, r.xstep, r.xmode);
        r.ymode = 0;
        r.ylow = p1y;
    } else {
        r.n = static_cast<int>(std::max(vx / rx, vy / ry)) + 1;
        lin_axis(r.n, p1x, p2x, r.xlow, r.xhigh, r.xstep, r.xmode);
        lin_axis(r.n, p1y, p2y, r.ylow, r.yhigh, r.ystep, r.ymode);
    }
    if (r.n < 0) r.n = 0;
    return r;
}

// Proxy of the L2 sweep's time per (slice, 64-row chunk), for the launch order of a build without history (only such a
// build asks for it: run_build): a row's chain is as long as the slice has seeded columns, and the columns whose seeds lie
// outside the chunk's 64 rows (rows far from the line: long runs without an envelope vertex) count double
// (tools/k2_cost_model.py: r = 0.85).
void sweep_cost_proxy(const BuildPlan& plan, std::vector<int32_t>& cost) {
    const int HW64 = (int)((plan.H + 63) / 64);
    cost.assign((size_t)plan.m * HW64, 0);
    for (const LineBox& b : plan.boxes) {
        const int ncols = (int)(b.xhi - b.xlo) + 1;
        for (int c = 0; c < HW64; ++c) {
            const float c0 = (float)(c * 64), c1 = (float)(c * 64 + 63);
            const float ov = std::min(b.yhi, c1) - std::max(b.ylo, c0) + 1.f;  // rows of the line's y range inside the chunk
            const int inside = ov <= 0.f ? 0 : std::min(ncols, (int)(ncols * ov / (b.yhi - b.ylo + 1.f)) + 1);
            cost[(size_t)b.slice * HW64 + c] += 2 * ncols - inside;
        }
    }
}

void make_build_plan(const float* lines, int64_t n, int64_t depth, float coeff, float padding, BuildPlan& plan) {
    plan = BuildPlan{};
    if (n == 0) return;
    // getSceneCenteredTranslation, dt3cpu.cpp:109-116 (+ minmaxPoint, math.h:166-171)
    float mnx = lines[0], mny = lines[1], mxx = mnx, mxy = mny;
    for (int64_t i = 0; i < 2 * n; ++i) {
        const float x = lines[2 * i], y = lines[2 * i + 1];
        mnx = std::min(mnx, x); mxx = std::max(mxx, x);
        mny = std::min(mny, y); mxy = std::max(mxy, y);
    }
    const float dx = mxx - mnx, dy = mxy - mny;
    const float corrected_ratio = std::max(1.f, padding);
    const float req = (corrected_ratio * std::max(dx, dy)) * 1.f;
    plan.tx = req / 2.f - (mxx + mnx) / 2.f;
    plan.ty = req / 2.f - (mxy + mny) / 2.f;
    plan.W = plan.H = (int64_t)(size_t)std::ceil(req + 1.f);
    // angle keys, dt3cpu.h:188-190 (std::set<float>)
    for (int64_t i = 0; i < depth; ++i) plan.keys.push_back(float(i) * kPif / float(depth) - kPi2f);
    std::sort(plan.keys.begin(), plan.keys.end());
    plan.keys.erase(std::unique(plan.keys.begin(), plan.keys.end()), plan.keys.end());
    plan.m = (int64_t)plan.keys.size();
    const int m = (int)plan.m;
    // classifyLines (dt3cpu.h:123-134) + clipLines + rasterizeLine per class.  Lines are visited
    // slice by slice in index order like the reference; the seed set is order independent.
    const float xmax = (float)(plan.W - 1), ymax = (float)(plan.H - 1);
    std::vector<std::vector<RasterLine>> per_slice((size_t)m);
    for (int64_t i = 0; i < n; ++i) {
        float l[4] = {lines[4 * i] + plan.tx, lines[4 * i + 1] + plan.ty, lines[4 * i + 2] + plan.tx,
                      lines[4 * i + 3] + plan.ty};  // translate, math.h:352-354
        const float angle = std::atan((l[3] - l[1]) / (l[2] - l[0]));  // getAngle, math.h:295-299
        const int k = closest_orientation(plan.keys.data(), m, angle);
        if (!clip_line(l, xmax, ymax)) continue;
        per_slice[k].push_back(raster_descriptor(l, k));
        plan.boxes.push_back(LineBox{k, std::min(l[0], l[2]), std::max(l[0], l[2]), std::min(l[1], l[3]), std::max(l[1], l[3])});
    }
    plan.slice_first.assign((size_t)m + 1, 0);
    for (int k = 0; k < m; ++k) {
        plan.slice_first[k] = (int32_t)plan.raster.size();
        plan.raster.insert(plan.raster.end(), per_slice[k].begin(), per_slice[k].end());
    }
    plan.slice_first[m] = (int32_t)plan.raster.size();
    // propagateOrientation step table, dt3cpu.cpp:86-106
    {
        const int fwd = static_cast<int>(std::ceil(1.5 * m));
        const int bwd = -static_cast<int>(std::floor(1.5 * m));
        auto emit = [&](int start, int end, int step) {
            for (int c = start; c != end; c += step) {
                const int c1 = (m + ((c - step) % m)) % m;
                const int c2 = (m + (c % m)) % m;
                const float h = std::abs(plan.keys[c1] - plan.keys[c2]);
                const float min_h = std::min(h, std::abs(h - kPif));
                plan.prop.push_back(PropStep{c1, c2, coeff * min_h, 0});
            }
        };
        emit(0, fwd, 1);
        emit(m, bwd, -1);
    }
    // lineIntegral direction per slice, imgproc.h:42-50,66
    for (int k = 0; k < m; ++k) {
        float rx, ry;
        rasterize_vector(std::cos(plan.keys[k]), std::sin(plan.keys[k]), rx, ry);
        IntegralDesc d{0, 0, 0.f, 0};
        if (std::abs(rx) == 1) { d.mode = 1; d.s = (int)(long)rx; d.r = ry; }
        else if (std::abs(ry) == 1) { d.mode = 2; d.s = (int)(long)ry; d.r = rx; }
        plan.integral.push_back(d);
    }
}


// ---------------------------------------------------------------- result buffers
// Match arrays handed to the caller (fdcm_search, fdcm_pipeline_wait) are pinned host buffers from
// a small process-wide pool: the device-to-host copy lands in the buffer the caller receives (no
// second copy), and a released buffer is reused instead of being unmapped (a fresh 1 MB malloc
// costs ~250 page faults on first touch).  fdcm_matches_free returns a buffer to the pool.
namespace {
struct ResultPool {
    std::mutex mu;
    std::vector<std::pair<void*, size_t>> free_list;   // idle buffers
    std::unordered_map<void*, size_t> capacity;        // every live buffer
    static constexpr size_t kMaxIdle = 16;
};
ResultPool& result_pool() {
    static ResultPool* p = new ResultPool();  // never destroyed: buffers may outlive static destructors
    return *p;
}
}  // namespace

fdcm_match* result_acquire(size_t bytes) {
    bytes = std::max<size_t>(bytes, 64);
    ResultPool& P = result_pool();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        size_t best = P.free_list.size();
        for (size_t i = 0; i < P.free_list.size(); ++i)
            if (P.free_list[i].second >= bytes && (best == P.free_list.size() || P.free_list[i].second < P.free_list[best].second))
                best = i;
        if (best != P.free_list.size()) {
            void* p = P.free_list[best].first;
            P.free_list.erase(P.free_list.begin() + (long)best);
            return (fdcm_match*)p;
        }
    }
    const size_t cap = (bytes + (1u << 16) - 1) & ~(size_t)((1u << 16) - 1);
    void* p = nullptr;
    FDCM_HIP(hipHostMalloc(&p, cap, hipHostMallocPortable));
    std::lock_guard<std::mutex> lk(P.mu);
    P.capacity[p] = cap;
    return (fdcm_match*)p;
}

void result_release(fdcm_match* m) {
    if (!m) return;
    ResultPool& P = result_pool();
    void* victim = nullptr;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.capacity.find((void*)m);
        if (it == P.capacity.end()) return;  // not ours: ignore rather than corrupt the heap
        for (const auto& f : P.free_list)
            if (f.first == (void*)m) return;  // released twice: ignore
        if (P.free_list.size() < ResultPool::kMaxIdle) {
            P.free_list.emplace_back((void*)m, it->second);
        } else {
            victim = (void*)m;
            P.capacity.erase(it);
        }
    }
    if (victim) (void)hipHostFree(victim);
}

}  // namespace fdcm
