// fdcm_oracle.cpp -- CPU restatement of the OpenFDCM hot path.  TEST INFRASTRUCTURE ONLY.
//
// This file is the parity oracle and the `cpu_baseline` ("port") of bench.py.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
// (openfdcm_amd/, libfdcm_hip.so) never links, imports or calls anything in oracle/.
//
// Parity status: the reference (Innoptech/OpenFDCM v0.10.0) cannot be compiled in this image
// (Eigen 3.4.0, BS::thread_pool 4.1.0, packio 0.2.1 are network FetchContent dependencies that
// are absent), so this restatement is pinned against the reference's own known-answer tests
// (tests/test_oracle_kat.py, transcribed from tests/core/src/*.test.cpp and
// tests/matching/src/**/*.test.cpp) and against an independent numpy restatement
// (oracle/pyoracle.py) on random inputs.  Large-image float behaviour, Eigen's sum() order and
// std::sort tie-breaking are pinned only by agreement of the two restatements.
//
// Every function cites the reference file:line it follows (paths relative to /root/reference).
// Eigen behaviours restated from Eigen 3.4.0 (x86-64 default build: SSE2, no FMA):
//   LinSpaced  -> Eigen/src/Core/functors/NullaryFunctors.h linspaced_op(_impl)
//   sum()      -> Eigen/src/Core/Redux.h redux_impl<LinearVectorizedTraversal, NoUnrolling> + SSE predux
//   colwise().normalized()/norm() -> Eigen/src/Core/VectorwiseOp.h
//
// Build: g++ -O3 -fno-math-errno -ffp-contract=off -std=c++17 -shared -fPIC (oracle/Makefile).
#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <numeric>
#include <optional>
#include <thread>
#include <vector>

namespace fdcmo {

static const float kPif = 3.14159265358979323846f;    // M_PIf   (math.h:38-40)
static const float kPi2f = 1.57079632679489661923f;   // M_PI_2f (math.h:44-46)

// ------------------------------------------------------------------------------------------
// Containers.  LineArray = 4 x N float, column-major (math.h:66): line i = l[4i..4i+3] = x1,y1,x2,y2.
// RawImage = rows x cols float, column-major (math.h:57): (y,x) at x*rows + y.
// ------------------------------------------------------------------------------------------
struct Lines {
    std::vector<float> d;
    long n() const { return (long)(d.size() / 4); }
    const float* line(long i) const { return &d[4 * i]; }
    float* line(long i) { return &d[4 * i]; }
    void push(float x1, float y1, float x2, float y2) { d.insert(d.end(), {x1, y1, x2, y2}); }
};

struct Image {
    long rows = 0, cols = 0;
    std::vector<float> d;
    Image() = default;
    Image(long r, long c, float v) : rows(r), cols(c), d((size_t)r * c, v) {}
    float& at(long y, long x) { return d[(size_t)x * rows + y]; }
    float at(long y, long x) const { return d[(size_t)x * rows + y]; }
};

struct Point2 { float x, y; };
struct Mat23 { float m[6]; };  // row-major: r00 r01 tx ; r10 r11 ty

// ------------------------------------------------------------------------------------------
// math.h
// ------------------------------------------------------------------------------------------
// relativelyEqual<float,float>, math.h:182-188.  rtol/atol are double; fabs/max act on float.
static inline bool relativelyEqual(float a, float b, double rtol = 1e-10,
                                   double atol = std::numeric_limits<float>::epsilon()) {
    return std::fabs(a - b) <= atol + rtol * std::max(std::fabs(a), std::fabs(b));
}

// allClose on float 2-vectors, math.h:202-208 (rtol = 0.f, atol = 1e-5f, all float arithmetic).
static inline bool allClose2(Point2 a, Point2 b, float rtol = 0.0f, float atol = 1e-5f) {
    bool c0 = std::fabs(a.x - b.x) <= (atol + rtol * std::fabs(b.x));
    bool c1 = std::fabs(a.y - b.y) <= (atol + rtol * std::fabs(b.y));
    return c0 && c1;
}

// minmaxPoint, math.h:166-171.
static inline void minmaxPoint(const Lines& l, Point2& mn, Point2& mx) {
    mn = {l.d[0], l.d[1]};
    mx = mn;
    for (long i = 0; i < 2 * l.n(); ++i) {
        float x = l.d[2 * i], y = l.d[2 * i + 1];
        mn.x = std::min(mn.x, x); mx.x = std::max(mx.x, x);
        mn.y = std::min(mn.y, y); mx.y = std::max(mx.y, y);
    }
}

// getAngle, math.h:295-299: atanf(dy/dx).
static inline float getAngle(const float* l) { return std::atan((l[3] - l[1]) / (l[2] - l[0])); }

// getLength, math.h:306-308: colwise().norm() = sqrt(dx*dx + dy*dy).
static inline float getLength(const float* l) {
    float dx = l[2] - l[0], dy = l[3] - l[1];
    return std::sqrt(dx * dx + dy * dy);
}

// normalize, math.h:331-333: colwise().normalized() = v / v.norm() with no zero check
// (VectorwiseOp.h normalized() = cwiseQuotient by the replicated norm()).
static inline Point2 normalize(const float* l) {
    float dx = l[2] - l[0], dy = l[3] - l[1];
    float n = std::sqrt(dx * dx + dy * dy);
    return {dx / n, dy / n};
}

// transform, math.h:341-344: R * p (coefficient = r0*x + r1*y, separate multiply/add) + t.
static inline Lines transform(const Lines& in, const Mat23& T) {
    Lines out;
    out.d.resize(in.d.size());
    for (long i = 0; i < 2 * in.n(); ++i) {
        float x = in.d[2 * i], y = in.d[2 * i + 1];
        out.d[2 * i] = (T.m[0] * x + T.m[1] * y) + T.m[2];
        out.d[2 * i + 1] = (T.m[3] * x + T.m[4] * y) + T.m[5];
    }
    return out;
}

// translate, math.h:352-354.
static inline Lines translate(const Lines& in, Point2 t) {
    Lines out;
    out.d.resize(in.d.size());
    for (long i = 0; i < 2 * in.n(); ++i) {
        out.d[2 * i] = in.d[2 * i] + t.x;
        out.d[2 * i + 1] = in.d[2 * i + 1] + t.y;
    }
    return out;
}

// align, math.h:387-406.
static inline std::array<Mat23, 2> align(const float* tl, const float* rl) {
    Point2 tmpl_d = normalize(tl), align_d = normalize(rl);
    const float cos = align_d.x * tmpl_d.x + align_d.y * tmpl_d.y;
    const float sin = align_d.y * tmpl_d.x - align_d.x * tmpl_d.y;
    // getCenter(ref_line) = (p2 + p1)/2, math.h:286-288
    const float rcx = (rl[2] + rl[0]) / 2, rcy = (rl[3] + rl[1]) / 2;
    auto centre_of_rotated = [&](float r00, float r01, float r10, float r11, float& cx, float& cy) {
        // rotate(line, rot) math.h:362-364 then getCenter
        float x1 = r00 * tl[0] + r01 * tl[1], y1 = r10 * tl[0] + r11 * tl[1];
        float x2 = r00 * tl[2] + r01 * tl[3], y2 = r10 * tl[2] + r11 * tl[3];
        cx = (x2 + x1) / 2;
        cy = (y2 + y1) / 2;
    };
    float c1x, c1y, c2x, c2y;
    centre_of_rotated(cos, -sin, sin, cos, c1x, c1y);
    centre_of_rotated(-cos, sin, -sin, -cos, c2x, c2y);
    Mat23 t1{{cos, -sin, rcx - c1x, sin, cos, rcy - c1y}};
    Mat23 t2{{-cos, sin, rcx - c2x, -sin, -cos, rcy - c2y}};
    return {t1, t2};
}

// combine(translation, transform), math.h:427-432.
static inline Mat23 combine(Point2 t, const Mat23& T) {
    Mat23 r = T;
    r.m[2] = T.m[2] + t.x;
    r.m[5] = T.m[5] + t.y;
    return r;
}

// ------------------------------------------------------------------------------------------
// drawing.h / drawing.cpp
// ------------------------------------------------------------------------------------------
// rasterizeVector, drawing.h:57-67.
static inline Point2 rasterizeVector(Point2 v) {
    const float tan_angle = v.y / v.x;
    if (tan_angle >= -1.0 && tan_angle < 1) {
        bool cond1{v.x < 0};
        return {(float)(1 - 2 * cond1), (float)(tan_angle - 2.0 * cond1 * tan_angle)};
    }
    bool cond2{v.y < 0};
    return {(float)(1 / tan_angle - 2.0 * cond2 * (1 / tan_angle)), (float)(1 - 2 * cond2)};
}

// Eigen 3.4.0 LinSpaced<float>(n, low, high), scalar path (see header comment).
static inline void linSpaced(long n, float low, float high, std::vector<float>& out) {
    out.resize((size_t)n);
    if (n == 1) low = high;  // linspaced_op ctor: impl((num_steps==1 ? high : low), high, num_steps)
    const long size1 = (n == 1) ? 1 : n - 1;
    const float step = (n == 1) ? 0.0f : (high - low) / (float)(n - 1);
    const bool flip = std::fabs(high) < std::fabs(low);
    for (long i = 0; i < n; ++i) {
        if (flip) out[i] = (i == 0) ? low : (high - (float)(size1 - i) * step);
        else      out[i] = (i == size1) ? high : (low + (float)i * step);
    }
}

// rasterizeLine, drawing.h:74-102.
static inline void rasterizeLine(const float* l, std::vector<long>& xs, std::vector<long>& ys) {
    xs.clear(); ys.clear();
    Point2 p1{l[0], l[1]}, p2{l[2], l[3]};
    if (allClose2(p2, p1)) {
        xs.push_back((long)std::round(p1.x));
        ys.push_back((long)std::round(p1.y));
        return;
    }
    Point2 line_vec{p2.x - p1.x, p2.y - p1.y};
    Point2 rastvec = rasterizeVector(line_vec);
    std::vector<float> fx, fy;
    if (relativelyEqual(rastvec.x, 0.0f)) {
        int const size = int(line_vec.y / rastvec.y) + 1;
        fx.assign((size_t)std::max(size, 0), p1.x);
        linSpaced(size, p1.y, p2.y, fy);
    } else if (relativelyEqual(rastvec.y, 0.0f)) {
        int const size = int(line_vec.x / rastvec.x) + 1;
        linSpaced(size, p1.x, p2.x, fx);
        fy.assign((size_t)std::max(size, 0), p1.y);
    } else {
        int size = static_cast<int>(std::max(line_vec.x / rastvec.x, line_vec.y / rastvec.y)) + 1;
        linSpaced(size, p1.x, p2.x, fx);
        linSpaced(size, p1.y, p2.y, fy);
    }
    for (size_t i = 0; i < fx.size(); ++i) {
        xs.push_back((long)std::round(fx[i]));
        ys.push_back((long)std::round(fy[i]));
    }
}

// clipLines, drawing.cpp:29-112 (Cohen-Sutherland; deleteOob = true).
struct Box { float xmin, xmax, ymin, ymax; };
static inline int computeOutCode(float x, float y, const Box& b) {
    int code = 0;
    if (x < b.xmin) code |= 1; else if (x > b.xmax) code |= 2;
    if (y < b.ymin) code |= 4; else if (y > b.ymax) code |= 8;
    return code;
}
static inline void clipAgainstY(float* p1, const float* p2, float y_crop) {  // drawing.cpp:53-56
    p1[0] = p1[0] + (p2[0] - p1[0]) * (y_crop - p1[1]) / (p2[1] - p1[1]);
    p1[1] = y_crop;
}
static inline void clipAgainstX(float* p1, const float* p2, float x_crop) {  // drawing.cpp:58-61
    p1[1] = p1[1] + (p2[1] - p1[1]) * (x_crop - p1[0]) / (p2[0] - p1[0]);
    p1[0] = x_crop;
}
static inline Lines clipLines(const Lines& lines, const Box& box) {
    Lines out;
    for (long i = 0; i < lines.n(); ++i) {
        float l[4];
        std::memcpy(l, lines.line(i), sizeof l);
        float* p1 = l; float* p2 = l + 2;
        int code1 = computeOutCode(p1[0], p1[1], box), code2 = computeOutCode(p2[0], p2[1], box);
        bool keep = false;
        for (int guard = 0; guard < 1000; ++guard) {  // the reference loops without a bound
            if (code1 == 0 && code2 == 0) { keep = true; break; }
            if (code1 & code2) break;
            if (code1 != 0) {
                if (code1 & 8) clipAgainstY(p1, p2, box.ymax);
                else if (code1 & 4) clipAgainstY(p1, p2, box.ymin);
                else if (code1 & 2) clipAgainstX(p1, p2, box.xmax);
                else if (code1 & 1) clipAgainstX(p1, p2, box.xmin);
                code1 = computeOutCode(p1[0], p1[1], box);
                continue;
            }
            if (code2 & 8) clipAgainstY(p2, p1, box.ymax);
            else if (code2 & 4) clipAgainstY(p2, p1, box.ymin);
            else if (code2 & 2) clipAgainstX(p2, p1, box.xmax);
            else if (code2 & 1) clipAgainstX(p2, p1, box.xmin);
            code2 = computeOutCode(p2[0], p2[1], box);
        }
        if (keep) out.push(l[0], l[1], l[2], l[3]);
    }
    return out;
}

// drawLines, drawing.h:111-125.
static inline void drawLines(Image& img, const Lines& lines, float color) {
    if (lines.n() == 0) return;
    Lines clipped = clipLines(lines, Box{0, (float)(img.cols - 1), 0, (float)(img.rows - 1)});
    std::vector<long> xs, ys;
    for (long i = 0; i < clipped.n(); ++i) {
        rasterizeLine(clipped.line(i), xs, ys);
        for (size_t k = 0; k < xs.size(); ++k) img.at(ys[k], xs[k]) = color;
    }
}

// ------------------------------------------------------------------------------------------
// imgproc.h
// ------------------------------------------------------------------------------------------
static inline Image transposed(const Image& a) {
    Image t(a.cols, a.rows, 0.f);
    const long B = 32;
    for (long x0 = 0; x0 < a.cols; x0 += B)
        for (long y0 = 0; y0 < a.rows; y0 += B)
            for (long x = x0; x < std::min(x0 + B, a.cols); ++x)
                for (long y = y0; y < std::min(y0 + B, a.rows); ++y)
                    t.d[(size_t)y * t.rows + x] = a.d[(size_t)x * a.rows + y];
    return t;
}

// _distanceTransformColumnPassL2, imgproc.h:91-130.  In place, including the re-read of
// already overwritten cells at :126-127.
static inline void columnPassL2(Image& img) {
    const long R = img.rows;
    std::vector<long> square_idx((size_t)R);
    for (long i = 0; i < R; ++i) square_idx[i] = i * i;
    std::vector<long> v((size_t)R);
    std::vector<float> z((size_t)R + 1);
    for (long i = 0; i < img.cols; ++i) {
        float* f = &img.d[(size_t)i * R];
        long k = 0;
        v[0] = 0;
        z[0] = -std::numeric_limits<float>::infinity();
        z[1] = std::numeric_limits<float>::infinity();
        for (long q = 1; q < R; ++q) {
            while (true) {
                long const v_k = v[k];
                float const s = (f[q] + square_idx[q] - f[v_k] - square_idx[v_k]) / (2 * q - 2 * v_k);
                if (s > z[k]) {
                    ++k;
                    v[k] = q;
                    z[k] = s;
                    z[k + 1] = std::numeric_limits<float>::infinity();
                    break;
                }
                --k;
            }
        }
        k = 0;
        for (long q = 0; q < R; ++q) {
            while (z[k + 1] < (float)q) ++k;
            long const v_k = v[k];
            long const q_ = q - v_k;
            f[q] = f[v_k] + square_idx[std::abs(q_)];
        }
    }
}

// _distanceTransformColumnPassL1, imgproc.h:137-146.
static inline void columnPassL1(Image& img) {
    const long R = img.rows;
    for (long q = 1; q < img.cols; ++q) {
        float* c = &img.d[(size_t)q * R];
        const float* p = &img.d[(size_t)(q - 1) * R];
        for (long y = 0; y < R; ++y) c[y] = std::min(c[y], p[y] + 1);
    }
    for (long q = img.cols - 2; q >= 0; --q) {
        float* c = &img.d[(size_t)q * R];
        const float* p = &img.d[(size_t)(q + 1) * R];
        for (long y = 0; y < R; ++y) c[y] = std::min(c[y], p[y] + 1);
    }
}

enum Distance { L2 = 0, L2_SQUARED = 1, L1 = 2 };  // imgproc.h:148

// distanceTransform<float, D>, imgproc.h:169-194.  size = (x = W, y = H).
static inline Image distanceTransform(const Lines& lines, long W, long H, int D) {
    Image img(H, W, std::numeric_limits<float>::max());
    drawLines(img, lines, 0.f);
    if (D == L1) {
        columnPassL1(img);
        img = transposed(img);
        columnPassL1(img);
        return transposed(img);
    }
    columnPassL2(img);
    img = transposed(img);
    columnPassL2(img);
    img = transposed(img);
    if (D == L2)
        for (float& v : img.d) v = std::sqrt(v);
    return img;
}

// lineIntegral, imgproc.h:38-84.
static inline void lineIntegral(Image& img, float lineAngle) {
    Point2 rastvec = rasterizeVector(Point2{std::cos(lineAngle), std::sin(lineAngle)});
    long p0x = 0, p0y = 0;
    if (rastvec.x < 0) p0x += img.cols - 1;
    if (rastvec.y < 0) p0y += img.rows - 1;
    const long R = img.rows, C = img.cols;
    if (std::abs(rastvec.x) == 1) {
        long previous_p1x = p0x;
        for (long i = 1; i < C; ++i) {
            long p1x = p0x + i * (long)rastvec.x;
            long p1y = static_cast<long>(std::round(i * rastvec.y)) - static_cast<long>(std::round((i - 1) * rastvec.y));
            long y1 = std::max(p1y, 0L), y2 = std::max(-p1y, 0L);
            long col_len = R - std::abs(p1y);
            float* dst = &img.d[(size_t)p1x * R + y1];
            const float* src = &img.d[(size_t)previous_p1x * R + y2];
            for (long j = 0; j < col_len; ++j) dst[j] += src[j];
            previous_p1x = p1x;
        }
    } else if (std::abs(rastvec.y) == 1) {
        Image rm = transposed(img);  // row-major copy (imgproc.h:68): (y,x) at y*C + x
        long previous_p1y = p0y;
        for (long i = 1; i < R; ++i) {
            long p1x = static_cast<long>(std::round(i * rastvec.x)) - static_cast<long>(std::round((i - 1) * rastvec.x));
            long p1y = p0y + i * (long)rastvec.y;
            long x1 = std::max(p1x, 0L), x2 = std::max(-p1x, 0L);
            long row_len = C - std::abs(p1x);
            float* dst = &rm.d[(size_t)p1y * C + x1];
            const float* src = &rm.d[(size_t)previous_p1y * C + x2];
            for (long j = 0; j < row_len; ++j) dst[j] += src[j];
            previous_p1y = p1y;
        }
        img = transposed(rm);
    }
}

// ------------------------------------------------------------------------------------------
// dt3cpu.h / dt3cpu.cpp
// ------------------------------------------------------------------------------------------
// closestOrientation, dt3cpu.h:93-114.  `keys` is the sorted key list of the std::map.
static inline long closestOrientation(const std::vector<float>& keys, const float* line) {
    const float line_angle = getAngle(line);
    long itlow = std::upper_bound(keys.begin(), keys.end(), line_angle) - keys.begin();
    const long end = (long)keys.size();
    if (itlow != end && itlow != 0) {
        const float upper_bound_diff = std::abs(line_angle - keys[itlow]);
        itlow--;
        const float lower_bound_diff = std::abs(line_angle - keys[itlow]);
        if (lower_bound_diff < upper_bound_diff) return itlow;
        itlow++;
        return itlow;
    }
    itlow = end - 1;
    const float angle1 = line_angle - keys[0];
    const float angle2 = line_angle - keys[itlow];
    if (std::min(angle1, std::abs(angle1 - kPif)) < std::min(angle2, std::abs(angle2 - kPif))) return 0;
    return itlow;
}

// getSceneCenteredTranslation, dt3cpu.cpp:109-116.
static inline void getSceneCenteredTranslation(const Lines& scene, float scene_padding, Point2& t, long& W, long& H) {
    Point2 mn, mx;
    minmaxPoint(scene, mn, mx);
    Point2 d{mx.x - mn.x, mx.y - mn.y};
    float const corrected_ratio = std::max(1.f, scene_padding);
    float const rm = corrected_ratio * std::max(d.x, d.y);
    Point2 required_max{rm * 1.f, rm * 1.f};
    t = {required_max.x / 2.f - (mx.x + mn.x) / 2.f, required_max.y / 2.f - (mx.y + mn.y) / 2.f};
    W = (long)(size_t)std::ceil(required_max.x + 1.f);
    H = (long)(size_t)std::ceil(required_max.y + 1.f);
}

// propagateOrientation, dt3cpu.cpp:77-107.
static inline void propagateOrientation(std::vector<float>& keys, std::vector<Image>& maps, float coeff) {
    const int m = (int)maps.size();
    const int fwd = static_cast<int>(std::ceil(1.5 * m));
    const int bwd = -static_cast<int>(std::floor(1.5 * m));
    auto propagate = [&](int start, int end, int step) {
        for (int c = start; c != end; c += step) {
            int c1 = (m + ((c - step) % m)) % m;
            int c2 = (m + (c % m)) % m;
            const float h = std::abs(keys[c1] - keys[c2]);
            const float min_h = std::min(h, std::abs(h - kPif));
            const float w = coeff * min_h;
            float* a = maps[c2].d.data();
            const float* b = maps[c1].d.data();
            const size_t n = maps[c2].d.size();
            for (size_t i = 0; i < n; ++i) a[i] = std::min(a[i], b[i] + w);
        }
    };
    propagate(0, fwd, 1);
    propagate(m, bwd, -1);
}

// minmaxTranslation, dt3cpu.cpp:30-75.
static inline float maxPropNaN(const float* a, int n) {
    float r = a[0];
    for (int i = 0; i < n; ++i) { if (std::isnan(a[i])) return a[i]; r = std::max(r, a[i]); }
    return r;
}
static inline float minPropNaN(const float* a, int n) {
    float r = a[0];
    for (int i = 0; i < n; ++i) { if (std::isnan(a[i])) return a[i]; r = std::min(r, a[i]); }
    return r;
}
static inline std::array<float, 2> minmaxTranslation(const Lines& tmpl, Point2 align_vec, long W, long H, Point2 extra) {
    const float inf = std::numeric_limits<float>::infinity();
    if (allClose2(align_vec, Point2{0, 0})) return {inf, inf};
    const float size[2] = {(float)W, (float)H};
    Point2 mn{0, 0}, mx{0, 0};
    if (tmpl.n() > 0) minmaxPoint(tmpl, mn, mx);
    const float minp[2] = {mn.x + extra.x, mn.y + extra.y};
    const float maxp[2] = {mx.x + extra.x, mx.y + extra.y};
    for (int r = 0; r < 2; ++r) if ((size[r] - 1 - maxp[r]) < 0) return {NAN, NAN};
    for (int r = 0; r < 2; ++r) if (minp[r] < 0) return {NAN, NAN};
    const float av[2] = {align_vec.x, align_vec.y};
    float mult[2][4], pos[2][4], neg[2][4];
    for (int r = 0; r < 2; ++r) {
        mult[r][0] = -maxp[r];
        mult[r][1] = -minp[r];
        mult[r][2] = (size[r] - maxp[r] - 1.f);
        mult[r][3] = (size[r] - minp[r] - 1.f);
        for (int c = 0; c < 4; ++c) {
            mult[r][c] /= av[r];
            bool sgn = std::signbit(mult[r][c]);
            pos[r][c] = sgn ? inf : mult[r][c];
            neg[r][c] = sgn ? mult[r][c] : -inf;
        }
    }
    // extremum_coeffs = {{neg_x, neg_y}, {pos_x, pos_y}}
    const float e00 = maxPropNaN(neg[0], 4), e01 = maxPropNaN(neg[1], 4);
    const float e10 = minPropNaN(pos[0], 4), e11 = minPropNaN(pos[1], 4);
    auto fin = [](float v) { return std::isfinite(v); };
    if (fin(e00) && fin(e01) && fin(e10) && fin(e11)) return {std::max(e00, e01), std::min(e10, e11)};
    if (fin(e00) && fin(e10)) return {e00, e10};
    return {e01, e11};
}

// Eigen 3.4.0 VectorXf::sum() (SSE2 Packet4f, 16-byte aligned data): see header comment.
static inline float eigenSum(const float* v, long size) {
    if (size == 0) return 0.f;
    const long ps = 4;
    const long alignedSize2 = (size / (2 * ps)) * (2 * ps);
    const long alignedSize = (size / ps) * ps;
    float res;
    if (alignedSize) {
        float p0[4] = {v[0], v[1], v[2], v[3]};
        if (alignedSize > ps) {
            float p1[4] = {v[4], v[5], v[6], v[7]};
            for (long idx = 2 * ps; idx < alignedSize2; idx += 2 * ps)
                for (int l = 0; l < 4; ++l) { p0[l] = p0[l] + v[idx + l]; p1[l] = p1[l] + v[idx + ps + l]; }
            for (int l = 0; l < 4; ++l) p0[l] = p0[l] + p1[l];
            if (alignedSize > alignedSize2)
                for (int l = 0; l < 4; ++l) p0[l] = p0[l] + v[alignedSize2 + l];
        }
        res = (p0[0] + p0[2]) + (p0[1] + p0[3]);  // predux<Packet4f>: movehl add, then add_ss
        for (long idx = alignedSize; idx < size; ++idx) res = res + v[idx];
    } else {
        res = v[0];
        for (long idx = 1; idx < size; ++idx) res = res + v[idx];
    }
    return res;
}

struct Dt3 {
    std::vector<float> keys;   // ascending angles = std::map order
    std::vector<Image> maps;   // one H x W col-major image per key
    Point2 sceneTranslation{0, 0};
    long W = 0, H = 0;
};

// Task pool with the reference's granularity (one task per item, dynamic pick-up).  The reference shares one
// BS::thread_pool between calls (batchoptimize.h:11,22: a shared_ptr), so thread start-up is not part of a timed
// search there; likewise the workers here are long-lived (one process-wide pool that grows to the largest thread count
// asked for) and a call hands them a job and waits.
class TaskPool {
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::vector<std::thread> workers;
    const std::function<void(long)>* fn = nullptr;
    std::atomic<long> next{0};
    long n = 0;
    unsigned long generation = 0;
    int want = 0, active = 0;   // workers allowed to join the current job / still running it
    bool quit = false;
    void worker(int id) {
        unsigned long seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        while (true) {
            cv_job.wait(lk, [&] { return quit || (generation != seen && id < want); });
            if (quit) return;
            seen = generation;
            const std::function<void(long)>* f = fn;
            const long total = n;
            lk.unlock();
            for (long i; (i = next.fetch_add(1)) < total;) (*f)(i);
            lk.lock();
            if (--active == 0) cv_done.notify_all();
        }
    }
public:
    void run(long count, int nthreads, const std::function<void(long)>& f) {
        static std::mutex one_job;  // one job at a time (callers are the tests and the bench: sequential)
        std::lock_guard<std::mutex> job(one_job);
        std::unique_lock<std::mutex> lk(mu);
        const int nt = (int)std::min<long>(nthreads, count);
        while ((int)workers.size() < nt) { const int id = (int)workers.size(); workers.emplace_back([this, id] { worker(id); }); }
        fn = &f; n = count; next.store(0); want = nt; active = nt; ++generation;
        cv_job.notify_all();
        cv_done.wait(lk, [&] { return active == 0; });
        fn = nullptr; want = 0;
    }
    ~TaskPool() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_job.notify_all();
        for (auto& t : workers) t.join();
    }
};
static inline void parallelFor(long n, int nthreads, const std::function<void(long)>& fn) {
    if (nthreads <= 1 || n <= 1) { for (long i = 0; i < n; ++i) fn(i); return; }
    static TaskPool* pool = new TaskPool();  // never destroyed: no join at process exit
    pool->run(n, nthreads, fn);
}

// buildCpuFeaturemap<D>, dt3cpu.h:174-234.
static inline Dt3 buildCpuFeaturemap(const Lines& scene, long depth, float coeff, float padding, int D, int nthreads,
                                     int stop_after = 3) {
    Dt3 out;
    if (scene.n() == 0) return out;
    getSceneCenteredTranslation(scene, padding, out.sceneTranslation, out.W, out.H);
    const Lines translatedScene = translate(scene, out.sceneTranslation);
    // angle set, dt3cpu.h:188-190 (std::set<float>: sorted, unique)
    for (long i = 0; i < depth; ++i) out.keys.push_back(float(i) * kPif / float(depth) - kPi2f);
    std::sort(out.keys.begin(), out.keys.end());
    out.keys.erase(std::unique(out.keys.begin(), out.keys.end()), out.keys.end());
    const long m = (long)out.keys.size();
    // classifyLines, dt3cpu.h:123-134
    std::vector<Lines> classified((size_t)m);
    for (long i = 0; i < translatedScene.n(); ++i) {
        const float* l = translatedScene.line(i);
        classified[closestOrientation(out.keys, l)].push(l[0], l[1], l[2], l[3]);
    }
    out.maps.resize((size_t)m);
    parallelFor(m, nthreads, [&](long k) { out.maps[k] = distanceTransform(classified[k], out.W, out.H, D); });
    if (stop_after < 2) return out;
    propagateOrientation(out.keys, out.maps, coeff);
    if (stop_after < 3) return out;
    for (long k = 0; k < m; ++k) lineIntegral(out.maps[k], out.keys[k]);
    return out;
}

// evaluate<Dt3Cpu> for one template and a list of translations, dt3cpu.cpp:126-179.
static inline void evaluate(const Dt3& fm, const Lines& tmpl, const std::vector<Point2>& translations,
                            std::vector<float>& scores, long* n_reads) {
    scores.clear();
    std::vector<long> bin((size_t)tmpl.n());
    for (long i = 0; i < tmpl.n(); ++i) bin[i] = closestOrientation(fm.keys, tmpl.line(i));
    std::vector<float> score_per_line((size_t)tmpl.n());
    for (const Point2& tr : translations) {
        const Lines tt = translate(tmpl, Point2{fm.sceneTranslation.x + tr.x, fm.sceneTranslation.y + tr.y});
        for (long i = 0; i < tt.n(); ++i) {
            const float* l = tt.line(i);
            int p1x = (int)l[0], p1y = (int)l[1], p2x = (int)l[2], p2y = (int)l[3];
            const Image& feature = fm.maps[bin[i]];
            float lookup_p1 = feature.at(p1y, p1x);
            float lookup_p2 = feature.at(p2y, p2x);
            score_per_line[i] = std::abs(lookup_p1 - lookup_p2);
        }
        scores.push_back(eigenSum(score_per_line.data(), tt.n()));
        if (n_reads) *n_reads += 2 * tt.n();
    }
}

struct OptimalTranslation { float score; Point2 translation; };

// optimize<BatchOptimize>, batchoptimize.cpp:15-99 (one candidate).  batchSize = 1 with the second
// break removed is optimize<DefaultOptimize>, defaultoptimize.cpp:13-66 (`kind` = 0).
static inline std::optional<OptimalTranslation> optimizeOne(const Dt3& fm, const Lines& tmpl, Point2 align_vec,
                                                           int kind, long batchSize, long* n_reads) {
    if (relativelyEqual(std::fabs(align_vec.x) + std::fabs(align_vec.y), 0.f)) return std::nullopt;
    const Point2 sav = rasterizeVector(align_vec);
    const auto mm = minmaxTranslation(tmpl, sav, fm.W, fm.H, fm.sceneTranslation);
    const float min_mul = mm[0], max_mul = mm[1];
    if (!std::isfinite(min_mul) || !std::isfinite(max_mul)) return std::nullopt;
    std::vector<float> sc;
    evaluate(fm, tmpl, {Point2{0, 0}}, sc, n_reads);
    std::vector<Point2> translations{Point2{0, 0}};
    std::vector<float> scores{sc[0]};
    if (kind == 2) {
        // optimize<IndulgentOptimize>, indulgentoptimize.cpp:33-77 (batchSize = the number of passthroughs).
        // A rejected score is "passed through" without advancing the multiplier, so the same translation
        // is scored again until the allowance is used up: the walk stops at the first rejected score like
        // DefaultOptimize.  What differs from DefaultOptimize is :59-63: (0,0) and the initial score are
        // appended again before the negative direction, so it compares against the initial score.
        const unsigned long allowed = (unsigned long)batchSize;
        auto walk = [&](int dir) {
            const long lim = dir > 0 ? static_cast<long>(max_mul) : static_cast<long>(min_mul);
            unsigned long passthroughs = 0;
            long tm = dir;
            while (dir > 0 ? tm <= lim : tm >= lim) {
                const Point2 tr{(float)tm * sav.x, (float)tm * sav.y};
                std::vector<float> one;
                evaluate(fm, tmpl, {tr}, one, n_reads);
                if (one[0] > scores.back()) {
                    if (passthroughs >= allowed) break;
                    ++passthroughs;
                    continue;
                }
                translations.push_back(tr);
                scores.push_back(one[0]);
                tm += dir;
            }
        };
        walk(+1);
        translations.push_back(Point2{0, 0});
        scores.push_back(sc[0]);
        walk(-1);
        size_t best = std::min_element(scores.begin(), scores.end()) - scores.begin();
        return OptimalTranslation{scores[best], translations[best]};
    }
    if (kind == 0) batchSize = 1;
    auto run = [&](int dir) {
        const long lim = dir > 0 ? static_cast<long>(max_mul) : static_cast<long>(min_mul);
        for (long tm = dir; dir > 0 ? tm <= lim : tm >= lim; tm += dir * batchSize) {
            std::vector<Point2> bt;
            for (long b = tm; (dir > 0 ? (b < tm + batchSize && b <= lim) : (b > tm - batchSize && b >= lim)); b += dir)
                bt.push_back(Point2{(float)b * sav.x, (float)b * sav.y});
            std::vector<float> bs;
            evaluate(fm, tmpl, bt, bs, n_reads);
            int argmin = (int)(std::min_element(bs.begin(), bs.end()) - bs.begin());
            if (bs[argmin] > scores.back()) break;
            translations.push_back(bt[argmin]);
            scores.push_back(bs[argmin]);
            if (kind == 1 && bs[argmin] < bs.back()) break;
        }
    };
    run(+1);
    run(-1);
    size_t best = std::min_element(scores.begin(), scores.end()) - scores.begin();
    return OptimalTranslation{scores[best], translations[best]};
}

// establishSearchStrategy<DefaultSearch>, defaultsearch.cpp:29-49 (+ argsort math.h:106-116,
// binarySearch math.h:137-146, getCenteredRange defaultsearch.h:40-47).
struct Combo { long tmplLine, sceneLine; };
static inline std::vector<long> argsortGreater(const std::vector<float>& v) {
    std::vector<long> ind(v.size());
    std::iota(ind.begin(), ind.end(), 0);
    std::sort(ind.begin(), ind.end(), [&v](long const i1, long const i2) { return v[i1] > v[i2]; });
    return ind;
}
static inline size_t binarySearchGreater(const std::vector<float>& sorted, float value) {
    auto it = std::lower_bound(sorted.begin(), sorted.end(), value, std::greater<float>());
    if (it == sorted.begin()) return 0;
    if (it == sorted.end()) return (size_t)((it - 1) - sorted.begin());
    return std::abs(value - *it) < std::abs(value - *(it - 1)) ? (size_t)(it - sorted.begin())
                                                              : (size_t)((it - 1) - sorted.begin());
}
static inline void getCenteredRange(size_t center_idx, size_t vec_size, size_t max_length, size_t& b, size_t& e) {
    b = (size_t)std::max(0, int(center_idx) - int(max_length / 2));
    e = std::min(size_t(b + max_length), vec_size);
    b = (size_t)std::max(0, int(e) - int(max_length));
}
static inline std::vector<Combo> defaultSearch(const Lines& tmpl, const Lines& scene, size_t maxT, size_t maxS) {
    std::vector<float> sl((size_t)scene.n()), tl((size_t)tmpl.n());
    for (long i = 0; i < scene.n(); ++i) sl[i] = getLength(scene.line(i));
    for (long i = 0; i < tmpl.n(); ++i) tl[i] = getLength(tmpl.line(i));
    std::vector<long> ssi = argsortGreater(sl), sti = argsortGreater(tl);
    std::vector<float> ssl(sl.size());
    for (size_t i = 0; i < sl.size(); ++i) ssl[i] = sl[ssi[i]];
    std::vector<Combo> out;
    const long nt = (long)std::min((size_t)tmpl.n(), maxT);
    for (long j = 0; j < nt; ++j) {
        size_t c = binarySearchGreater(ssl, tl[sti[j]]);
        size_t b, e;
        getCenteredRange(c, ssl.size(), maxS, b, e);
        for (size_t i = b; i < e; ++i) out.push_back({sti[j], ssi.at(i)});
    }
    return out;
}

// filterInRange, searchstrategies/concentricrange.h:73-84.
static inline std::vector<long> filterInRange(const Lines& l, Point2 center, float min_radius, float max_radius) {
    std::vector<long> idx;
    for (long i = 0; i < l.n(); ++i) {
        const float* p = l.line(i);
        const float cx = (p[2] + p[0]) / 2 - center.x, cy = (p[3] + p[1]) / 2 - center.y;  // getCenter - center
        const float rad = std::sqrt(cx * cx + cy * cy);
        if (rad > (min_radius - std::numeric_limits<float>::epsilon()) && rad < max_radius) idx.push_back(i);
    }
    return idx;
}

// establishSearchStrategy<ConcentricRangeStrategy>, src/searchstrategies/concentricrange.cpp:29-60.
static inline std::vector<Combo> concentricSearch(const Lines& tmpl, const Lines& scene, size_t maxT, size_t maxS,
                                                  Point2 center, float lo, float hi) {
    const std::vector<long> fidx = filterInRange(scene, center, lo, hi);
    if (fidx.empty()) return {};
    Lines filtered;
    for (long i : fidx) { const float* p = scene.line(i); filtered.push(p[0], p[1], p[2], p[3]); }
    std::vector<Combo> c = defaultSearch(tmpl, filtered, maxT, maxS);
    for (Combo& k : c) k.sceneLine = fidx.at((size_t)k.sceneLine);  // sliceVector(filtered idx, sorted idx)
    return c;
}

struct Match { int tmplIdx; float score; float transform[6]; };

// search<DefaultMatch>, defaultmatch.cpp:32-89.
struct Concentric { bool on = false; Point2 center{0, 0}; float lo = 0, hi = 0; };
static inline std::vector<Match> searchDefaultMatch(const Dt3& fm, const std::vector<Lines>& templates, const Lines& scene,
                                                    size_t maxT, size_t maxS, int kind, long batch, int nthreads,
                                                    long* n_reads_total, long* n_candidates, Concentric cr = Concentric{}) {
    std::vector<Match> all;
    if (templates.empty() || scene.n() == 0 || (fm.W == 0 && fm.H == 0)) return all;
    std::vector<Lines> aligned;
    std::vector<int> tidx;
    std::vector<Point2> alignments;
    std::vector<Mat23> transforms;
    for (size_t t = 0; t < templates.size(); ++t) {
        const Lines& tmpl = templates[t];
        if (tmpl.n() == 0) continue;
        for (const Combo& c : (cr.on ? concentricSearch(tmpl, scene, maxT, maxS, cr.center, cr.lo, cr.hi)
                                     : defaultSearch(tmpl, scene, maxT, maxS))) {
            const float* scene_line = scene.line(c.sceneLine);
            const float* tmpl_line = tmpl.line(c.tmplLine);
            Point2 align_vec = normalize(scene_line);
            auto tr = align(tmpl_line, scene_line);
            for (int f = 0; f < 2; ++f) {
                transforms.push_back(tr[f]);
                tidx.push_back((int)t);
                aligned.push_back(transform(tmpl, tr[f]));
                alignments.push_back(align_vec);
            }
        }
    }
    std::vector<std::optional<OptimalTranslation>> results(aligned.size());
    std::vector<long> reads(aligned.size(), 0);
    parallelFor((long)aligned.size(), nthreads, [&](long i) {
        results[i] = optimizeOne(fm, aligned[i], alignments[i], kind, batch, &reads[i]);
    });
    for (size_t i = 0; i < aligned.size(); ++i) {
        if (results[i].has_value()) {
            Mat23 c = combine(results[i]->translation, transforms[i]);
            Match mt;
            mt.tmplIdx = tidx[i];
            mt.score = results[i]->score;
            std::memcpy(mt.transform, c.m, sizeof c.m);
            all.push_back(mt);
        }
    }
    if (n_reads_total) { *n_reads_total = 0; for (long r : reads) *n_reads_total += r; }
    if (n_candidates) *n_candidates = (long)aligned.size();
    return all;
}

static inline Lines fromRaw(const float* p, long n) {
    Lines l;
    l.d.assign(p, p + 4 * n);
    return l;
}

}  // namespace fdcmo

// ==========================================================================================
// C API for ctypes (tests / bench cpu_baseline only)
// ==========================================================================================
using namespace fdcmo;
extern "C" {

struct FdcmoMatch { int tmpl_idx; float score; float transform[6]; };

void* fdcmo_build(const float* lines, long n, long depth, float coeff, float padding, int distance, int nthreads,
                  int stop_after) {
    Dt3* h = new Dt3(buildCpuFeaturemap(fromRaw(lines, n), depth, coeff, padding, distance, nthreads, stop_after));
    return h;
}
void fdcmo_free(void* h) { delete (Dt3*)h; }
void fdcmo_info(void* h, long* W, long* H, float* tx, float* ty, long* depth) {
    Dt3* d = (Dt3*)h;
    *W = d->W; *H = d->H; *tx = d->sceneTranslation.x; *ty = d->sceneTranslation.y; *depth = (long)d->maps.size();
}
void fdcmo_keys(void* h, float* keys) { Dt3* d = (Dt3*)h; std::memcpy(keys, d->keys.data(), d->keys.size() * 4); }
void fdcmo_slice(void* h, long k, float* out) {
    Dt3* d = (Dt3*)h;
    std::memcpy(out, d->maps[k].d.data(), d->maps[k].d.size() * 4);
}
// Construct a feature map from caller-provided slices (Dt3Cpu ctor, dt3cpu.h:55-58).
void* fdcmo_from_slices(const float* keys, long m, const float* data, long W, long H, float tx, float ty) {
    Dt3* d = new Dt3;
    d->W = W; d->H = H; d->sceneTranslation = {tx, ty};
    for (long k = 0; k < m; ++k) {
        d->keys.push_back(keys[k]);
        Image im(H, W, 0.f);
        std::memcpy(im.d.data(), data + (size_t)k * W * H, (size_t)W * H * 4);
        d->maps.push_back(std::move(im));
    }
    return d;
}

// templates: concatenated lines, offsets[T+1] in lines.  Returns the number of matches; *out is
// malloc'ed (free with fdcmo_free_matches).  stats[0] = volume reads by the reference rule,
// stats[1] = candidates.
long fdcmo_search(void* h, const float* tl, const long* offsets, long T, const float* scene, long ns, long maxT,
                  long maxS, int kind, long batch, int nthreads, FdcmoMatch** out, long* stats) {
    Dt3* d = (Dt3*)h;
    std::vector<Lines> templates((size_t)T);
    for (long t = 0; t < T; ++t) templates[t] = fromRaw(tl + 4 * offsets[t], offsets[t + 1] - offsets[t]);
    long reads = 0, cands = 0;
    std::vector<Match> m = searchDefaultMatch(*d, templates, fromRaw(scene, ns), (size_t)maxT, (size_t)maxS, kind, batch,
                                              nthreads, &reads, &cands);
    if (stats) { stats[0] = reads; stats[1] = cands; }
    *out = (FdcmoMatch*)std::malloc(std::max<size_t>(1, m.size()) * sizeof(FdcmoMatch));
    for (size_t i = 0; i < m.size(); ++i) {
        (*out)[i].tmpl_idx = m[i].tmplIdx; (*out)[i].score = m[i].score;
        std::memcpy((*out)[i].transform, m[i].transform, 24);
    }
    return (long)m.size();
}
void fdcmo_free_matches(FdcmoMatch* p) { std::free(p); }

long fdcmo_search_concentric(void* h, const float* tl, const long* offsets, long T, const float* scene, long ns, long maxT,
                             long maxS, float cx, float cy, float lo, float hi, int kind, long batch, int nthreads,
                             FdcmoMatch** out) {
    Dt3* d = (Dt3*)h;
    std::vector<Lines> templates((size_t)T);
    for (long t = 0; t < T; ++t) templates[t] = fromRaw(tl + 4 * offsets[t], offsets[t + 1] - offsets[t]);
    Concentric cr; cr.on = true; cr.center = {cx, cy}; cr.lo = lo; cr.hi = hi;
    std::vector<Match> m = searchDefaultMatch(*d, templates, fromRaw(scene, ns), (size_t)maxT, (size_t)maxS, kind, batch,
                                              nthreads, nullptr, nullptr, cr);
    *out = (FdcmoMatch*)std::malloc(std::max<size_t>(1, m.size()) * sizeof(FdcmoMatch));
    for (size_t i = 0; i < m.size(); ++i) {
        (*out)[i].tmpl_idx = m[i].tmplIdx; (*out)[i].score = m[i].score;
        std::memcpy((*out)[i].transform, m[i].transform, 24);
    }
    return (long)m.size();
}
long fdcmo_filter_in_range(const float* lines, long n, float cx, float cy, float lo, float hi, long* out) {
    auto v = filterInRange(fromRaw(lines, n), {cx, cy}, lo, hi);
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return (long)v.size();
}
long fdcmo_concentric_search(const float* tmpl, long nt, const float* scene, long ns, long maxT, long maxS, float cx,
                             float cy, float lo, float hi, long* out) {
    auto c = concentricSearch(fromRaw(tmpl, nt), fromRaw(scene, ns), (size_t)maxT, (size_t)maxS, {cx, cy}, lo, hi);
    for (size_t i = 0; i < c.size(); ++i) { out[2 * i] = c[i].tmplLine; out[2 * i + 1] = c[i].sceneLine; }
    return (long)c.size();
}

// ---- unit-level entry points for the known-answer tests ----
void fdcmo_rasterize_vector(float x, float y, float* out) { Point2 r = rasterizeVector({x, y}); out[0] = r.x; out[1] = r.y; }
long fdcmo_rasterize_line(const float* line, long* xs, long* ys, long cap) {
    std::vector<long> x, y;
    rasterizeLine(line, x, y);
    for (size_t i = 0; i < x.size() && (long)i < cap; ++i) { xs[i] = x[i]; ys[i] = y[i]; }
    return (long)x.size();
}
long fdcmo_clip_lines(const float* lines, long n, float xmin, float xmax, float ymin, float ymax, float* out) {
    Lines c = clipLines(fromRaw(lines, n), Box{xmin, xmax, ymin, ymax});
    std::memcpy(out, c.d.data(), c.d.size() * 4);
    return c.n();
}
void fdcmo_draw_lines(float* img, long rows, long cols, const float* lines, long n, float color) {
    Image im(rows, cols, 0.f);
    std::memcpy(im.d.data(), img, (size_t)rows * cols * 4);
    drawLines(im, fromRaw(lines, n), color);
    std::memcpy(img, im.d.data(), (size_t)rows * cols * 4);
}
void fdcmo_distance_transform(const float* lines, long n, long W, long H, int D, float* out) {
    Image im = distanceTransform(fromRaw(lines, n), W, H, D);
    std::memcpy(out, im.d.data(), im.d.size() * 4);
}
void fdcmo_column_pass_l2(float* img, long rows, long cols) {
    Image im(rows, cols, 0.f);
    std::memcpy(im.d.data(), img, (size_t)rows * cols * 4);
    columnPassL2(im);
    std::memcpy(img, im.d.data(), (size_t)rows * cols * 4);
}
void fdcmo_line_integral(float* img, long rows, long cols, float angle) {
    Image im(rows, cols, 0.f);
    std::memcpy(im.d.data(), img, (size_t)rows * cols * 4);
    lineIntegral(im, angle);
    std::memcpy(img, im.d.data(), (size_t)rows * cols * 4);
}
void fdcmo_scene_centered_translation(const float* lines, long n, float padding, float* t, long* size) {
    Point2 tr; long W, H;
    getSceneCenteredTranslation(fromRaw(lines, n), padding, tr, W, H);
    t[0] = tr.x; t[1] = tr.y; size[0] = W; size[1] = H;
}
void fdcmo_minmax_translation(const float* tmpl, long n, float ax, float ay, long W, long H, float ex, float ey, float* out) {
    auto r = minmaxTranslation(fromRaw(tmpl, n), {ax, ay}, W, H, {ex, ey});
    out[0] = r[0]; out[1] = r[1];
}
long fdcmo_closest_orientation(const float* keys, long m, const float* line) {
    return closestOrientation(std::vector<float>(keys, keys + m), line);
}
void fdcmo_propagate(const float* keys, long m, float* data, long W, long H, float coeff) {
    std::vector<float> k(keys, keys + m);
    std::vector<Image> maps;
    for (long i = 0; i < m; ++i) {
        Image im(H, W, 0.f);
        std::memcpy(im.d.data(), data + (size_t)i * W * H, (size_t)W * H * 4);
        maps.push_back(std::move(im));
    }
    propagateOrientation(k, maps, coeff);
    for (long i = 0; i < m; ++i) std::memcpy(data + (size_t)i * W * H, maps[i].d.data(), (size_t)W * H * 4);
}
long fdcmo_default_search(const float* tmpl, long nt, const float* scene, long ns, long maxT, long maxS, long* out) {
    auto c = defaultSearch(fromRaw(tmpl, nt), fromRaw(scene, ns), (size_t)maxT, (size_t)maxS);
    for (size_t i = 0; i < c.size(); ++i) { out[2 * i] = c[i].tmplLine; out[2 * i + 1] = c[i].sceneLine; }
    return (long)c.size();
}
void fdcmo_centered_range(long c, long n, long maxlen, long* out) {
    size_t b, e;
    getCenteredRange((size_t)c, (size_t)n, (size_t)maxlen, b, e);
    out[0] = (long)b; out[1] = (long)e;
}
void fdcmo_align(const float* tl, const float* rl, float* out12) {
    auto a = align(tl, rl);
    std::memcpy(out12, a[0].m, 24);
    std::memcpy(out12 + 6, a[1].m, 24);
}
void fdcmo_transform(const float* lines, long n, const float* T, float* out) {
    Mat23 m; std::memcpy(m.m, T, 24);
    Lines r = transform(fromRaw(lines, n), m);
    std::memcpy(out, r.d.data(), r.d.size() * 4);
}
// optimize one candidate; returns 1 and fills out[3] = score, tx, ty if it has a value.
int fdcmo_optimize(void* h, const float* tmpl, long n, float ax, float ay, int kind, long batch, float* out) {
    long reads = 0;
    auto r = optimizeOne(*(Dt3*)h, fromRaw(tmpl, n), {ax, ay}, kind, batch, &reads);
    if (!r) return 0;
    out[0] = r->score; out[1] = r->translation.x; out[2] = r->translation.y;
    return 1;
}
void fdcmo_evaluate(void* h, const float* tmpl, long n, const float* tr, long ntr, float* scores) {
    std::vector<Point2> t;
    for (long i = 0; i < ntr; ++i) t.push_back({tr[2 * i], tr[2 * i + 1]});
    std::vector<float> s;
    evaluate(*(Dt3*)h, fromRaw(tmpl, n), t, s, nullptr);
    std::memcpy(scores, s.data(), s.size() * 4);
}
float fdcmo_eigen_sum(const float* v, long n) { return eigenSum(v, n); }
// unit-level entry points for the math.test.cpp known answers
void fdcmo_argsort_greater(const float* v, long n, long* out) {
    auto ind = argsortGreater(std::vector<float>(v, v + n));
    std::memcpy(out, ind.data(), ind.size() * sizeof(long));
}
long fdcmo_binary_search_greater(const float* sorted_desc, long n, float value) {
    return (long)binarySearchGreater(std::vector<float>(sorted_desc, sorted_desc + n), value);
}
void fdcmo_minmax_point(const float* lines, long n, float* out4) {
    Point2 mn, mx;
    minmaxPoint(fromRaw(lines, n), mn, mx);
    out4[0] = mn.x; out4[1] = mn.y; out4[2] = mx.x; out4[3] = mx.y;
}
void fdcmo_line_props(const float* line, float* out4) {  // angle, length, normalised direction
    out4[0] = getAngle(line); out4[1] = getLength(line);
    Point2 d = normalize(line); out4[2] = d.x; out4[3] = d.y;
}
void fdcmo_translate(const float* lines, long n, float tx, float ty, float* out) {
    Lines r = translate(fromRaw(lines, n), {tx, ty});
    std::memcpy(out, r.d.data(), r.d.size() * 4);
}
void fdcmo_combine(float tx, float ty, const float* T, float* out6) {
    Mat23 m; std::memcpy(m.m, T, 24);
    Mat23 c = combine({tx, ty}, m);
    std::memcpy(out6, c.m, 24);
}
float fdcmo_atanf(float x) { return std::atan(x); }
// sortMatches, matchstrategy.h:46-50: std::sort over the Match structs with operator< on the score.
void fdcmo_sort_matches(Match* m, long n) {
    std::sort(m, m + n, [](const Match& a, const Match& b) { return a.score < b.score; });
}
// sortMatches(matches, maxNumCandidates), matchstrategy.h:52-55
void fdcmo_partial_sort_matches(Match* m, long n, long k) {
    std::partial_sort(m, m + std::min(k, n), m + n, [](const Match& a, const Match& b) { return a.score < b.score; });
}
// penalize<DefaultPenalty> (defaultpenalty.cpp:33-45) / penalize<ExponentialPenalty> (exponentialpenalty.cpp:33-48);
// returns -1 where templatelengths.at() throws.
int fdcmo_penalize(int exponential, float tau, Match* m, long n, const float* lengths, long n_lengths) {
    for (long i = 0; i < n; ++i) {
        if (m[i].tmplIdx < 0 || m[i].tmplIdx >= n_lengths) return -1;
        const float len = std::max(lengths[m[i].tmplIdx], 1e-6f);
        m[i].score = exponential ? m[i].score / std::pow(len, tau) : m[i].score / len;
    }
    return 0;
}
}
