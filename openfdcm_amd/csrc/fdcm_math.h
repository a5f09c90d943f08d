// fdcm_math.h -- exact-float geometry shared by the host side and the HIP kernels.
//
// Every function here is a chain of IEEE-754 binary32 operations (+ - * / sqrt, compares,
// truncation) evaluated in the order the reference evaluates them, so that host (x86-64, no FMA)
// and device (gfx950, compiled with -ffp-contract=off and correctly rounded divide/sqrt) produce
// the same bits.  Reference locations are cited per function (paths relative to the reference).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <float.h>

#define FDCM_HD __host__ __device__ __forceinline__

namespace fdcm {

static constexpr float kPif = 3.14159265358979323846f;   // M_PIf,   math.h:38-40
static constexpr float kPi2f = 1.57079632679489661923f;  // M_PI_2f, math.h:44-46

FDCM_HD float f_from_bits(uint32_t u) {
    union { uint32_t u; float f; } c;
    c.u = u;
    return c.f;
}
FDCM_HD uint32_t bits_from_f(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    return c.u;
}
FDCM_HD bool f_signbit(float f) { return (bits_from_f(f) >> 31) != 0; }
FDCM_HD bool f_isnan(float f) { return (bits_from_f(f) & 0x7fffffffu) > 0x7f800000u; }
FDCM_HD bool f_isfinite(float f) { return (bits_from_f(f) & 0x7f800000u) != 0x7f800000u; }
FDCM_HD float f_inf() { return f_from_bits(0x7f800000u); }
FDCM_HD float f_nan() { return f_from_bits(0x7fc00000u); }
FDCM_HD float f_abs(float f) { return f_from_bits(bits_from_f(f) & 0x7fffffffu); }
// std::min / std::max semantics (second operand wins only on a strict compare).
FDCM_HD float std_min(float a, float b) { return (b < a) ? b : a; }
FDCM_HD float std_max(float a, float b) { return (a < b) ? b : a; }

// Bit-exact restatement of this image's glibc (2.35) atanf, sysdeps/ieee754/flt-32/s_atanf.c
// (fdlibm float port: argument reduction to 5 intervals + an 11-term odd/even split polynomial).
// getAngle (math.h:295-299) is atanf(dy/dx) and decides the orientation bin of every aligned
// template line, so the device must reproduce the host libm.  tests/test_capi_load.py checks equality with
// libm atanf on every 257th of the 2^32 inputs (and on all of them in its slow test) through
// fdcm_selftest_atanf; the first search of a process runs a sampled check as well and refuses to run on a
// host whose libm disagrees (a newer glibc ships a correctly rounded atanf).
//
// The algorithm and constants are those of fdlibm's s_atanf.c, which carries this notice:
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//   Developed at SunPro, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this software is freely granted, provided that this
//   notice is preserved.
// (float conversion of the original by Ian Lance Taylor, Cygnus Support.)
FDCM_HD float atanf_glibc(float x) {
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f,  -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
                          9.0908870101e-02f,  -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
                          4.9768779427e-02f,  -3.6531571299e-02f, 1.6285819933e-02f};
    const int32_t hx = (int32_t)bits_from_f(x);
    const int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {  // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;  // NaN
        if (hx > 0) return atanhi[3] + atanlo[3];
        return -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {  // |x| < 0.4375
        if (ix < 0x31000000) return x;  // |x| < 2^-29
        id = -1;
    } else {
        x = f_abs(x);
        if (ix < 0x3f980000) {      // |x| < 1.1875
            if (ix < 0x3f300000) {  // 7/16 <= |x| < 11/16
                id = 0;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            } else {                // 11/16 <= |x| < 19/16
                id = 1;
                x = (x - 1.0f) / (x + 1.0f);
            }
        } else {
            if (ix < 0x401c0000) {  // |x| < 2.4375
                id = 2;
                x = (x - 1.5f) / (1.0f + 1.5f * x);
            } else {                // 2.4375 <= |x| < 2^25
                id = 3;
                x = -1.0f / x;
            }
        }
    }
    float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return (hx < 0) ? -z : z;
}

// relativelyEqual<float,float>(a, b) with the default double tolerances, math.h:182-188.
FDCM_HD bool relatively_equal(float a, float b) {
    const float fa = f_abs(a), fb = f_abs(b);
    const float mx = (fa < fb) ? fb : fa;
    return (double)f_abs(a - b) <= (double)FLT_EPSILON + 1e-10 * (double)mx;
}

// allClose(a, b) on 2-vectors with rtol = 0.f, atol = 1e-5f, math.h:202-208.
FDCM_HD bool all_close2(float ax, float ay, float bx, float by) {
    return (f_abs(ax - bx) <= (1e-5f + 0.0f * f_abs(bx))) && (f_abs(ay - by) <= (1e-5f + 0.0f * f_abs(by)));
}

// rasterizeVector, drawing.h:57-67.  The reference's mixed double terms reduce to exact sign
// flips: tan - 2.0*c*tan is tan or -tan exactly, so the float result is (+-1, +-tan) or
// (+-(1/tan), +-1).  Written with the same select structure; -0.0 results are preserved
// (0.0 - 2.0*1*0.0 etc.) by computing in double exactly like the reference.
FDCM_HD void rasterize_vector(float vx, float vy, float& rx, float& ry) {
    const float tan_angle = vy / vx;
    if (tan_angle >= -1.0f && tan_angle < 1.0f) {
        const int c1 = vx < 0 ? 1 : 0;
        rx = (float)(1 - 2 * c1);
        ry = (float)((double)tan_angle - 2.0 * (double)c1 * (double)tan_angle);
        return;
    }
    const int c2 = vy < 0 ? 1 : 0;
    const float inv = 1.0f / tan_angle;
    rx = (float)((double)inv - 2.0 * (double)c2 * (double)inv);
    ry = (float)(1 - 2 * c2);
}

// closestOrientation over the sorted key list, dt3cpu.h:93-114 (std::map::upper_bound + the
// wrap-around branch).  Returns the slice index.
FDCM_HD int closest_orientation(const float* keys, int m, float line_angle) {
    int lo = 0, hi = m;  // upper_bound: first key with line_angle < key
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (line_angle < keys[mid]) hi = mid; else lo = mid + 1;
    }
    int it = lo;
    if (it != m && it != 0) {
        const float upper_bound_diff = f_abs(line_angle - keys[it]);
        const float lower_bound_diff = f_abs(line_angle - keys[it - 1]);
        return (lower_bound_diff < upper_bound_diff) ? it - 1 : it;
    }
    it = m - 1;
    const float angle1 = line_angle - keys[0];
    const float angle2 = line_angle - keys[it];
    if (std_min(angle1, f_abs(angle1 - kPif)) < std_min(angle2, f_abs(angle2 - kPif))) return 0;
    return it;
}

// Eigen maxCoeff/minCoeff<PropagateNaN> over 4 values.
FDCM_HD float max4_prop_nan(const float* a) {
    float r = a[0];
    for (int i = 0; i < 4; ++i) {
        if (f_isnan(a[i])) return a[i];
        r = std_max(r, a[i]);
    }
    return r;
}
FDCM_HD float min4_prop_nan(const float* a) {
    float r = a[0];
    for (int i = 0; i < 4; ++i) {
        if (f_isnan(a[i])) return a[i];
        r = std_min(r, a[i]);
    }
    return r;
}

// detail::minmaxTranslation, dt3cpu.cpp:30-75, on an already reduced bounding box
// (minmaxPoint, math.h:166-171).  W,H = feature size, (ex,ey) = scene translation.
FDCM_HD void minmax_translation(float mnx, float mny, float mxx, float mxy, float ax, float ay, float W, float H,
                                float ex, float ey, float& min_mul, float& max_mul) {
    const float inf = f_inf();
    if (all_close2(ax, ay, 0.f, 0.f)) { min_mul = inf; max_mul = inf; return; }
    const float size[2] = {W, H};
    const float minp[2] = {mnx + ex, mny + ey};
    const float maxp[2] = {mxx + ex, mxy + ey};
    if ((size[0] - 1 - maxp[0]) < 0 || (size[1] - 1 - maxp[1]) < 0) { min_mul = max_mul = f_nan(); return; }
    if (minp[0] < 0 || minp[1] < 0) { min_mul = max_mul = f_nan(); return; }
    const float av[2] = {ax, ay};
    float pos[2][4], neg[2][4];
    for (int r = 0; r < 2; ++r) {
        float mult[4];
        mult[0] = -maxp[r];
        mult[1] = -minp[r];
        mult[2] = (size[r] - maxp[r] - 1.f);
        mult[3] = (size[r] - minp[r] - 1.f);
        for (int c = 0; c < 4; ++c) {
            const float q = mult[c] / av[r];
            const bool sgn = f_signbit(q);
            pos[r][c] = sgn ? inf : q;
            neg[r][c] = sgn ? q : -inf;
        }
    }
    const float e00 = max4_prop_nan(neg[0]), e01 = max4_prop_nan(neg[1]);
    const float e10 = min4_prop_nan(pos[0]), e11 = min4_prop_nan(pos[1]);
    if (f_isfinite(e00) && f_isfinite(e01) && f_isfinite(e10) && f_isfinite(e11)) {
        min_mul = std_max(e00, e01);
        max_mul = std_min(e10, e11);
    } else if (f_isfinite(e00) && f_isfinite(e10)) {
        min_mul = e00;
        max_mul = e10;
    } else {
        min_mul = e01;
        max_mul = e11;
    }
}

// align, math.h:387-406: the two rigid transforms (row-major 2x3) that put tmpl line `tl` on
// scene line `rl` (centre on centre, direction on direction / reversed direction).
FDCM_HD void align_pair(const float* tl, const float* rl, float* t1, float* t2) {
    // normalize(): colwise().normalized() = v / sqrt(x*x + y*y), math.h:331-333
    float tdx = tl[2] - tl[0], tdy = tl[3] - tl[1];
    const float tn = sqrtf(tdx * tdx + tdy * tdy);
    tdx = tdx / tn; tdy = tdy / tn;
    float adx = rl[2] - rl[0], ady = rl[3] - rl[1];
    const float an = sqrtf(adx * adx + ady * ady);
    adx = adx / an; ady = ady / an;
    const float c = adx * tdx + ady * tdy;
    const float s = ady * tdx - adx * tdy;
    const float rcx = (rl[2] + rl[0]) / 2, rcy = (rl[3] + rl[1]) / 2;
    {
        const float x1 = c * tl[0] + (-s) * tl[1], y1 = s * tl[0] + c * tl[1];
        const float x2 = c * tl[2] + (-s) * tl[3], y2 = s * tl[2] + c * tl[3];
        t1[0] = c; t1[1] = -s; t1[2] = rcx - (x2 + x1) / 2;
        t1[3] = s; t1[4] = c;  t1[5] = rcy - (y2 + y1) / 2;
    }
    {
        const float x1 = (-c) * tl[0] + s * tl[1], y1 = (-s) * tl[0] + (-c) * tl[1];
        const float x2 = (-c) * tl[2] + s * tl[3], y2 = (-s) * tl[2] + (-c) * tl[3];
        t2[0] = -c; t2[1] = s;  t2[2] = rcx - (x2 + x1) / 2;
        t2[3] = -s; t2[4] = -c; t2[5] = rcy - (y2 + y1) / 2;
    }
}

// getCenteredRange, defaultsearch.h:40-47.
FDCM_HD void centered_range(int center_idx, int vec_size, int max_length, int& b, int& e) {
    int bb = center_idx - max_length / 2;
    if (bb < 0) bb = 0;
    e = (bb + max_length < vec_size) ? bb + max_length : vec_size;
    b = e - max_length;
    if (b < 0) b = 0;
}

// binarySearch(sorted descending, value, std::greater), math.h:137-146.
FDCM_HD int binary_search_greater(const float* sorted, int n, float value) {
    int lo = 0, hi = n;  // lower_bound with comp = greater: first i with !(sorted[i] > value)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sorted[mid] > value) lo = mid + 1; else hi = mid;
    }
    if (lo == 0) return 0;
    if (lo == n) return n - 1;
    return (f_abs(value - sorted[lo]) < f_abs(value - sorted[lo - 1])) ? lo : lo - 1;
}

}  // namespace fdcm
