// exact_owner_sim.cpp -- CPU check (diagnostic, not product code) of the statement the balanced L2 sweep rests on:
//
//   For an image with W^2 + H^2 <= 2^24 the reference's second 1-D pass (imgproc.h:91-130), run on the output of its
//   first pass, gives   out[q] = base(o, q) + (q - o)^2   with  o = the EXACT owner of pixel q -- the seeded column that
//   minimises f_u + (q - u)^2 over the integers, the smallest such column on a tie -- and
//   base(o, q) = f[o] for q <= o, out[o] for q > o (the cell the reference reads back after overwriting it, :126-127).
//
// Every number of the pass is then an integer below 2^24 (or FLT_MAX), the only rounded quantity is the quotient s, and
// for a quotient N / D with D <= 2 (W - 1) an integer q <= 2047 satisfies q <= RN(N / D) <=> q <= N / D.
//
//   usage: exact_owner_sim <seed file of make_seeds.py> [rows step=1]      (part 1: every row of every slice)
//          exact_owner_sim --random <cases> <n>                             (part 2: random and near-degenerate columns)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

static const float FMAX = std::numeric_limits<float>::max();

// the reference's pass on one vector, imgproc.h:100-128 (as oracle/fdcm_oracle.cpp: columnPassL2 restates it)
static void literal_pass(std::vector<float>& f) {
    const long R = (long)f.size();
    std::vector<long> v((size_t)R);
    std::vector<float> z((size_t)R + 1);
    long k = 0;
    v[0] = 0;
    z[0] = -INFINITY;
    z[1] = INFINITY;
    for (long q = 1; q < R; ++q) {
        while (true) {
            long const v_k = v[k];
            float const s = (f[q] + (q * q) - f[v_k] - (v_k * v_k)) / (2 * q - 2 * v_k);
            if (s > z[k]) { ++k; v[k] = q; z[k] = s; z[k + 1] = INFINITY; break; }
            --k;
        }
    }
    k = 0;
    for (long q = 0; q < R; ++q) {
        while (z[k + 1] < (float)q) ++k;
        long const v_k = v[k];
        long const d = std::labs(q - v_k);
        f[q] = f[v_k] + (d * d);
    }
}

// the exact statement: integer owners by brute force, then the in-place rule
static void exact_pass(std::vector<float>& f) {
    const long R = (long)f.size();
    std::vector<long> cols;
    for (long u = 0; u < R; ++u) if (f[u] != FMAX) cols.push_back(u);
    if (cols.empty()) return;  // every cell stays FLT_MAX (FLT_MAX + d^2 == FLT_MAX)
    std::vector<int64_t> fi((size_t)R, 0);
    for (long u : cols) fi[u] = (int64_t)f[u];
    std::vector<int64_t> out((size_t)R);
    for (long q = 0; q < R; ++q) {
        int64_t best = INT64_MAX; long o = -1;
        for (long u : cols) { const int64_t c = fi[u] + (q - u) * (q - u); if (c < best) { best = c; o = u; } }
        const int64_t base = (o < q) ? out[o] : fi[o];
        out[q] = base + (q - o) * (q - o);
    }
    for (long q = 0; q < R; ++q) f[q] = (float)out[q];
}

static uint64_t rng_state = 0x1234567;
static uint64_t rnd() { rng_state += 0x9E3779B97F4A7C15ull; uint64_t z = rng_state; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "--random")) {
        const long cases = atol(argv[2]), n = atol(argv[3]);
        long bad = 0;
        for (long c = 0; c < cases; ++c) {
            std::vector<float> f((size_t)n);
            const int kind = (int)(rnd() % 6);
            const long H = 1 + (long)(rnd() % 2048);  // pass-1 values are squares of distances below H
            const double dens = (double)(rnd() % 1000) / 1000.0;
            // kinds: 0 random squares, 1 squares of a smooth line (a scene line crossing), 2 near-collinear P = f + u^2,
            // 3 few seeds, 4 parabola-like (all points on the hull), 5 mixture with plateaus
            const double a = ((double)(rnd() % 2001) - 1000.0) / 500.0, b = (double)(rnd() % 2048);
            const long c0 = (long)(rnd() % n);
            const double eps = (double)(rnd() % 1000) / 1.0e5;
            for (long u = 0; u < n; ++u) {
                float val = FMAX;
                const bool seeded = kind == 3 ? (rnd() % 97 == 0) : ((double)(rnd() % 1000) / 1000.0 < dens);
                if (seeded) {
                    long d = 0;
                    if (kind == 0 || kind == 3) d = (long)(rnd() % H);
                    else if (kind == 1) d = std::labs((long)std::llround(a * (double)(u - c0) + b)) % H;
                    else if (kind == 2) {
                        // P_u close to the line K + L u with a slight convexity eps (u - c0)^2: f = P - u^2 must be a square of
                        // an integer, so take the nearest square below
                        const double P = 2.0 * (double)n * (double)n + 2.0 * (double)n * (double)(u - c0) * 0.999 + eps * (double)(u - c0) * (double)(u - c0);
                        const double fv = P - (double)u * (double)u;
                        d = fv <= 0 ? 0 : (long)std::floor(std::sqrt(fv));
                        if (rnd() % 3 == 0) d += (long)(rnd() % 3);
                        if (d >= 2896) d = 2895;
                    } else if (kind == 4) d = std::labs(u - c0) / 2 + (long)(rnd() % 2);
                    else d = ((rnd() % 5) == 0) ? (long)(rnd() % H) : std::labs((long)std::llround(a * 8.0) + (long)(u / 64) * 3) % H;
                    if ((double)d * (double)d + (double)n * (double)n > 16777216.0) d = (long)std::floor(std::sqrt(16777216.0 - (double)n * (double)n));
                    val = (float)(d * d);
                }
                f[u] = val;
            }
            std::vector<float> g = f, h = f;
            literal_pass(g);
            exact_pass(h);
            if (memcmp(g.data(), h.data(), (size_t)n * 4)) {
                ++bad;
                if (bad < 5) {
                    for (long q = 0; q < n; ++q) if (g[q] != h[q]) { fprintf(stderr, "case %ld kind %d: first difference at %ld: literal %.9g exact %.9g\n", c, kind, q, g[q], h[q]); break; }
                }
            }
        }
        printf("random: %ld cases of %ld cells, %ld differ\n", cases, n, bad);
        return bad != 0;
    }
    FILE* fp = fopen(argv[1], "rb");
    int32_t hdr[3];
    if (!fp || fread(hdr, 4, 3, fp) != 3) return 1;
    const int m = hdr[0], W = hdr[1], H = hdr[2];
    const int step = argc > 2 ? atoi(argv[2]) : 1;
    std::vector<uint8_t> seed((size_t)W * H);
    long rows = 0, bad = 0;
    for (int k = 0; k < m; ++k) {
        if (fread(seed.data(), 1, seed.size(), fp) != seed.size()) return 2;
        // pass 1: squared distance to the nearest seed of the column, FLT_MAX for a column without one
        std::vector<float> p1((size_t)W * H);
        for (int x = 0; x < W; ++x) {
            const uint8_t* c = &seed[(size_t)x * H];
            bool any = false;
            for (int y = 0; y < H; ++y) any |= c[y] != 0;
            long last = -(1 << 20);
            for (int y = 0; y < H; ++y) { if (c[y]) last = y; p1[(size_t)x * H + y] = (float)(y - last); }
            long nxt = 1 << 20;
            for (int y = H - 1; y >= 0; --y) { if (c[y]) nxt = y; const float d = std::min(p1[(size_t)x * H + y], (float)(nxt - y)); p1[(size_t)x * H + y] = any ? d * d : FMAX; }
        }
        for (int y = 0; y < H; y += step) {
            std::vector<float> f((size_t)W);
            for (int x = 0; x < W; ++x) f[x] = p1[(size_t)x * H + y];
            std::vector<float> g = f, h = f;
            literal_pass(g);
            exact_pass(h);
            ++rows;
            if (memcmp(g.data(), h.data(), (size_t)W * 4)) {
                ++bad;
                if (bad < 5) for (int q = 0; q < W; ++q) if (g[q] != h[q]) { fprintf(stderr, "slice %d row %d: first difference at %d: literal %.9g exact %.9g\n", k, y, q, g[q], h[q]); break; }
            }
        }
        fprintf(stderr, "slice %d done, %ld rows, %ld differ\n", k, rows, bad);
    }
    printf("%d x %d x %d: %ld rows compared, %ld differ\n", m, W, H, rows, bad);
    return bad != 0;
}
