// cursor_sim.cpp -- CPU statistics (diagnostic, not product code): how long is the local run's chain of k_sweep_balanced
// under different lane models?  A workgroup of 512 lanes works on R rows x S column ranges (R * S = 512) of one slice.
//   A   the wave shares one column cursor; a column costs the maximum over the wave's lanes of its tests (round 4's kernel)
//   A2  same, two stack entries tested per pass: ceil(tests / 2)
//   B   every lane has its own column cursor and makes one test per pass: a lane costs the sum of its tests,
//       a wave the maximum over its lanes
//   B2  own cursor, two entries per pass
//   D<n> own cursors inside a sliding window of n columns (a lane runs at most n columns ahead of the wave's slowest lane)
//   C<n> own cursors inside blocks of n columns, the wave moves to the next block when its slowest lane is through
// usage: cursor_sim <seed file of make_seeds.py> <S> [rows per workgroup = 512 / S] [min columns per range = 16]
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct Ent { int v; float f, z; };
static inline float isect(float fq, int q, float fv, int v) {
    const float qf = (float)q, vf = (float)v;
    return ((fq + qf * qf) - fv - vf * vf) / (2 * qf - 2 * vf);
}
struct Dist {
    std::vector<long> v;
    void add(long x) { v.push_back(x); }
    void print(const char* name) {
        if (v.empty()) { printf("%-40s (none)\n", name); return; }
        std::sort(v.begin(), v.end());
        double sum = 0; for (long x : v) sum += (double)x;
        printf("%-40s n %7zu  mean %8.1f  p50 %5ld  p90 %5ld  p99 %5ld  max %5ld  sum %.3e\n", name, v.size(), sum / (double)v.size(), v[v.size() / 2],
               v[(size_t)((double)v.size() * 0.9)], v[(size_t)((double)v.size() * 0.99)], v.back(), sum);
    }
};
int main(int argc, char** argv) {
    FILE* fp = fopen(argv[1], "rb");
    int32_t hdr[3];
    if (!fp || fread(hdr, 4, 3, fp) != 3) return 1;
    const int m = hdr[0], W = hdr[1], H = hdr[2];
    const int Smax = argc > 2 ? atoi(argv[2]) : 8, Rarg = argc > 3 ? atoi(argv[3]) : 512 / Smax, mincols = argc > 4 ? atoi(argv[4]) : 16;
    std::vector<uint8_t> seed((size_t)W * H);
    Dist wgA, wgA2, wgB, wgB2, waveA, waveB, ncols, wgC[4], wgD[3];
    const int dws[3] = {4, 8, 16};
    const int bcs[4] = {4, 8, 16, 32};
    double sumA = 0, sumB = 0;
    for (int k = 0; k < m; ++k) {
        if (fread(seed.data(), 1, seed.size(), fp) != seed.size()) return 2;
        std::vector<int> cols;
        for (int x = 0; x < W; ++x) { bool any = false; for (int y = 0; y < H && !any; ++y) any = seed[(size_t)x * H + y]; if (any) cols.push_back(x); }
        const int n = (int)cols.size();
        if (n == 0) continue;
        const int S = std::min(Smax, std::max(1, n / mincols)), R = Rarg;
        std::vector<float> f((size_t)n * H);
        for (int j = 0; j < n; ++j) {
            const uint8_t* c = &seed[(size_t)cols[j] * H];
            int last = -(1 << 20);
            for (int y = 0; y < H; ++y) { if (c[y]) last = y; f[(size_t)j * H + y] = (float)(y - last); }
            int nxt = 1 << 20;
            for (int y = H - 1; y >= 0; --y) { if (c[y]) nxt = y; float d = std::min(f[(size_t)j * H + y], (float)(nxt - y)); f[(size_t)j * H + y] = d * d; }
        }
        long sliceA = 0, sliceB = 0;
        for (int c0 = 0; c0 < H; c0 += R) {
            const int RR = std::min(R, H - c0);
            // tests[w][r][column]: tests of column j in (range w, row r)
            long gA = 0, gA2 = 0, gB = 0, gB2 = 0, gC[4] = {0, 0, 0, 0}, gD[3] = {0, 0, 0};
            // lanes of the workgroup in wave order: lane id = w * R + r; wave = id / 64
            const int nl = S * R;
            std::vector<std::vector<int>> T(nl);  // per lane: tests per column of its range
            for (int w = 0; w < S; ++w) {
                const int k0 = (int)((long)n * w / S), k1 = (int)((long)n * (w + 1) / S);
                for (int r = 0; r < RR; ++r) {
                    std::vector<Ent> s;
                    s.push_back(Ent{cols[k0], f[(size_t)k0 * H + c0 + r], -INFINITY});
                    auto& t = T[w * R + r];
                    for (int j = k0 + 1; j < k1; ++j) {
                        const float fq = f[(size_t)j * H + c0 + r];
                        int tt = 0; float sv;
                        for (;;) { ++tt; sv = isect(fq, cols[j], s.back().f, s.back().v); if (sv > s.back().z) break; s.pop_back(); }
                        s.push_back(Ent{cols[j], fq, sv});
                        t.push_back(tt);
                    }
                }
            }
            for (int w0 = 0; w0 < nl; w0 += 64) {
                long a = 0, a2 = 0, b = 0, b2 = 0;
                size_t mc = 0;
                for (int l = w0; l < std::min(nl, w0 + 64); ++l) mc = std::max(mc, T[l].size());
                for (size_t j = 0; j < mc; ++j) {
                    int mx = 0;
                    for (int l = w0; l < std::min(nl, w0 + 64); ++l) if (j < T[l].size()) mx = std::max(mx, T[l][j]);
                    a += mx; a2 += (mx + 1) / 2;
                }
                for (int l = w0; l < std::min(nl, w0 + 64); ++l) {
                    long sb = 0, sb2 = 0;
                    for (int t : T[l]) { sb += t; sb2 += (t + 1) / 2; }
                    b = std::max(b, sb); b2 = std::max(b2, sb2);
                }
                for (int bi = 0; bi < 4; ++bi) {
                    long c = 0;
                    for (size_t j0 = 0; j0 < mc; j0 += bcs[bi]) {
                        long mx = 0;
                        for (int l = w0; l < std::min(nl, w0 + 64); ++l) { long sb = 0; for (size_t j = j0; j < std::min(T[l].size(), j0 + bcs[bi]); ++j) sb += T[l][j]; mx = std::max(mx, sb); }
                        c += mx;
                    }
                    gC[bi] = std::max(gC[bi], c);
                }
                for (int di = 0; di < 3; ++di) {
                    const int n0 = w0, n1 = std::min(nl, w0 + 64);
                    std::vector<size_t> cj(n1 - n0, 0); std::vector<int> rem(n1 - n0, 0);
                    for (int l = n0; l < n1; ++l) rem[l - n0] = T[l].empty() ? 0 : T[l][0];
                    long passes = 0;
                    for (;;) {
                        size_t ws = (size_t)-1; bool any = false;
                        for (int l = n0; l < n1; ++l) if (cj[l - n0] < T[l].size()) { ws = std::min(ws, cj[l - n0]); any = true; }
                        if (!any) break;
                        ++passes;
                        for (int l = n0; l < n1; ++l) {
                            const int i = l - n0;
                            if (cj[i] >= T[l].size() || cj[i] >= ws + (size_t)dws[di]) continue;
                            if (--rem[i] == 0) { ++cj[i]; if (cj[i] < T[l].size()) rem[i] = T[l][cj[i]]; }
                        }
                    }
                    gD[di] = std::max(gD[di], passes);
                }
                waveA.add(a); waveB.add(b);
                gA = std::max(gA, a); gA2 = std::max(gA2, a2); gB = std::max(gB, b); gB2 = std::max(gB2, b2);
                sliceA += a; sliceB += b;
            }
            wgA.add(gA); wgA2.add(gA2); wgB.add(gB); wgB2.add(gB2); for (int bi = 0; bi < 4; ++bi) wgC[bi].add(gC[bi]); for (int di = 0; di < 3; ++di) wgD[di].add(gD[di]);
        }
        ncols.add(n);
        sumA += sliceA; sumB += sliceB;
    }
    printf("%d x %d x %d, S <= %d, %d rows per workgroup, >= %d columns per range\n", m, W, H, Smax, Rarg, mincols);
    ncols.print("slice: seeded columns");
    waveA.print("wave A (shared cursor)");
    waveB.print("wave B (lane cursors)");
    wgA.print("workgroup A: longest wave");
    wgA2.print("workgroup A2 (2 entries per pass)");
    wgB.print("workgroup B (lane cursors)");
    wgB2.print("workgroup B2 (cursors, 2 per pass)");
    for (int di = 0; di < 3; ++di) { char nm[64]; snprintf(nm, 64, "workgroup D%d (sliding window)", dws[di]); wgD[di].print(nm); }
    for (int bi = 0; bi < 4; ++bi) { char nm[64]; snprintf(nm, 64, "workgroup C%d (cursors inside blocks)", bcs[bi]); wgC[bi].print(nm); }
    return 0;
}
