// cut_sim.cpp -- CPU statistics (diagnostic, not product code): STATIC range cuts of k_sweep_balanced by a weight per column
// instead of by equal column counts.  A wave's cost is the sum over its columns of (max over the 64 rows of the tests the
// column takes) + overhead; the pathological waves are those whose columns carry seeds INSIDE the chunk's rows (some row is
// next to the seed and pops everything the column now dominates).  Weight = 1 + beta * [the column has a seed inside the
// chunk] (+ gamma * popcount of its seed word / 64): what the kernel could compute from the descriptors before the run.
//   usage: cut_sim <seed file> [ovh = 0.5]
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
int main(int argc, char** argv) {
    FILE* fp = fopen(argv[1], "rb");
    int32_t hdr[3];
    if (!fp || fread(hdr, 4, 3, fp) != 3) return 1;
    const int m = hdr[0], W = hdr[1], H = hdr[2];
    const double ovh = argc > 2 ? atof(argv[2]) : 0.5;
    const int S = 8, R = 64;
    std::vector<uint8_t> seed((size_t)W * H);
    const double betas[] = {0, 1, 2, 3, 4, 6, 8};
    const int NB = 7;
    std::vector<double> worst(NB, 0), sum_max(NB, 0);
    std::vector<std::vector<double>> all(NB);
    double sa = 0, sn = 0, ca = 0, cn = 0;
    long nwg = 0;
    for (int k = 0; k < m; ++k) {
        if (fread(seed.data(), 1, seed.size(), fp) != seed.size()) return 2;
        std::vector<int> cols;
        for (int x = 0; x < W; ++x) { bool any = false; for (int y = 0; y < H && !any; ++y) any = seed[(size_t)x * H + y]; if (any) cols.push_back(x); }
        const int n = (int)cols.size();
        if (n < 16 * S) continue;
        std::vector<float> f((size_t)n * H);
        for (int j = 0; j < n; ++j) {
            const uint8_t* c = &seed[(size_t)cols[j] * H];
            int last = -(1 << 20);
            for (int y = 0; y < H; ++y) { if (c[y]) last = y; f[(size_t)j * H + y] = (float)(y - last); }
            int nxt = 1 << 20;
            for (int y = H - 1; y >= 0; --y) { if (c[y]) nxt = y; float d = std::min(f[(size_t)j * H + y], (float)(nxt - y)); f[(size_t)j * H + y] = d * d; }
        }
        for (int c0 = 0; c0 < H; c0 += R) {
            const int RR = std::min(R, H - c0);
            std::vector<char> anchor(n, 0);
            for (int j = 0; j < n; ++j) { const uint8_t* c = &seed[(size_t)cols[j] * H]; for (int y = c0; y < c0 + RR; ++y) anchor[j] |= c[y]; }
            auto cost_of = [&](int k0, int k1) {  // wave cost of the columns [k0, k1) with a fresh stack at k0
                std::vector<double> cost(std::max(0, k1 - k0), 0.0);
                for (int r = 0; r < RR; ++r) {
                    std::vector<Ent> s;
                    s.push_back(Ent{cols[k0], f[(size_t)k0 * H + c0 + r], -INFINITY});
                    for (int j = k0 + 1; j < k1; ++j) {
                        const float fq = f[(size_t)j * H + c0 + r];
                        int tt = 0; float sv;
                        for (;;) { ++tt; sv = isect(fq, cols[j], s.back().f, s.back().v); if (sv > s.back().z) break; s.pop_back(); }
                        s.push_back(Ent{cols[j], fq, sv});
                        cost[j - k0] = std::max(cost[j - k0], (double)tt);
                    }
                }
                return cost;
            };
            {   // how much an anchor column costs against the others (whole row as one range)
                auto c = cost_of(0, n);
                for (int j = 1; j < n; ++j) { if (anchor[j]) { sa += c[j]; ++ca; } else { sn += c[j]; ++cn; } }
            }
            for (int bi = 0; bi < NB; ++bi) {
                const double beta = betas[bi];
                double tot = 0; for (int j = 0; j < n; ++j) tot += 1 + beta * anchor[j];
                double mx = 0, acc = 0; int k0 = 0, w = 0;
                for (int j = 0; j < n; ++j) {
                    acc += 1 + beta * anchor[j];
                    const bool cut = (w < S - 1 && acc >= tot * (w + 1) / S && j + 1 - k0 >= 4) || j == n - 1;
                    if (cut) {
                        auto c = cost_of(k0, j + 1);
                        double t = 0; for (double x : c) t += x + ovh;
                        mx = std::max(mx, t);
                        k0 = j + 1; ++w;
                    }
                }
                worst[bi] = std::max(worst[bi], mx); sum_max[bi] += mx; all[bi].push_back(mx);
            }
            ++nwg;
        }
    }
    printf("%ld workgroups; a column with a seed inside the chunk costs %.2f passes, another %.2f (%.0f%% of the columns are such)\n", nwg, sa / std::max(1.0, ca), sn / std::max(1.0, cn), 100 * ca / (ca + cn));
    for (int bi = 0; bi < NB; ++bi) {
        std::sort(all[bi].begin(), all[bi].end());
        printf("beta %.0f: slowest wave of a workgroup: mean %.1f  p99 %.1f  max %.1f\n", betas[bi], sum_max[bi] / nwg, all[bi][(size_t)(all[bi].size() * 0.99)], worst[bi]);
    }
    return 0;
}
