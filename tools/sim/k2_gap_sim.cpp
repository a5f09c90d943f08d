// k2_gap_sim.cpp -- CPU statistics: how much of the L2 sweep's per-row chain sits in runs of "far" columns (no seed
// inside the 64-row chunk) that could be skipped under a float-safe sufficient condition (diagnostic, not product code).
//   gap = maximal run of >= Gmin consecutive seeded columns that are far for the chunk, followed by a column r.
//   With t = the stack top before the gap:  (A) m_A = min_g s(g, t) > z_t   (no gap column can pop t)
//                                            (B) max_g s(r, g) <= m_A        (r pops every gap column it meets)
//   => processing r directly on the pre-gap stack gives the state the sequential run reaches after the gap and r.
//   usage: k2_gap_sim <seed file of make_seeds.py> [Gmin=4] [cap=0 (max gap length, 0 = unlimited)] [speedup=8]
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
static inline void step(std::vector<Ent>& st, int q, float fq) {
    while (true) {
        const int k = (int)st.size() - 1;
        const float s = isect(fq, q, st[k].f, st[k].v);
        if (s > st[k].z || k == 0) { st.push_back(Ent{q, fq, s}); break; }
        st.pop_back();
    }
}
static bool same(const std::vector<Ent>& a, const std::vector<Ent>& b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i) if (a[i].v != b[i].v || !(a[i].z == b[i].z || (i == 0))) return false;
    return true;
}
int main(int argc, char** argv) {
    FILE* fp = fopen(argv[1], "rb");
    int32_t hdr[3];
    if (!fp || fread(hdr, 4, 3, fp) != 3) return 1;
    const int m = hdr[0], W = hdr[1], H = hdr[2];
    const int Gmin = argc > 2 ? atoi(argv[2]) : 4, cap = argc > 3 ? atoi(argv[3]) : 0;
    const double speed = argc > 4 ? atof(argv[4]) : 8.0;
    std::vector<uint8_t> seed((size_t)W * H);
    long cols_total = 0, cols_in_gaps = 0, cols_skippable_rows = 0, cols_skippable_wave = 0, unsound = 0, truth_rows = 0;
    double path_now_sum = 0, path_new_sum = 0, path_now_max = 0, path_new_max = 0;
    long nchunks = 0;
    for (int k = 0; k < m; ++k) {
        if (fread(seed.data(), 1, seed.size(), fp) != seed.size()) return 2;
        std::vector<int> cols;
        for (int x = 0; x < W; ++x) { bool any = false; for (int y = 0; y < H && !any; ++y) any = seed[(size_t)x * H + y]; if (any) cols.push_back(x); }
        const int n = (int)cols.size();
        if (n < 2) continue;
        std::vector<float> f((size_t)n * H);
        for (int j = 0; j < n; ++j) {
            const uint8_t* c = &seed[(size_t)cols[j] * H];
            int last = -(1 << 20);
            for (int y = 0; y < H; ++y) { if (c[y]) last = y; f[(size_t)j * H + y] = (float)(y - last); }
            int nxt = 1 << 20;
            for (int y = H - 1; y >= 0; --y) { if (c[y]) nxt = y; float d = std::min(f[(size_t)j * H + y], (float)(nxt - y)); f[(size_t)j * H + y] = d * d; }
        }
        for (int c0 = 0; c0 < H; c0 += 64, ++nchunks) {
            const int c1 = std::min(H, c0 + 64);
            std::vector<char> far((size_t)n);
            for (int j = 0; j < n; ++j) { bool in = false; for (int y = c0; y < c1 && !in; ++y) in = seed[(size_t)cols[j] * H + y]; far[j] = !in; }
            // per row: the real run; the final stack's adjacent vertex pairs (v_i, v_{i+1}) with >= Gmin seeded columns
            // between them are the no-vertex gaps; a gap is skippable when (A) and (B) hold with the real z(v_i)
            std::vector<int> inside_cnt((size_t)n, 0), inside_ok((size_t)n, 0);
            for (int y = c0; y < c1; ++y) {
                std::vector<Ent> st;
                st.push_back(Ent{cols[0], f[y], -INFINITY});
                for (int j = 1; j < n; ++j) step(st, cols[j], f[(size_t)j * H + y]);
                for (size_t i = 0; i + 1 < st.size(); ++i) {
                    const int ja = (int)(std::lower_bound(cols.begin(), cols.end(), st[i].v) - cols.begin());
                    const int jb = (int)(std::lower_bound(cols.begin(), cols.end(), st[i + 1].v) - cols.begin());
                    if (jb - ja - 1 < Gmin) continue;
                    float mA = INFINITY, MB = -INFINITY;
                    for (int g = ja + 1; g < jb; ++g) {
                        const float fg = f[(size_t)g * H + y];
                        mA = std::min(mA, isect(fg, cols[g], st[i].f, st[i].v));
                        MB = std::max(MB, isect(st[i + 1].f, st[i + 1].v, fg, cols[g]));
                    }
                    const bool ok = (mA > st[i].z) && MB <= mA;
                    for (int g = ja + 1; g < jb; ++g) { ++inside_cnt[g]; if (ok) ++inside_ok[g]; }
                }
            }
            long ingap = 0, skipw = 0;
            const int rows = c1 - c0;
            for (int j = 0; j < n; ++j) {
                ingap += inside_cnt[j] == rows;          // every row of the chunk is inside a no-vertex gap at column j
                skipw += inside_ok[j] == rows;           // ... and every one of those gaps passes (A) and (B)
                cols_skippable_rows += inside_ok[j];
                truth_rows += inside_cnt[j];
            }
            cols_total += n; cols_in_gaps += ingap; cols_skippable_wave += skipw;
            const double now = n, nw = n - skipw + skipw / speed;
            path_now_sum += now; path_new_sum += nw; path_now_max = std::max(path_now_max, now); path_new_max = std::max(path_new_max, nw);
        }
        fprintf(stderr, "slice %d (%d seeded)\n", k, n);
    }
    printf("Gmin=%d cap=%d: columns %ld (per chunk row), in candidate gaps %.1f %%, skippable by whole waves %.1f %% (row-level: %.1f %% of 64x, truth %.1f %%), unsound %ld\n",
           Gmin, cap, cols_total, 100.0 * cols_in_gaps / cols_total, 100.0 * cols_skippable_wave / cols_total,
           100.0 * cols_skippable_rows / (64.0 * cols_total), 100.0 * truth_rows / (64.0 * cols_total), unsound);
    printf("  whole-row chain per chunk (columns, S=1): now avg %.1f max %.0f -> with gaps at 1/%.0f cost: avg %.1f max %.1f\n",
           path_now_sum / nchunks, path_now_max, speed, path_new_sum / nchunks, path_new_max);
    return 0;
}
