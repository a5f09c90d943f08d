// k2_sim.cpp -- CPU statistics for the L2 sweep's row splitting (diagnostic, not product code).
// Simulates the reference's second 1-D pass (imgproc.h:91-130) per image row in float, records the stack before every
// column, and evaluates how often a speculative start state for a segment is the real one.
//   input: binary file {int32 m, W, H; uint8 seed[m][W][H]}  (seed = 1 where the first-stage image is 0)
//   usage: k2_sim <file> <K seeded columns per secondary segment> <L lookback columns> [S primary segments]
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

// process column q on stack st (bottom index lo never popped); returns lowest index tested against
static inline int step(std::vector<Ent>& st, int lo, int q, float fq) {
    int deepest = (int)st.size() - 1;
    while (true) {
        const int k = (int)st.size() - 1;
        deepest = std::min(deepest, k);
        const float s = isect(fq, q, st[k].f, st[k].v);
        if (s > st[k].z || k == lo) { st.push_back(Ent{q, fq, k == lo && !(s > st[k].z) ? s : s}); break; }
        st.pop_back();
    }
    return deepest;
}

int main(int argc, char** argv) {
    if (argc < 4) return 1;
    FILE* fp = fopen(argv[1], "rb");
    int32_t hdr[3];
    if (!fp || fread(hdr, 4, 3, fp) != 3) return 1;
    const int m = hdr[0], W = hdr[1], H = hdr[2];
    const int K = atoi(argv[2]), L = atoi(argv[3]), S = argc > 4 ? atoi(argv[4]) : 4;
    std::vector<uint8_t> seed((size_t)W * H);
    const float FMAX = 3.402823466e38f;
    long n_seg = 0, n_ok = 0, n_bad_guess = 0, n_bad_depth = 0;
    long n_wseg = 0, n_wok = 0;  // per 64-row chunk: a segment validates only if it does for all 64 rows
    long tot_cols = 0, tot_lookback = 0;
    long chain_now = 0, chain_new = 0;  // critical path (columns) per chunk: now (max primary segment) vs new (max secondary + lookback), summed over chunks
    long max_now = 0, max_new = 0;
    std::vector<long> dist_hist(64, 0);
    for (int k = 0; k < m; ++k) {
        if (fread(seed.data(), 1, seed.size(), fp) != seed.size()) return 2;
        // seeded columns of the slice
        std::vector<int> cols;
        for (int x = 0; x < W; ++x) {
            bool any = false;
            for (int y = 0; y < H && !any; ++y) any = seed[(size_t)x * H + y];
            if (any) cols.push_back(x);
        }
        const int n = (int)cols.size();
        if (n < 2 * S) continue;
        // pass 1: squared vertical distance to the nearest seed of the column
        std::vector<float> f((size_t)n * H);
        for (int j = 0; j < n; ++j) {
            const uint8_t* c = &seed[(size_t)cols[j] * H];
            int last = -1 << 20;
            for (int y = 0; y < H; ++y) { if (c[y]) last = y; f[(size_t)j * H + y] = (float)(y - last); }
            int nxt = 1 << 20;
            for (int y = H - 1; y >= 0; --y) { if (c[y]) nxt = y; float d = std::min(f[(size_t)j * H + y], (float)(nxt - y)); f[(size_t)j * H + y] = d * d; }
        }
        for (int c0 = 0; c0 < H; c0 += 64) {
            // per secondary segment index -> all rows ok?
            std::vector<std::vector<char>> okmap;  // [row][segment]
            long worst_now = 0, worst_new = 0;
            for (int y = c0; y < std::min(H, c0 + 64); ++y) {
                // ---- the real run over the seeded columns, keeping the stack before every column
                std::vector<Ent> st;
                std::vector<std::vector<Ent>> before((size_t)n + 1);
                st.push_back(Ent{cols[0], f[y], -INFINITY});
                // column 0 of the image, if seedless, is the real bottom (f = FLT_MAX); it matters only for all-seedless rows: ignored here
                for (int j = 1; j < n; ++j) {
                    before[j] = st;
                    step(st, 0, cols[j], f[(size_t)j * H + y]);
                }
                before[n] = st;
                // final vertices (owners): entries of the final stack; primary junctions: owner of the quantile pixel
                std::vector<int> prim;  // indices j (into cols) of primary bottoms, prim[0] = 0
                prim.push_back(0);
                for (int w = 1; w < S; ++w) {
                    const int xq = cols[(int)((long)n * w / S)];
                    // owner of pixel xq: last final entry with z < xq
                    int own = st[0].v;
                    for (auto& e : st) if (e.z < (float)xq) own = e.v;
                    const int j = (int)(std::lower_bound(cols.begin(), cols.end(), own) - cols.begin());
                    if (j > prim.back()) prim.push_back(j);
                }
                prim.push_back(n - 1 + 1);  // sentinel: one past the last column index
                std::vector<char> rowok;
                long longest_now = 0, longest_new = 0;
                for (size_t p = 0; p + 1 < prim.size(); ++p) {
                    const int jb = prim[p], je = prim[p + 1];  // primary segment: bottom column jb, columns (jb, je) .. plus je itself if not sentinel
                    const int jend = std::min(je, n - 1);
                    longest_now = std::max<long>(longest_now, jend - jb);
                    // secondary splits every K columns
                    for (int js = jb; js < jend; js += K) {
                        const int jl = std::min(js + K, jend);  // segment processes columns (js, jl]
                        longest_new = std::max<long>(longest_new, (jl - js) + (js == jb ? 0 : std::min(L, js - jb)));
                        if (js == jb) continue;  // the first secondary segment starts on the primary bottom itself: exact
                        // lookback run: stack [v_b], columns (max(jb, js - L), js]
                        std::vector<Ent> loc;
                        loc.push_back(Ent{cols[jb], f[(size_t)jb * H + y], -INFINITY});
                        const int j0 = std::max(jb, js - L);
                        for (int j = j0 + 1; j <= js; ++j) step(loc, 0, cols[j], f[(size_t)j * H + y]);
                        tot_lookback += js - j0;
                        const int init_sz = (int)loc.size();
                        std::vector<Ent> init = loc;
                        int deepest = init_sz - 1;
                        for (int j = js + 1; j <= jl; ++j) deepest = std::min(deepest, step(loc, 0, cols[j], f[(size_t)j * H + y]));
                        tot_cols += jl - js;
                        // real state before column js + 1
                        const std::vector<Ent>& real = before[js + 1];
                        // entries init[deepest - 1 .. init_sz - 1] must be the top of the real stack (deepest - 1: so that init[deepest].z is real)
                        const int need_from = std::max(0, deepest - 1);
                        const int cnt = init_sz - need_from;
                        bool ok = cnt <= (int)real.size();
                        for (int i = 0; ok && i < cnt; ++i) ok = init[init_sz - 1 - i].v == real[real.size() - 1 - i].v;
                        // (if the chain reaches init[0] = the primary bottom, its z is -inf locally: fine only if it is never popped, which holds for a final vertex)
                        ++n_seg;
                        if (ok) ++n_ok; else {
                            // why: is the real second-to-top outside {v} u lookback?
                            const int a = real.size() >= 2 ? real[real.size() - 2].v : -1;
                            if (!(a == cols[jb] || a >= cols[j0 + 1 > js ? js : j0 + 1])) ++n_bad_guess; else ++n_bad_depth;
                        }
                        rowok.push_back(ok);
                        // distance (in seeded columns) from c to the real second-to-top
                        if (real.size() >= 2) {
                            const int a = real[real.size() - 2].v;
                            const int ja = (int)(std::lower_bound(cols.begin(), cols.end(), a) - cols.begin());
                            int d = js - ja, b = 0;
                            while (d > 1 && b < 63) { d >>= 1; ++b; }
                            if (a == cols[jb]) b = 62;  // the primary bottom
                            ++dist_hist[b];
                        }
                    }
                }
                okmap.push_back(rowok);
                worst_now = std::max(worst_now, longest_now);
                worst_new = std::max(worst_new, longest_new);
            }
            chain_now += worst_now; chain_new += worst_new;
            max_now = std::max(max_now, worst_now); max_new = std::max(max_new, worst_new);
            // chunk-level: rows have different segment counts (junctions differ per row); approximate by position index
            size_t mins = (size_t)-1;
            for (auto& r : okmap) mins = std::min(mins, r.size());
            if (mins != (size_t)-1)
                for (size_t sidx = 0; sidx < mins; ++sidx) {
                    bool all = true;
                    for (auto& r : okmap) all = all && r[sidx];
                    ++n_wseg; n_wok += all;
                }
        }
        fprintf(stderr, "slice %d: seeded %d, segments so far %ld ok %.4f\n", k, n, n_seg, n_seg ? (double)n_ok / n_seg : 0.0);
    }
    printf("K=%d L=%d S=%d: secondary segments %ld, valid %.5f (bad guess %ld, bad depth %ld); chunk-level %ld valid %.4f\n", K, L, S, n_seg,
           (double)n_ok / std::max(1L, n_seg), n_bad_guess, n_bad_depth, n_wseg, (double)n_wok / std::max(1L, n_wseg));
    printf("  work: columns %ld + lookback %ld (%.1f %%); critical path per chunk (columns): now avg %.1f max %ld -> new avg %.1f max %ld\n", tot_cols,
           tot_lookback, 100.0 * tot_lookback / std::max(1L, tot_cols), (double)chain_now / (m * ((H + 63) / 64)), max_now,
           (double)chain_new / (m * ((H + 63) / 64)), max_new);
    printf("  log2 distance (seeded columns) from c to the real second-to-top (62 = the primary bottom):");
    for (int b = 0; b < 64; ++b) if (dist_hist[b]) printf(" [%d]=%ld", b, dist_hist[b]);
    printf("\n");
    return 0;
}
