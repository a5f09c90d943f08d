// balanced_sim.cpp -- CPU statistics (diagnostic, not product code) for the balanced form of the L2 sweep: a row's seeded
// columns are cut into S ranges of equal count (no junction search), every range runs the reference's construction on
// its own stack (bottom = its first column, z = -inf), and the ranges' stacks are merged from left to right by landing
// the next range's entries on the accumulated stack until one of them stays on its local predecessor (that IS the
// reference's run on the union of the local stacks; see exact_owner_sim.cpp for why the pixel owners are the reference's).
//
// Model of a wave (lane = row): a column costs the maximum over the wave's rows of the number of tests it takes.
//   usage: balanced_sim <seed file of make_seeds.py> <S (0: ceil(n / cpw) per slice)> [cpw=64] [check=0]
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
        if (v.empty()) { printf("%-28s (none)\n", name); return; }
        std::sort(v.begin(), v.end());
        double sum = 0; for (long x : v) sum += (double)x;
        printf("%-28s n %8zu  mean %9.1f  p50 %6ld  p90 %6ld  p99 %6ld  max %6ld  sum %.3e\n", name, v.size(), sum / (double)v.size(), v[v.size() / 2],
               v[(size_t)((double)v.size() * 0.9)], v[(size_t)((double)v.size() * 0.99)], v.back(), sum);
    }
};
int main(int argc, char** argv) {
    FILE* fp = fopen(argv[1], "rb");
    int32_t hdr[3];
    if (!fp || fread(hdr, 4, 3, fp) != 3) return 1;
    const int m = hdr[0], W = hdr[1], H = hdr[2];
    const int Sarg = argc > 2 ? atoi(argv[2]) : 8, cpw = argc > 3 ? atoi(argv[3]) : 64, check = argc > 4 ? atoi(argv[4]) : 0, prune = argc > 5 ? atoi(argv[5]) : 0;
    Dist kept_frac;
    std::vector<uint8_t> seed((size_t)W * H);
    Dist cur_adv, m_pops;
    Dist wave_passes, wave_cols, block_maxpass, depth_max, local_n, merge_tests_j, merge_tests_blk, valid_n, own_seg, own_row, segs, own_blk_maxseg, lit_passes;
    long wrong = 0, rows = 0;
    for (int k = 0; k < m; ++k) {
        if (fread(seed.data(), 1, seed.size(), fp) != seed.size()) return 2;
        std::vector<int> cols;
        for (int x = 0; x < W; ++x) { bool any = false; for (int y = 0; y < H && !any; ++y) any = seed[(size_t)x * H + y]; if (any) cols.push_back(x); }
        const int n = (int)cols.size();
        if (n == 0) continue;
        const int S = Sarg > 0 ? std::min(Sarg, n) : std::max(1, std::min(16, (n + cpw - 1) / cpw));
        segs.add(S);
        std::vector<float> f((size_t)n * H);
        for (int j = 0; j < n; ++j) {
            const uint8_t* c = &seed[(size_t)cols[j] * H];
            int last = -(1 << 20);
            for (int y = 0; y < H; ++y) { if (c[y]) last = y; f[(size_t)j * H + y] = (float)(y - last); }
            int nxt = 1 << 20;
            for (int y = H - 1; y >= 0; --y) { if (c[y]) nxt = y; float d = std::min(f[(size_t)j * H + y], (float)(nxt - y)); f[(size_t)j * H + y] = d * d; }
        }
        for (int c0 = 0; c0 < H; c0 += 64) {
            const int R = std::min(64, H - c0);
            // columns kept for this chunk: a column without a seed inside the chunk whose distance to the chunk is lb is pruned when
            // anchors (columns with a seed inside the chunk, f <= 63^2 in every row) a1 < u < a2 exist with lb^2 >= 63^2 + ((a2 - a1) / 2)^2:
            // its point lies on or above the chord of the anchors in every row, so it owns no pixel
            std::vector<int> kept;
            if (prune) {
                std::vector<char> anchor(n, 0);
                std::vector<long> lb(n, 0);
                for (int j = 0; j < n; ++j) {
                    const uint8_t* cc = &seed[(size_t)cols[j] * H];
                    bool in = false; for (int y = c0; y < c0 + R; ++y) in |= cc[y] != 0;
                    anchor[j] = in;
                    if (!in) { long up = 1 << 20, dn = 1 << 20; for (int y = c0 - 1; y >= 0; --y) if (cc[y]) { up = c0 - y; break; } for (int y = c0 + R; y < H; ++y) if (cc[y]) { dn = y - (c0 + R - 1); break; } lb[j] = std::min(up, dn); }
                }
                std::vector<int> prevA(n, -1), nextA(n, -1);
                int last = -1; for (int j = 0; j < n; ++j) { prevA[j] = last; if (anchor[j]) last = j; }
                last = -1; for (int j = n - 1; j >= 0; --j) { nextA[j] = last; if (anchor[j]) last = j; }
                for (int j = 0; j < n; ++j) {
                    bool drop = false;
                    if (!anchor[j] && prevA[j] >= 0 && nextA[j] >= 0) {
                        const long d = cols[nextA[j]] - cols[prevA[j]];
                        drop = 4 * lb[j] * lb[j] >= 4 * 3969 + d * d;
                    }
                    if (!drop) kept.push_back(j);
                }
            } else for (int j = 0; j < n; ++j) kept.push_back(j);
            kept_frac.add((long)(1000.0 * kept.size() / n));
            const int nk = (int)kept.size();
            std::vector<std::vector<std::vector<Ent>>> st(S, std::vector<std::vector<Ent>>(R));  // local stacks [segment][row]
            long blk_max = 0;
            for (int w = 0; w < S; ++w) {
                const int k0 = (int)((long)nk * w / S), k1 = (int)((long)nk * (w + 1) / S);  // kept columns [k0, k1)
                if (k0 >= k1) { wave_passes.add(0); wave_cols.add(0); depth_max.add(0); local_n.add(0); for (int r = 0; r < R; ++r) st[w][r].clear(); continue; }
                const int j0 = kept[k0], j1 = 0; (void)j1;
                long passes = 0, dmax = 0;
                for (int r = 0; r < R; ++r) st[w][r].push_back(Ent{cols[j0], f[(size_t)j0 * H + c0 + r], -INFINITY});
                for (int kk = k0 + 1; kk < k1; ++kk) { const int j = kept[kk];
                    int mx = 0;
                    for (int r = 0; r < R; ++r) {
                        auto& s = st[w][r];
                        const float fq = f[(size_t)j * H + c0 + r];
                        int t = 0;
                        float sv;
                        for (;;) { ++t; sv = isect(fq, cols[j], s.back().f, s.back().v); if (sv > s.back().z) break; s.pop_back(); }
                        s.push_back(Ent{cols[j], fq, sv});
                        mx = std::max(mx, t);
                        dmax = std::max(dmax, (long)s.size());
                    }
                    passes += mx;
                }
                wave_passes.add(passes); wave_cols.add(k1 - k0); depth_max.add(dmax);
                blk_max = std::max(blk_max, passes);
                long ln = 0; for (int r = 0; r < R; ++r) ln = std::max(ln, (long)st[w][r].size());
                local_n.add(ln);
            }
            block_maxpass.add(blk_max);
            // literal run of the whole row in one wave, for comparison (the one-wave-per-chunk kernel's chain)
            {
                std::vector<std::vector<Ent>> s(R);
                for (int r = 0; r < R; ++r) s[r].push_back(Ent{cols[0], f[(size_t)0 * H + c0 + r], -INFINITY});
                long passes = 0;
                for (int j = 1; j < n; ++j) {
                    int mx = 0;
                    for (int r = 0; r < R; ++r) {
                        const float fq = f[(size_t)j * H + c0 + r];
                        int t = 0; float sv;
                        for (;;) { ++t; sv = isect(fq, cols[j], s[r].back().f, s[r].back().v); if (sv > s[r].back().z) break; s[r].pop_back(); }
                        s[r].push_back(Ent{cols[j], fq, sv});
                        mx = std::max(mx, t);
                    }
                    passes += mx;
                }
                lit_passes.add(passes);
                // merge, per row; the merged stack must equal the literal one in the entries that own pixels
                long blk_merge = 0;
                std::vector<long> jt(S, 0);
                std::vector<long> ownmaxseg(S, 0);
                for (int r = 0; r < R; ++r) {
                    int w0 = 0; while (st[w0][r].empty()) ++w0;
                    std::vector<Ent> M = st[w0][r];
                    std::vector<int> segof(M.size(), w0);
                    for (int w = w0 + 1; w < S; ++w) {
                        const auto& B = st[w][r];
                        if (B.empty()) continue;
                        long tests = 0, pops = 0;
                        size_t cur = 0;
                        float zc;
                        for (;;) {
                            for (;;) { ++tests; zc = isect(B[cur].f, B[cur].v, M.back().f, M.back().v); if (zc > M.back().z) break; M.pop_back(); segof.pop_back(); ++pops; }
                            if (cur + 1 < B.size() && B[cur + 1].z <= zc) { ++cur; continue; }
                            break;
                        }
                        M.push_back(Ent{B[cur].v, B[cur].f, zc}); segof.push_back(w);
                        for (size_t i = cur + 1; i < B.size(); ++i) { M.push_back(B[i]); segof.push_back(w); }
                        jt[w] = std::max(jt[w], tests);
                        cur_adv.add((long)cur); m_pops.add(pops);
                    }
                    // owners of the merged stack and of the literal one
                    auto owners = [&](const std::vector<Ent>& s, std::vector<int>* sg, std::vector<long>* cnt) {
                        std::vector<std::pair<int, int>> o;  // (first pixel, column)
                        int last_st = -1;
                        for (size_t i = 0; i < s.size(); ++i) {
                            const int stp = (int)std::floor(std::min(std::max(s[i].z, -1.f), (float)W - 0.5f)) + 1;
                            const int nx = i + 1 < s.size() ? (int)std::floor(std::min(std::max(s[i + 1].z, -1.f), (float)W - 0.5f)) + 1 : W;
                            if (stp < nx && stp > last_st) { o.push_back({stp, s[i].v}); last_st = stp; if (cnt) (*cnt)[(*sg)[i]]++; }
                        }
                        return o;
                    };
                    std::vector<long> cnt(S, 0);
                    auto om = owners(M, &segof, &cnt);
                    if (check) { auto ol = owners(s[r], nullptr, nullptr); if (om != ol) ++wrong; }
                    ++rows;
                    own_row.add((long)om.size());
                    valid_n.add((long)M.size());
                    for (int w = 0; w < S; ++w) ownmaxseg[w] = std::max(ownmaxseg[w], cnt[w]);
                }
                long mo = 0;
                for (int w = 1; w < S; ++w) { merge_tests_j.add(jt[w]); blk_merge += jt[w]; }
                for (int w = 0; w < S; ++w) { own_seg.add(ownmaxseg[w]); mo = std::max(mo, ownmaxseg[w]); }
                merge_tests_blk.add(blk_merge);
                own_blk_maxseg.add(mo);
            }
        }
        fprintf(stderr, "slice %d: %d seeded columns, %d segments\n", k, n, S);
    }
    printf("%d x %d x %d, S = %d (cpw %d)\n", m, W, H, Sarg, cpw);
    kept_frac.print("chunk: kept columns (permille)");
    segs.print("segments per slice");
    lit_passes.print("one wave per chunk: passes");
    wave_passes.print("wave: test passes");
    wave_cols.print("wave: columns");
    block_maxpass.print("block: longest wave");
    depth_max.print("wave: deepest stack");
    local_n.print("wave: local stack (max row)");
    merge_tests_j.print("junction: tests (max row)");
    merge_tests_blk.print("block: merge tests");
    cur_adv.print("(row, junction): entries of the right stack popped");
    m_pops.print("(row, junction): entries of the left stack popped");
    valid_n.print("row: merged stack");
    own_row.print("row: owner entries");
    own_seg.print("segment: owners (max row)");
    own_blk_maxseg.print("block: largest segment list");
    if (check) printf("rows %ld, owner lists differing from the literal run's: %ld\n", rows, wrong);
    return wrong != 0;
}
