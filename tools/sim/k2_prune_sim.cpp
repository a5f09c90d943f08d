// k2_prune_sim.cpp -- CPU statistics for "prune, run, verify" (diagnostic, not product code):
//   1. per row, strong points = minima of b_u = f_u + u^2 over blocks of B seeded columns; their lower convex hull is
//      an upper bound of the true one (the lower envelope of the parabolas = the lower hull of the points (u, b_u));
//   2. columns clearly above that coarse hull are pruned (they cannot be envelope vertices in exact arithmetic);
//   3. the reference's float pass runs on the surviving columns only -> final stack V;
//   4. every other column g, between the final neighbours v_i < g < v_{i+1}, is verified to be without effect in the
//      reference's float run:  (A) s(g, v_i) > z(v_i);  (B) s(v_{i+1}, g) <= min over the gap of s(., v_i);  columns
//      behind the last vertex: (A) and min s(., v_last) >= W - 1.
//   usage: k2_prune_sim <seed file> [B=8]
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
int main(int argc, char** argv) {
    FILE* fp = fopen(argv[1], "rb");
    int32_t hdr[3];
    if (!fp || fread(hdr, 4, 3, fp) != 3) return 1;
    const int m = hdr[0], W = hdr[1], H = hdr[2];
    const int B = argc > 2 ? atoi(argv[2]) : 8;
    const int uniform = argc > 3 ? atoi(argv[3]) : 0;
    const int statS = argc > 4 ? atoi(argv[4]) : 5;  // primary segments per row for the longest-segment statistic (modes other than 3)  // 1: one prune decision per (chunk, column) from the value range over the chunk's rows
    std::vector<uint8_t> seed((size_t)W * H);
    long rows = 0, rows_fail = 0, rows_diff = 0, chunks = 0, chunks_fail = 0;
    long cols_total = 0, surv_rows = 0, surv_wave = 0;
    double path_now = 0, path_new = 0, pmax_now = 0, pmax_new = 0;
    long seg_now_sum = 0, seg_new_sum = 0, seg_now_max = 0, seg_new_max = 0;
    long la_dead = 0, la_mat = 0, la_unsound = 0, la_wave_proc = 0, la_wave_mat = 0;
    long b_fail1 = 0, b_fail2 = 0, b_fail3 = 0, b_fail23 = 0, b_runs = 0;
    long inloop_rows_fail = 0, inloop_failA = 0, inloop_failB = 0, inloop_unsound = 0, inloop_wave_keep = 0;
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
        for (int c0 = 0; c0 < H; c0 += 64, ++chunks) {
            const int c1 = std::min(H, c0 + 64);
            std::vector<int> keep_cnt((size_t)n, 0), keep2_cnt((size_t)n, 0), la_mat_col((size_t)n, 0), la_proc_col((size_t)n, 0);
            bool chunk_fail = false;
            std::vector<char> ukeep((size_t)n, 1);
            std::vector<std::vector<int>> seg_all(16, std::vector<int>((size_t)n, 0)), seg_keep(16, std::vector<int>((size_t)n, 0));
            if (uniform) {
                std::vector<double> lo((size_t)n), hi((size_t)n);
                for (int j = 0; j < n; ++j) {
                    double l = 1e300, h = -1;
                    for (int y = c0; y < c1; ++y) { l = std::min<double>(l, f[(size_t)j * H + y]); h = std::max<double>(h, f[(size_t)j * H + y]); }
                    lo[j] = l + (double)cols[j] * cols[j]; hi[j] = h + (double)cols[j] * cols[j];
                }
                std::vector<int> strong;
                strong.push_back(0);
                for (int j0 = 0; j0 < n; j0 += B) {
                    int best = j0;
                    for (int j = j0; j < std::min(n, j0 + B); ++j) if (hi[j] < hi[best]) best = j;
                    if (best != strong.back()) strong.push_back(best);
                }
                if (strong.back() != n - 1) strong.push_back(n - 1);
                std::sort(strong.begin(), strong.end());
                strong.erase(std::unique(strong.begin(), strong.end()), strong.end());
                std::vector<int> hull;
                for (int j : strong) {
                    while (hull.size() >= 2) {
                        const int a = hull[hull.size() - 2], b = hull.back();
                        const double cross = (hi[b] - hi[a]) * (cols[j] - cols[a]) - (hi[j] - hi[a]) * (cols[b] - cols[a]);
                        if (cross >= 0) hull.pop_back(); else break;
                    }
                    hull.push_back(j);
                }
                size_t hp = 0;
                for (int j = 0; j < n; ++j) {
                    while (hp + 1 < hull.size() && hull[hp + 1] <= j) ++hp;
                    if (hull[hp] == j || hp + 1 >= hull.size()) continue;
                    const int a = hull[hp], c = hull[hp + 1];
                    const double chord = hi[a] + (hi[c] - hi[a]) * (cols[j] - cols[a]) / (double)(cols[c] - cols[a]);
                    ukeep[j] = !(lo[j] > chord + 1e-4 * std::fabs(chord) + 0.5);
                }
                ukeep[0] = 1;
            }
            for (int y = c0; y < c1; ++y, ++rows) {
                auto F = [&](int j) { return f[(size_t)j * H + y]; };
                auto Bv = [&](int j) { return (double)F(j) + (double)cols[j] * cols[j]; };
                // strong points
                std::vector<int> strong;
                strong.push_back(0);
                for (int j0 = 0; j0 < n; j0 += B) {
                    int best = j0;
                    for (int j = j0; j < std::min(n, j0 + B); ++j) if (Bv(j) < Bv(best)) best = j;
                    if (best != strong.back()) strong.push_back(best);
                }
                if (strong.back() != n - 1) strong.push_back(n - 1);
                std::sort(strong.begin(), strong.end());
                strong.erase(std::unique(strong.begin(), strong.end()), strong.end());
                // lower hull of the strong points
                std::vector<int> hull;
                for (int j : strong) {
                    while (hull.size() >= 2) {
                        const int a = hull[hull.size() - 2], b = hull.back();
                        // b above or on segment a-j ?
                        const double cross = (Bv(b) - Bv(a)) * (cols[j] - cols[a]) - (Bv(j) - Bv(a)) * (cols[b] - cols[a]);
                        if (cross >= 0) hull.pop_back(); else break;
                    }
                    hull.push_back(j);
                }
                // prune
                std::vector<char> keep((size_t)n, 0);
                std::vector<double> chordv((size_t)n, 0.0);
                std::vector<int> edge_id((size_t)n, -1);
                size_t hp = 0;
                for (int j = 0; j < n; ++j) {
                    while (hp + 1 < hull.size() && hull[hp + 1] <= j) ++hp;
                    if (hull[hp] == j || hp + 1 >= hull.size()) { keep[j] = 1; continue; }
                    const int a = hull[hp], c = hull[hp + 1];
                    const double chord = Bv(a) + (Bv(c) - Bv(a)) * (cols[j] - cols[a]) / (double)(cols[c] - cols[a]);
                    keep[j] = !(Bv(j) > chord + 1e-4 * std::fabs(chord) + 0.5);
                    chordv[j] = chord; edge_id[j] = (int)hp;
                }
                keep[0] = 1;
                if (uniform == 1) for (int j = 0; j < n; ++j) keep[j] = ukeep[j];
                if (uniform == 3) {
                    // the true run gives the final vertices; primary junctions = owners of the j / S quantile pixels of the
                    // seeded columns (what k_sweep's phase A finds), plus the owner of the row's last pixel
                    std::vector<Ent> tr0;
                    tr0.push_back(Ent{cols[0], F(0), -INFINITY});
                    for (int j = 1; j < n; ++j) step(tr0, cols[j], F(j));
                    const int S = B;  // (the second argument is the segment count in this mode)
                    std::vector<int> jun;
                    jun.push_back(0);
                    auto owner_of = [&](int xq) { int own = tr0[0].v; for (auto& e : tr0) if (e.z < (float)xq) own = e.v; return (int)(std::lower_bound(cols.begin(), cols.end(), own) - cols.begin()); };
                    for (int w = 1; w < S; ++w) { const int j = owner_of(cols[(int)((long)n * w / S)]); if (j > jun.back()) jun.push_back(j); }
                    { const int j = owner_of(W - 1); if (j > jun.back()) jun.push_back(j); }
                    for (int j = 0; j < n; ++j) keep[j] = 1;
                    for (size_t p = 0; p + 1 < jun.size(); ++p) {
                        const int a = jun[p], c = jun[p + 1];
                        for (int j = a + 1; j < c; ++j) {
                            const double chord = Bv(a) + (Bv(c) - Bv(a)) * (cols[j] - cols[a]) / (double)(cols[c] - cols[a]);
                            keep[j] = !(Bv(j) > chord + 1e-4 * std::fabs(chord) + 0.5);
                        }
                    }
                    keep[0] = 1;
                    for (size_t p = 0; p < jun.size(); ++p) {  // segment p = columns (jun[p], jun[p + 1]] (the last one: to the end)
                        const int a = jun[p], c = p + 1 < jun.size() ? jun[p + 1] : n - 1;
                        for (int j = a + 1; j <= c; ++j) { seg_all[p][j] = 1; if (keep[j]) seg_keep[p][j] = 1; }
                    }
                }
                if (uniform == 2) {
                    // block minima, then for a column of block k: a = lowest point of blocks < k, c = lowest point of blocks > k
                    const int nb = (n + B - 1) / B;
                    std::vector<int> bmin((size_t)nb), pre((size_t)nb), suf((size_t)nb);
                    for (int k2 = 0; k2 < nb; ++k2) { int best = k2 * B; for (int j = k2 * B; j < std::min(n, k2 * B + B); ++j) if (Bv(j) < Bv(best)) best = j; bmin[k2] = best; }
                    pre[0] = bmin[0];
                    for (int k2 = 1; k2 < nb; ++k2) pre[k2] = Bv(bmin[k2]) < Bv(pre[k2 - 1]) ? bmin[k2] : pre[k2 - 1];
                    suf[nb - 1] = bmin[nb - 1];
                    for (int k2 = nb - 2; k2 >= 0; --k2) suf[k2] = Bv(bmin[k2]) <= Bv(suf[k2 + 1]) ? bmin[k2] : suf[k2 + 1];
                    for (int j = 0; j < n; ++j) {
                        const int k2 = j / B;
                        keep[j] = 1;
                        if (k2 == 0 || k2 == nb - 1) continue;
                        const int a = pre[k2 - 1], c = suf[k2 + 1];
                        const double chord = Bv(a) + (Bv(c) - Bv(a)) * (cols[j] - cols[a]) / (double)(cols[c] - cols[a]);
                        keep[j] = !(Bv(j) > chord + 1e-4 * std::fabs(chord) + 0.5);
                    }
                    keep[0] = 1;
                }
                if (uniform != 3) {  // longest-segment statistic with primary junctions like k_sweep's (owners of quantile pixels)
                    std::vector<Ent> tr0;
                    tr0.push_back(Ent{cols[0], F(0), -INFINITY});
                    for (int j = 1; j < n; ++j) step(tr0, cols[j], F(j));
                    std::vector<int> jun;
                    jun.push_back(0);
                    auto owner_of = [&](int xq) { int own = tr0[0].v; for (auto& e : tr0) if (e.z < (float)xq) own = e.v; return (int)(std::lower_bound(cols.begin(), cols.end(), own) - cols.begin()); };
                    for (int w = 1; w < statS; ++w) { const int j = owner_of(cols[(int)((long)n * w / statS)]); if (j > jun.back()) jun.push_back(j); }
                    for (size_t p = 0; p < jun.size(); ++p) {
                        const int a = jun[p], c = p + 1 < jun.size() ? jun[p + 1] : n - 1;
                        for (int j = a + 1; j <= c; ++j) { seg_all[p][j] = 1; if (keep[j]) seg_keep[p][j] = 1; }
                    }
                }
                {   // the in-loop rule: a column above the chord is skipped only if it would not pop the top at that time
                    // (else it is processed like any other); a run of skipped columns is certified (B) by the next
                    // processed column
                    std::vector<Ent> st2;
                    st2.push_back(Ent{cols[0], F(0), -INFINITY});
                    bool okrow = true;
                    float mA = INFINITY;
                    std::vector<int> run;
                    for (int j = 1; j < n; ++j) {
                        bool skip = !keep[j];
                        if (skip) {
                            const Ent& t = st2.back();
                            const float sg = isect(F(j), cols[j], t.f, t.v);
                            if (st2.size() > 1 && !(sg > t.z)) { skip = false; ++inloop_failA; }  // processed after all
                            else { mA = std::min(mA, sg); run.push_back(j); }
                        }
                        if (!skip) {
                            if (!run.empty() && uniform == 0) {
                                // the GPU's bounds for max_g s(r, g): chord based, chord + minimal excess, minimal b
                                const int gf = run.front(), gl = run.back();
                                double dmin = 1e300, bmn = 1e300;
                                for (int g : run) { dmin = std::min(dmin, Bv(g) - chordv[g]); bmn = std::min(bmn, Bv(g)); }
                                auto lin = [&](int g) { return chordv[gf] + (chordv[gl] - chordv[gf]) * (gl == gf ? 0.0 : (double)(cols[g] - cols[gf]) / (cols[gl] - cols[gf])); };
                                const double br = Bv(j);
                                const double phi1 = std::max((br - lin(gf)) / (2.0 * (cols[j] - cols[gf])), (br - lin(gl)) / (2.0 * (cols[j] - cols[gl])));
                                const double phi2 = std::max((br - lin(gf) - dmin) / (2.0 * (cols[j] - cols[gf])), (br - lin(gl) - dmin) / (2.0 * (cols[j] - cols[gl])));
                                const double num = br - bmn;
                                const double phi3 = num >= 0 ? num / (2.0 * (cols[j] - cols[gl])) : num / (2.0 * (cols[j] - cols[gf]));
                                if (!(phi1 <= mA)) ++b_fail1;
                                if (!(phi2 <= mA)) ++b_fail2;
                                if (!(phi3 <= mA)) ++b_fail3;
                                if (!(std::min(phi2, phi3) <= mA)) ++b_fail23;
                                ++b_runs;
                            }
                            for (int g : run) if (!(isect(F(j), cols[j], F(g), cols[g]) <= mA)) { okrow = false; ++inloop_failB; }
                            run.clear(); mA = INFINITY;
                            step(st2, cols[j], F(j));
                            keep2_cnt[j] += 1;
                        }
                    }
                    if (!run.empty()) okrow = false;
                    inloop_rows_fail += !okrow;
                    // sanity: same final stack as the true run
                    std::vector<Ent> tr2;
                    tr2.push_back(Ent{cols[0], F(0), -INFINITY});
                    for (int j = 1; j < n; ++j) step(tr2, cols[j], F(j));
                    bool same = tr2.size() == st2.size();
                    for (size_t i = 0; same && i < tr2.size(); ++i) same = tr2[i].v == st2[i].v && (i == 0 || tr2[i].z == st2[i].z);
                    if (okrow && !same) ++inloop_unsound;
                }
                if (uniform == 0) {   // lookahead rule: a candidate (above the chord, does not pop the top) is deferred; the next seeded
                    // column either pops it at once (s(next, g) <= s(g, t): it vanishes) or it is pushed after all
                    std::vector<Ent> st3;
                    st3.push_back(Ent{cols[0], F(0), -INFINITY});
                    int pend = -1; float pend_s = 0.f;
                    for (int j = 1; j < n; ++j) {
                        if (pend >= 0) {
                            const bool dead = isect(F(j), cols[j], F(pend), cols[pend]) <= pend_s;
                            if (dead) { ++la_dead; }
                            else { step(st3, cols[pend], F(pend)); ++la_mat; la_mat_col[j] += 1; }
                            pend = -1;
                        }
                        bool defer = false;
                        if (!keep[j] && j + 1 < n) {
                            const Ent& t = st3.back();
                            const float sg = isect(F(j), cols[j], t.f, t.v);
                            if (st3.size() == 1 || sg > t.z) { defer = true; pend = j; pend_s = sg; }
                        }
                        if (!defer) { step(st3, cols[j], F(j)); la_proc_col[j] += 1; }
                    }
                    if (pend >= 0) step(st3, cols[pend], F(pend));
                    std::vector<Ent> tr3;
                    tr3.push_back(Ent{cols[0], F(0), -INFINITY});
                    for (int j = 1; j < n; ++j) step(tr3, cols[j], F(j));
                    bool same = tr3.size() == st3.size();
                    for (size_t i = 0; same && i < tr3.size(); ++i) same = tr3[i].v == st3[i].v && (i == 0 || tr3[i].z == st3[i].z);
                    if (!same) ++la_unsound;
                }
                // float run on the survivors
                std::vector<Ent> st;
                st.push_back(Ent{cols[0], F(0), -INFINITY});
                for (int j = 1; j < n; ++j) if (keep[j]) step(st, cols[j], F(j));
                // the true run (sanity)
                std::vector<Ent> tr;
                tr.push_back(Ent{cols[0], F(0), -INFINITY});
                for (int j = 1; j < n; ++j) step(tr, cols[j], F(j));
                // verification of every non-final column
                bool ok = true;
                size_t i = 0;
                float mA = INFINITY, MB = -INFINITY;
                for (int j = 1; j < n && ok; ++j) {
                    if (i + 1 < st.size() && st[i + 1].v == cols[j]) {
                        if (mA != INFINITY) ok = ok && (mA > st[i].z) && (MB <= mA);
                        ++i; mA = INFINITY; MB = -INFINITY;
                        continue;
                    }
                    mA = std::min(mA, isect(F(j), cols[j], st[i].f, st[i].v));
                    if (i + 1 < st.size()) MB = std::max(MB, isect(st[i + 1].f, st[i + 1].v, F(j), cols[j]));
                }
                if (ok && mA != INFINITY) ok = (mA > st[i].z) && (i + 1 < st.size() ? false : mA >= (float)(W - 1));
                // does the survivors' stack own the same pixels as the true one?
                auto owners = [&](const std::vector<Ent>& s) {
                    std::vector<std::pair<int, int>> o;  // (first pixel, column)
                    size_t kk = 0;
                    int lastv = -1;
                    for (int q = 0; q < W; ++q) { while (kk + 1 < s.size() && s[kk + 1].z < (float)q) ++kk; if (s[kk].v != lastv) { o.push_back({q, s[kk].v}); lastv = s[kk].v; } }
                    return o;
                };
                const bool same_owners = owners(st) == owners(tr);
                if (!ok) { ++rows_fail; chunk_fail = true; }
                if (ok && !same_owners) ++rows_diff;   // must never happen: verified but different
                for (int j = 0; j < n; ++j) { keep_cnt[j] += keep[j]; surv_rows += keep[j]; }
            }
            {
                long wa = 0, wk = 0;
                for (int p = 0; p < 16; ++p) { long a = 0, kq = 0; for (int j = 0; j < n; ++j) { a += seg_all[p][j]; kq += seg_keep[p][j]; } wa = std::max(wa, a); wk = std::max(wk, kq); }
                seg_now_sum += wa; seg_new_sum += wk; seg_now_max = std::max(seg_now_max, wa); seg_new_max = std::max(seg_new_max, wk);
            }
            for (int j = 0; j < n; ++j) { inloop_wave_keep += keep2_cnt[j] > 0; la_wave_proc += la_proc_col[j] > 0; la_wave_mat += la_mat_col[j] > 0; }
            chunks_fail += chunk_fail;
            long wave_keep = 0;
            for (int j = 0; j < n; ++j) wave_keep += keep_cnt[j] > 0;
            surv_wave += wave_keep; cols_total += n;
            const double now = n, nw = wave_keep + (n - wave_keep) * 0.2 + n * 0.15;  // survivors full cost, pruned 0.2, + prune/verify passes ~0.15 each column
            path_now += now; path_new += nw; pmax_now = std::max(pmax_now, now); pmax_new = std::max(pmax_new, nw);
        }
        fprintf(stderr, "slice %d (%d seeded): rows failing so far %ld of %ld\n", k, n, rows_fail, rows);
    }
    printf("B=%d: survivors row-level %.1f %%, wave-level (kept by any of the 64 rows) %.1f %%; rows failing verification %ld of %ld (chunks %ld of %ld); verified-but-different rows %ld\n",
           B, 100.0 * surv_rows / (64.0 * cols_total), 100.0 * surv_wave / cols_total, rows_fail, rows, chunks_fail, chunks, rows_diff);
    printf("  chain per chunk (column equivalents, S=1): now avg %.1f max %.0f -> new avg %.1f max %.1f\n", path_now / chunks, pmax_now, path_new / chunks, pmax_new);
    printf("  in-loop rule: columns processed after all because they would pop the top %ld; rows failing (B) %ld of %ld (%ld pairs); certified-but-different rows %ld; processed per wave %.1f %%\n", inloop_failA, inloop_rows_fail, rows, inloop_failB, inloop_unsound, 100.0 * inloop_wave_keep / cols_total);
    printf("  lookahead rule: deferred columns that vanish %.1f %% of all (row, column), pushed after all %.1f %%; per wave: columns processed %.1f %%, columns with a late push %.1f %%; rows differing from the true run %ld\n", 100.0 * la_dead / (64.0 * cols_total), 100.0 * la_mat / (64.0 * cols_total), 100.0 * la_wave_proc / cols_total, 100.0 * la_wave_mat / cols_total, la_unsound);
    printf("  bounds for (B) over %ld runs: chord %ld fail, chord + minimal excess %ld, minimal b %ld, better of the last two %ld\n", b_runs, b_fail1, b_fail2, b_fail3, b_fail23);
    printf("  longest wave-level segment per chunk (columns): now avg %.1f max %ld -> processed after chord pruning avg %.1f max %ld\n", (double)seg_now_sum / chunks, seg_now_max, (double)seg_new_sum / chunks, seg_new_max);
    return 0;
}
