// k2_cert_sim.cpp -- CPU statistics (diagnostic, not product code): what the L2 sweep's longest waves would cost if a
// (row, segment) whose two junction columns turn out to be neighbours on the final stack were settled by a certificate
// instead of the literal run.  For a row with the range (cs, ce] of segment w:
//     mA = min over the seeded g in (cs, ce] of s(g, cs)        (also the junction check's minF, taken over MORE tests)
//     mB = max over the seeded g in (cs, ce) of s(ce, g)
//     mB <= mA  =>  the literal run leaves exactly [cs, ce] with z(ce) = s(ce, cs): every entry above cs has z >= mA,
//                   ce pops each of them and is pushed on the bottom.
// Optionally a failing row is split at the owner of a pixel between its junctions and each half tried again (depth D).
//   usage: k2_cert_sim <seed file of make_seeds.py> [S=4] [D=0] [Lmin=0: only rows whose range holds >= Lmin seeded columns try]
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
    const int S = argc > 2 ? atoi(argv[2]) : 4, D = argc > 3 ? atoi(argv[3]) : 0, Lmin = argc > 4 ? atoi(argv[4]) : 0;
    std::vector<uint8_t> seed((size_t)W * H);
    struct WaveStat { long it_now, it_new, cols_now, cols_new, cert_cols; int k, c, w; long it_half[2], it_quart[4], cols_half[2]; double it_mean; };
    std::vector<WaveStat> waves;
    long rows_total = 0, rows_pass = 0, rows_tried = 0, wrong = 0;
    for (int k = 0; k < m; ++k) {
        if (fread(seed.data(), 1, seed.size(), fp) != seed.size()) return 2;
        std::vector<int> cols;
        for (int x = 0; x < W; ++x) { bool any = false; for (int y = 0; y < H && !any; ++y) any = seed[(size_t)x * H + y]; if (any) cols.push_back(x); }
        const int n = (int)cols.size();
        if (n < 2 * S) continue;
        std::vector<float> f((size_t)n * H);
        for (int j = 0; j < n; ++j) {
            const uint8_t* c = &seed[(size_t)cols[j] * H];
            int last = -(1 << 20);
            for (int y = 0; y < H; ++y) { if (c[y]) last = y; f[(size_t)j * H + y] = (float)(y - last); }
            int nxt = 1 << 20;
            for (int y = H - 1; y >= 0; --y) { if (c[y]) nxt = y; float d = std::min(f[(size_t)j * H + y], (float)(nxt - y)); f[(size_t)j * H + y] = d * d; }
        }
        auto owner = [&](int y, int x) {  // index into cols of the owner of pixel x in row y (ties: smaller column)
            float best = INFINITY; int bj = 0;
            for (int j = 0; j < n; ++j) { const int du = cols[j] - x; const float v = f[(size_t)j * H + y] + (float)(du * du); if (v < best) { best = v; bj = j; } }
            return bj;
        };
        for (int c0 = 0; c0 < H; c0 += 64) {
            const int R = std::min(64, H - c0);
            // junction columns per row, as indices into cols; segment w runs over the indices (js, je]
            std::vector<int> cj((size_t)(S + 1) * 64, 0);
            for (int r = 0; r < R; ++r) {
                cj[0 * 64 + r] = -1;  // column 0 / "before the first seeded column"
                for (int w = 1; w < S; ++w) cj[w * 64 + r] = owner(c0 + r, cols[(int)(((long)n * w) / S)]);
                cj[S * 64 + r] = n - 1;
            }
            for (int w = 0; w < S; ++w) {
                WaveStat ws{0, 0, 0, 0, 0, k, c0 / 64, w, {0, 0}, {0, 0, 0, 0}, {0, 0}, 0.0};
                // sub-ranges per row that still need the literal run (after certificates)
                std::vector<std::vector<std::pair<int, int>>> todo(R);
                int lo = n, hi = -1;
                for (int r = 0; r < R; ++r) {
                    int js = cj[w * 64 + r], je = std::max(cj[(w + 1) * 64 + r], js);
                    if (w == 0) js = -1;
                    lo = std::min(lo, js + 1); hi = std::max(hi, je);
                }
                // literal run now: stacks per row, iterations per column = max over the rows
                {
                    std::vector<std::vector<Ent>> st(R);
                    for (int r = 0; r < R; ++r) {
                        const int js = w == 0 ? -1 : cj[w * 64 + r];
                        st[r].push_back(js < 0 ? Ent{0, seed[c0 + r] ? 0.f : 1e30f, -INFINITY} : Ent{cols[js], f[(size_t)js * H + c0 + r], -INFINITY});
                    }
                    for (int j = lo; j <= hi; ++j) {
                        int mx = 0; bool any = false; int mq[4] = {0, 0, 0, 0}; long sum = 0, na = 0;
                        for (int r = 0; r < R; ++r) {
                            const int js = w == 0 ? -1 : cj[w * 64 + r], je = std::max(cj[(w + 1) * 64 + r], js);
                            if (j <= js || j > je) continue;
                            if (w == 0 && cols[j] == 0) continue;
                            any = true;
                            int it = 0;
                            auto& s = st[r];
                            const float fq = f[(size_t)j * H + c0 + r];
                            while (true) {
                                ++it;
                                const int t = (int)s.size() - 1;
                                const float sv = isect(fq, cols[j], s[t].f, s[t].v);
                                if (sv > s[t].z || t == 0) { s.push_back(Ent{cols[j], fq, sv}); break; }
                                s.pop_back();
                            }
                            mx = std::max(mx, it); mq[r / 16] = std::max(mq[r / 16], it); sum += it; ++na;
                        }
                        (void)any;
                        ws.cols_now++; ws.it_now += std::max(mx, 1);
                        for (int h = 0; h < 4; ++h) ws.it_quart[h] += mq[h];
                        ws.it_half[0] += std::max(mq[0], mq[1]); ws.it_half[1] += std::max(mq[2], mq[3]);
                        ws.cols_half[0] += (mq[0] | mq[1]) != 0; ws.cols_half[1] += (mq[2] | mq[3]) != 0;
                        if (na) ws.it_mean += (double)sum / na;
                    }
                    if (const char* dz = getenv("K2_DUMP")) {
                        int dk, dc, dw;
                        if (sscanf(dz, "%d:%d:%d", &dk, &dc, &dw) == 3 && dk == k && dc == c0 / 64 && dw == w) {
                            printf("slice %d chunk %d wave %d: n %d, column index range [%d, %d], junction pixels:", k, dc, w, n, lo, hi);
                            for (int ww = 1; ww < S; ++ww) printf(" %d(idx %d)", cols[(int)(((long)n * ww) / S)], (int)(((long)n * ww) / S));
                            printf("\n");
                            for (int r = 0; r < R; r += 9) {
                                printf("  row %2d: cs idx %d ce idx %d; stack (col idx):", r, cj[w * 64 + r], cj[(w + 1) * 64 + r]);
                                for (const Ent& e : st[r]) { int ix = (int)(std::lower_bound(cols.begin(), cols.end(), e.v) - cols.begin()); printf(" %d", ix); }
                                printf("\n");
                            }
                        }
                    }
                    // certificates (recursive) against the literal stacks
                    for (int r = 0; r < R; ++r) {
                        const int y = c0 + r;
                        int js = w == 0 ? -1 : cj[w * 64 + r];
                        const int je = std::max(cj[(w + 1) * 64 + r], js);
                        ++rows_total;
                        if (js < 0 || je - js < std::max(Lmin, 2)) { if (je > js) todo[r].push_back({js, je}); continue; }
                        ++rows_tried;
                        if (getenv("K2_JUMP")) {
                            const int jmin = atoi(getenv("K2_JUMP"));
                            int a = js, b = je;  // literal run over (a, b] afterwards; b == je means no backward jump
                            const float fa = f[(size_t)js * H + y], fb = f[(size_t)je * H + y];
                            // forward: the next vertex after cs is the column of smallest s(g, cs) (ties: the last one)
                            {
                                float mn = INFINITY; int r1 = -1;
                                for (int j = js + 1; j <= je; ++j) { const float s = isect(f[(size_t)j * H + y], cols[j], fa, cols[js]); if (s <= mn) { mn = s; r1 = j; } }
                                if (r1 - js > jmin) {
                                    float mA = INFINITY, mB = -INFINITY;
                                    const float fr = f[(size_t)r1 * H + y];
                                    for (int j = js + 1; j < r1; ++j) { mA = std::min(mA, isect(f[(size_t)j * H + y], cols[j], fa, cols[js])); mB = std::max(mB, isect(fr, cols[r1], f[(size_t)j * H + y], cols[j])); }
                                    if (mB <= mA) {
                                        a = r1 - 1;  // r1 itself is processed literally (on the bottom)
                                        bool found = false;
                                        for (const Ent& e : st[r]) { if (e.v > cols[js] && e.v < cols[r1]) ++wrong; if (e.v == cols[r1]) found = true; }
                                        if (!found) ++wrong;
                                    }
                                }
                            }
                            // backward: the vertex before ce is the column of largest s(ce, g) (ties: the first one)
                            if (je - a > jmin + 1) {
                                float mx = -INFINITY; int l1 = -1;
                                for (int j = je - 1; j > a; --j) { const float s = isect(fb, cols[je], f[(size_t)j * H + y], cols[j]); if (s >= mx) { mx = s; l1 = j; } }
                                if (l1 >= 0 && je - l1 > jmin) {
                                    float mA = INFINITY, mB = -INFINITY;
                                    const float fl = f[(size_t)l1 * H + y];
                                    for (int j = l1 + 1; j < je; ++j) { mA = std::min(mA, isect(f[(size_t)j * H + y], cols[j], fl, cols[l1])); mB = std::max(mB, isect(fb, cols[je], f[(size_t)j * H + y], cols[j])); }
                                    // (A) needs z(l1) from the literal run; taken from the literal stack here when l1 is on it
                                    float zl = NAN;
                                    for (const Ent& e : st[r]) if (e.v == cols[l1]) zl = e.z;
                                    if (mB <= mA && (l1 == js || mA > zl)) {
                                        b = l1;
                                        for (const Ent& e : st[r]) if (e.v > cols[l1] && e.v < cols[je]) ++wrong;
                                    }
                                }
                            }
                            if (a != js || b != je) ++rows_pass;
                            if (b > a) todo[r].push_back({a, b});
                            if (b != je) todo[r].push_back({je - 1, je});
                            continue;
                        }
                        struct Rg { int a, b, d; };
                        std::vector<Rg> work{{js, je, 0}};
                        bool all = true;
                        while (!work.empty()) {
                            Rg g = work.back(); work.pop_back();
                            if (g.b - g.a <= 1) continue;  // neighbours among the seeded columns: nothing between them
                            const float fa = f[(size_t)g.a * H + y], fb = f[(size_t)g.b * H + y];
                            float mA = INFINITY, mB = -INFINITY;
                            for (int j = g.a + 1; j <= g.b; ++j) mA = std::min(mA, isect(f[(size_t)j * H + y], cols[j], fa, cols[g.a]));
                            for (int j = g.a + 1; j < g.b; ++j) mB = std::max(mB, isect(fb, cols[g.b], f[(size_t)j * H + y], cols[j]));
                            if (mB <= mA) {
                                // check against the literal stack: no entry strictly between a and b may remain
                                for (const Ent& e : st[r]) if (e.v > cols[g.a] && e.v < cols[g.b]) { ++wrong; break; }
                                continue;
                            }
                            if (g.d < D) {
                                const int xm = (cols[g.a] + cols[g.b]) / 2;
                                const int o = owner(y, xm);
                                if (o > g.a && o < g.b) { work.push_back({g.a, o, g.d + 1}); work.push_back({o, g.b, g.d + 1}); continue; }
                            }
                            todo[r].push_back({g.a, g.b}); all = false;
                        }
                        if (all) ++rows_pass;
                    }
                }
                // mode K2_LOOP=T:L:B -- windowed jumps inside the literal loop: after T consecutive columns in which every
                // active row popped, (and at the start of the segment), every row looks L columns ahead: r1 = the column of
                // smallest s(g, top); if the columns before r1 pass (A), (B) the row skips them.  A failed attempt backs off B columns.
                if (const char* lp = getenv("K2_LOOP")) {
                    int T = 4, L = 64, BK = 16;
                    sscanf(lp, "%d:%d:%d", &T, &L, &BK);
                    std::vector<std::vector<Ent>> s2(R);
                    std::vector<int> nq(R), csr(R), cer(R);
                    for (int r = 0; r < R; ++r) {
                        const int js = w == 0 ? -1 : cj[w * 64 + r];
                        csr[r] = js; cer[r] = std::max(cj[(w + 1) * 64 + r], js); nq[r] = js + 1;
                        s2[r].push_back(js < 0 ? Ent{0, seed[c0 + r] ? 0.f : 1e30f, -INFINITY} : Ent{cols[js], f[(size_t)js * H + c0 + r], -INFINITY});
                    }
                    long it = 0, att = 0, natt = 0;
                    int gapc = T, wait = 0;  // an attempt right at the start
                    for (int j = lo; j <= hi;) {
                        if (gapc >= T && wait <= 0) {
                            // attempt on [j, j + L)
                            const int we = std::min(j + L, hi + 1);
                            int skipped_min = 1 << 30; bool anyrow = false;
                            for (int r = 0; r < R; ++r) {
                                const int b0 = std::max(j, nq[r]), e0 = std::min(we, cer[r] + 1);  // this row's window [b0, e0)
                                if (b0 >= e0 || b0 > j) { if (b0 > j || j > cer[r]) continue; }
                                if (b0 >= e0) continue;
                                if (w == 0 && csr[r] < 0 && s2[r].size() == 1 && s2[r][0].f > 1e29f) { anyrow = true; skipped_min = 0; continue; }  // bottom is an unseeded column 0
                                anyrow = true;
                                const Ent t = s2[r].back();
                                const int y = c0 + r;
                                float mn = INFINITY; int r1 = -1;
                                for (int g = b0; g < e0; ++g) { if (w == 0 && cols[g] == 0) continue; const float s = isect(f[(size_t)g * H + y], cols[g], t.f, t.v); if (s <= mn) { mn = s; r1 = g; } }
                                int skip = 0;
                                if (r1 > b0) {
                                    float mA = INFINITY, mB = -INFINITY;
                                    const float fr = f[(size_t)r1 * H + y];
                                    for (int g = b0; g < r1; ++g) { mA = std::min(mA, isect(f[(size_t)g * H + y], cols[g], t.f, t.v)); mB = std::max(mB, isect(fr, cols[r1], f[(size_t)g * H + y], cols[g])); }
                                    if (mB <= mA && (s2[r].size() == 1 || mA > t.z)) { nq[r] = r1; skip = r1 - b0; }
                                }
                                skipped_min = std::min(skipped_min, skip);
                            }
                            att += we - j; ++natt;
                            gapc = 0;
                            if (!anyrow || skipped_min < 2) wait = BK;
                        }
                        int mx = 0; bool any = false, allpop = true;
                        for (int r = 0; r < R; ++r) {
                            if (j <= csr[r] || j > cer[r] || j < nq[r]) continue;
                            if (w == 0 && cols[j] == 0) continue;
                            any = true;
                            int n_it = 0;
                            auto& s = s2[r];
                            const float fq = f[(size_t)j * H + c0 + r];
                            while (true) {
                                ++n_it;
                                const int t = (int)s.size() - 1;
                                const float sv = isect(fq, cols[j], s[t].f, s[t].v);
                                if (sv > s[t].z || t == 0) { s.push_back(Ent{cols[j], fq, sv}); break; }
                                s.pop_back();
                            }
                            mx = std::max(mx, n_it);
                            // a row "looks like a gap" when the column popped something or sits directly on the bottom
                            if (n_it < 2 && s.size() > 2) allpop = false;
                        }
                        if (any) { it += mx; gapc = allpop ? gapc + 1 : 0; --wait; }
                        ++j;
                    }
                    ws.it_new = it; ws.cols_new = att; ws.cert_cols = natt;
                    // the result must be the literal run's (z of the first entry excepted: it is the bottom)
                    for (int r = 0; r < R; ++r) {
                        const int js = w == 0 ? -1 : cj[w * 64 + r];
                        (void)js;
                        // literal stack of this row: recompute
                        std::vector<Ent> s;
                        s.push_back(s2[r][0]);
                        for (int j = csr[r] + 1; j <= cer[r]; ++j) {
                            if (w == 0 && cols[j] == 0) continue;
                            const float fq = f[(size_t)j * H + c0 + r];
                            while (true) {
                                const int t = (int)s.size() - 1;
                                const float sv = isect(fq, cols[j], s[t].f, s[t].v);
                                if (sv > s[t].z || t == 0) { s.push_back(Ent{cols[j], fq, sv}); break; }
                                s.pop_back();
                            }
                        }
                        bool same = s.size() == s2[r].size();
                        for (size_t i = 0; same && i < s.size(); ++i) same = s[i].v == s2[r][i].v && (i == 0 || s[i].z == s2[r][i].z);
                        if (!same) ++wrong;
                    }
                    waves.push_back(ws);
                    continue;
                }
                // literal run on what is left: iterations per column = max over the rows that still hold it
                {
                    int lo2 = n, hi2 = -1;
                    for (int r = 0; r < R; ++r) for (auto& p : todo[r]) { lo2 = std::min(lo2, p.first + 1); hi2 = std::max(hi2, p.second); }
                    std::vector<std::vector<Ent>> st(R);
                    std::vector<int> cur(R, -2);  // bottom of the current sub-range
                    for (int j = lo2; j <= hi2; ++j) {
                        int mx = 0; bool any = false;
                        for (int r = 0; r < R; ++r) {
                            for (auto& p : todo[r]) {
                                if (j <= p.first || j > p.second) continue;
                                if (w == 0 && cols[j] == 0) continue;
                                if (cur[r] != p.first) {
                                    cur[r] = p.first; st[r].clear();
                                    st[r].push_back(p.first < 0 ? Ent{0, seed[c0 + r] ? 0.f : 1e30f, -INFINITY}
                                                                : Ent{cols[p.first], f[(size_t)p.first * H + c0 + r], -INFINITY});
                                }
                                any = true;
                                int it = 0;
                                auto& s = st[r];
                                const float fq = f[(size_t)j * H + c0 + r];
                                while (true) {
                                    ++it;
                                    const int t = (int)s.size() - 1;
                                    const float sv = isect(fq, cols[j], s[t].f, s[t].v);
                                    if (sv > s[t].z || t == 0) { s.push_back(Ent{cols[j], fq, sv}); break; }
                                    s.pop_back();
                                }
                                mx = std::max(mx, it);
                            }
                        }
                        if (any) { ws.cols_new++; ws.it_new += mx; }
                    }
                }
                ws.cert_cols = hi - lo + 1;
                waves.push_back(ws);
            }
        }
    }
    // per block: the longest wave decides
    long nb = (long)waves.size() / S;
    std::vector<long> b_now(nb), b_new(nb), b_cert(nb);
    double s_now = 0, s_new = 0, s_cert = 0;
    for (long b = 0; b < nb; ++b) {
        for (int w = 0; w < S; ++w) {
            const WaveStat& x = waves[b * S + w];
            b_now[b] = std::max(b_now[b], x.it_now); b_new[b] = std::max(b_new[b], x.it_new); b_cert[b] = std::max(b_cert[b], x.cert_cols);
            s_now += x.it_now; s_new += x.it_new; s_cert += x.cert_cols;
        }
    }
    auto pct = [&](std::vector<long> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
    printf("S=%d D=%d Lmin=%d  rows %ld tried %ld pass %ld (%.1f %% of tried)  wrong %ld\n", S, D, Lmin, rows_total, rows_tried, rows_pass,
           100.0 * rows_pass / std::max(1l, rows_tried), wrong);
    double s_att = 0; for (auto& x : waves) s_att += x.cols_new;
    printf("wave iterations, mean: now %.0f  left after certificates %.0f  (+ certificate columns %.0f; loop mode: attempt columns %.0f, attempts %.1f)\n", s_now / waves.size(), s_new / waves.size(), s_cert / waves.size(), s_att / waves.size(), s_cert / waves.size());
    printf("block's longest wave (iterations): now  p50 %ld p90 %ld p99 %ld max %ld\n", pct(b_now, .5), pct(b_now, .9), pct(b_now, .99), pct(b_now, 1));
    printf("                                   left p50 %ld p90 %ld p99 %ld max %ld\n", pct(b_new, .5), pct(b_new, .9), pct(b_new, .99), pct(b_new, 1));
    printf("                      certificate columns p50 %ld p90 %ld p99 %ld max %ld\n", pct(b_cert, .5), pct(b_cert, .9), pct(b_cert, .99), pct(b_cert, 1));
    std::vector<long> idx(nb);
    for (long b = 0; b < nb; ++b) idx[b] = b;
    std::sort(idx.begin(), idx.end(), [&](long a, long b) { return b_now[a] > b_now[b]; });
    { std::vector<long> bh(nb), bq(nb); double sh = 0, sq = 0, sm = 0; for (long b = 0; b < nb; ++b) for (int w = 0; w < S; ++w) { const WaveStat& x = waves[b * S + w]; for (int h = 0; h < 2; ++h) { bh[b] = std::max(bh[b], x.it_half[h]); sh += x.it_half[h]; } for (int h = 0; h < 4; ++h) { bq[b] = std::max(bq[b], x.it_quart[h]); sq += x.it_quart[h]; } sm += x.it_mean; }
      printf("rows split: halves  longest p50 %ld p90 %ld p99 %ld max %ld, total iterations x%.2f\n", pct(bh, .5), pct(bh, .9), pct(bh, .99), pct(bh, 1), sh / s_now);
      printf("            quarters longest p50 %ld p90 %ld p99 %ld max %ld, total iterations x%.2f;  sum of per-column MEAN tests / sum of max = %.2f\n", pct(bq, .5), pct(bq, .9), pct(bq, .99), pct(bq, 1), sq / s_now, sm / s_now); }
    for (int i = 0; i < 8 && i < nb; ++i) {
        const long b = idx[i];
        for (int w = 0; w < S; ++w) { const WaveStat& x = waves[b * S + w]; printf("    w%d: it %ld cols %ld mean-it %.0f halves %ld %ld quarters %ld %ld %ld %ld\n", w, x.it_now, x.cols_now, x.it_mean, x.it_half[0], x.it_half[1], x.it_quart[0], x.it_quart[1], x.it_quart[2], x.it_quart[3]); }
        { long a = 0; for (int w = 0; w < S; ++w) a = std::max(a, waves[b * S + w].cols_new); printf("  slice %d chunk %d: now %ld left %ld cert cols %ld (loop mode: attempt columns of the longest %ld)\n", waves[b * S].k, waves[b * S].c, b_now[b], b_new[b], b_cert[b], a); }
    }
    return 0;
}
