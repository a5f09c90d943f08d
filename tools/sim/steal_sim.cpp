// steal_sim.cpp -- CPU statistics (diagnostic, not product code): the local run of k_sweep_balanced with DYNAMIC range
// cuts.  Today a row's seeded columns are cut into 8 ranges of equal count, one wave each (lane = row, shared column
// cursor: a column costs the wave the maximum over its 64 rows of the tests it takes, plus a fixed overhead), and the
// workgroup's local run lasts as long as its slowest wave.  Here a wave that runs out of columns takes the far half of
// the remaining columns of the wave that has the most left, with a fresh stack (legal by the exact-owner theorem: the
// stacks of ANY cut merge into the reference's run), until no wave has more than `minsteal` columns left.
// Every steal adds a range, i.e. a junction to the merge.
//   usage: steal_sim <seed file> [ovh = 0.5 passes per column] [restart = 6 passes per steal] [minsteal = 8 columns] [S = 8] [R = 64]
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
    std::vector<double> v;
    void add(double x) { v.push_back(x); }
    void print(const char* name) {
        if (v.empty()) { printf("%-46s (none)\n", name); return; }
        std::sort(v.begin(), v.end());
        double sum = 0; for (double x : v) sum += x;
        printf("%-46s n %6zu  mean %8.1f  p50 %7.1f  p90 %7.1f  p99 %7.1f  max %7.1f\n", name, v.size(), sum / (double)v.size(), v[v.size() / 2],
               v[(size_t)((double)v.size() * 0.9)], v[(size_t)((double)v.size() * 0.99)], v.back());
    }
};
int main(int argc, char** argv) {
    FILE* fp = fopen(argv[1], "rb");
    int32_t hdr[3];
    if (!fp || fread(hdr, 4, 3, fp) != 3) return 1;
    const int m = hdr[0], W = hdr[1], H = hdr[2];
    const double ovh = argc > 2 ? atof(argv[2]) : 0.5, restart = argc > 3 ? atof(argv[3]) : 6.0;
    const int minsteal = argc > 4 ? atoi(argv[4]) : 8, S = argc > 5 ? atoi(argv[5]) : 8, R = argc > 6 ? atoi(argv[6]) : 64;
    std::vector<uint8_t> seed((size_t)W * H);
    Dist st_max, st_mean, dyn_fin, dyn_ranges, heavy_gain;
    struct Row { double smax, smean, dyn; int ranges, k, c; };
    std::vector<Row> all;
    for (int k = 0; k < m; ++k) {
        if (fread(seed.data(), 1, seed.size(), fp) != seed.size()) return 2;
        std::vector<int> cols;
        for (int x = 0; x < W; ++x) { bool any = false; for (int y = 0; y < H && !any; ++y) any = seed[(size_t)x * H + y]; if (any) cols.push_back(x); }
        const int n = (int)cols.size();
        if (n < 16 * S) continue;  // (smaller slices take fewer ranges: light anyway)
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
            // cost of the columns (k0, k1) for a wave that starts a fresh stack at column k0: per column max over rows of tests
            auto run = [&](int k0, int k1) {
                std::vector<double> cost(std::max(0, k1 - k0 - 1), 0.0);
                for (int r = 0; r < RR; ++r) {
                    std::vector<Ent> s;
                    s.push_back(Ent{cols[k0], f[(size_t)k0 * H + c0 + r], -INFINITY});
                    for (int j = k0 + 1; j < k1; ++j) {
                        const float fq = f[(size_t)j * H + c0 + r];
                        int tt = 0; float sv;
                        for (;;) { ++tt; sv = isect(fq, cols[j], s.back().f, s.back().v); if (sv > s.back().z) break; s.pop_back(); }
                        s.push_back(Ent{cols[j], fq, sv});
                        cost[j - k0 - 1] = std::max(cost[j - k0 - 1], (double)tt);
                    }
                }
                for (double& c : cost) c += ovh;
                return cost;
            };
            struct Wave { int k0, k1; std::vector<double> cost; size_t pos; double t; };  // next column = k0 + 1 + pos; t = time its current column ends
            std::vector<Wave> wv;
            double smax = 0, ssum = 0;
            for (int w = 0; w < S; ++w) {
                const int k0 = (int)((long)n * w / S), k1 = (int)((long)n * (w + 1) / S);
                Wave x{k0, k1, run(k0, k1), 0, 0.0};
                double tot = 0; for (double c : x.cost) tot += c;
                smax = std::max(smax, tot); ssum += tot;
                wv.push_back(std::move(x));
            }
            // event simulation: advance every wave column by column; a wave that ends steals
            int ranges = S;
            std::vector<double> now(S, 0.0);   // time at which wave w has finished its columns so far
            std::vector<int> owner_of(S);      // executing unit -> index of its current range in wv
            for (int w = 0; w < S; ++w) owner_of[w] = w;
            std::vector<char> active(S, 1);
            double finish = 0;
            // progress(w, T): columns of range wv[i] completed by time T given it started at start_i: precompute prefix sums lazily
            std::vector<double> start(S, 0.0);
            std::vector<std::vector<double>> pre(S);
            auto prefix = [&](int i) { pre[i].assign(wv[i].cost.size() + 1, 0.0); for (size_t j = 0; j < wv[i].cost.size(); ++j) pre[i][j + 1] = pre[i][j] + wv[i].cost[j]; };
            for (int i = 0; i < S; ++i) prefix(i);
            std::vector<int> unit_range(S); for (int w = 0; w < S; ++w) unit_range[w] = w;
            std::vector<double> unit_end(S);
            for (int w = 0; w < S; ++w) unit_end[w] = pre[w].back();
            for (;;) {
                // the unit that ends first
                int u = -1; for (int w = 0; w < S; ++w) if (active[w] && (u < 0 || unit_end[w] < unit_end[u])) u = w;
                if (u < 0) break;
                const double T = unit_end[u];
                finish = std::max(finish, T);
                // victim: the running range with the most columns left at time T
                int vu = -1, vleft = 0, vdone = 0;
                for (int w = 0; w < S; ++w) {
                    if (!active[w] || w == u) continue;
                    const int i = unit_range[w];
                    const double el = T - start[i];
                    const int done = (int)(std::upper_bound(pre[i].begin(), pre[i].end(), el) - pre[i].begin()) - 1;  // columns finished
                    const int inprog = std::min<int>(done + 1, (int)wv[i].cost.size());                             // the one it is in stays its own
                    const int left = (int)wv[i].cost.size() - inprog;
                    if (left > vleft) { vleft = left; vu = w; vdone = inprog; }
                }
                if (vu < 0 || vleft < 2 * minsteal) { active[u] = 0; continue; }
                const int i = unit_range[vu];
                const int keep = vdone + vleft / 2;              // victim keeps cost[0 .. keep), thief takes the rest with a fresh stack
                const int split_col = wv[i].k0 + 1 + keep;       // first column of the thief's range (its stack bottom)
                Wave th{split_col, wv[i].k1, run(split_col, wv[i].k1), 0, 0.0};
                wv[i].k1 = split_col; wv[i].cost.resize(keep); pre[i].resize(keep + 1);
                unit_end[vu] = start[i] + pre[i].back();
                wv.push_back(std::move(th)); pre.emplace_back(); start.push_back(T + restart);
                const int ni = (int)wv.size() - 1;
                prefix(ni);
                unit_range[u] = ni; unit_end[u] = start[ni] + pre[ni].back();
                ++ranges;
            }
            st_max.add(smax); st_mean.add(ssum / S); dyn_fin.add(finish); dyn_ranges.add(ranges);
            all.push_back(Row{smax, ssum / S, finish, ranges, k, c0 / R});
        }
    }
    printf("%d x %d x %d: overhead %.2f passes per column, %.1f per steal, steal while >= %d columns left; %d waves x %d rows\n", m, W, H, ovh, restart, 2 * minsteal, S, R);
    st_max.print("static cuts: slowest wave (pass units)");
    st_mean.print("static cuts: mean wave");
    dyn_fin.print("dynamic cuts: workgroup's local run");
    dyn_ranges.print("dynamic cuts: ranges per row at the end");
    std::sort(all.begin(), all.end(), [](const Row& a, const Row& b) { return a.smax > b.smax; });
    printf("the 12 heaviest workgroups (static): slice chunk  static-max  static-mean  dynamic  ranges\n");
    for (size_t i = 0; i < std::min<size_t>(12, all.size()); ++i) printf("   %3d %3d   %7.1f   %7.1f   %7.1f   %d\n", all[i].k, all[i].c, all[i].smax, all[i].smean, all[i].dyn, all[i].ranges);
    return 0;
}
