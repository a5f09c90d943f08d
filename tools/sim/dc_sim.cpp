// dc_sim.cpp -- CPU statistics (diagnostic, not product code): what would the row pass of the L2 transform cost as a
// divide-and-conquer owner search instead of a stack?  By the exact-owner theorem the owner of pixel q
// of a row is argmin over the slice's seeded columns u of f[u] + (q - u)^2 in exact integer arithmetic, the smallest u on a tie,
// and owners are monotone in q.  So: solve pixel 0, then level by level the pixels half way between two solved ones, each
// over the candidates between its neighbours' owners.  Lane = row: a wave evaluates one pixel q for R rows at once over the
// UNION of the rows' candidate ranges (a wave-uniform candidate index: the column's position and (q - u)^2 are scalars),
// which is still exact (a superset of the range, scanned in ascending order with a strict compare).
// Reports per workgroup (slice, R-row chunk, 8 waves): candidate evaluations per wave if a level's pixels (or, at the top
// levels, its candidates) are dealt round-robin to the waves, against the rows' own (non-union) ranges.
//   usage: dc_sim <seed file of make_seeds.py> [R = 64] [waves = 8] [stop = 1: levels down to every pixel; 4: every 4th pixel + 3 fused]
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct Dist {
    std::vector<long> v;
    void add(long x) { v.push_back(x); }
    void print(const char* name) {
        if (v.empty()) { printf("%-44s (none)\n", name); return; }
        std::sort(v.begin(), v.end());
        double sum = 0; for (long x : v) sum += (double)x;
        printf("%-44s n %7zu  mean %9.1f  p50 %7ld  p90 %7ld  p99 %7ld  max %7ld  sum %.3e\n", name, v.size(), sum / (double)v.size(), v[v.size() / 2],
               v[(size_t)((double)v.size() * 0.9)], v[(size_t)((double)v.size() * 0.99)], v.back(), sum);
    }
};
int main(int argc, char** argv) {
    FILE* fp = fopen(argv[1], "rb");
    int32_t hdr[3];
    if (!fp || fread(hdr, 4, 3, fp) != 3) return 1;
    const int m = hdr[0], W = hdr[1], H = hdr[2];
    const int R = argc > 2 ? atoi(argv[2]) : 64, NW = argc > 3 ? atoi(argv[3]) : 8;
    std::vector<uint8_t> seed((size_t)W * H);
    int P2 = 1; while (P2 < W) P2 <<= 1;
    Dist wg_chain_mix, wg_instr_mix, lvl_max[16], wg_chain, wg_total_union, wg_total_own, ncols, lvl_union[16], lvl_own[16], depth_chain, nonself_frac;
    long chain_hist[64] = {0};
    for (int k = 0; k < m; ++k) {
        if (fread(seed.data(), 1, seed.size(), fp) != seed.size()) return 2;
        std::vector<int> cols;
        for (int x = 0; x < W; ++x) { bool any = false; for (int y = 0; y < H && !any; ++y) any = seed[(size_t)x * H + y]; if (any) cols.push_back(x); }
        const int n = (int)cols.size();
        if (n == 0) continue;
        ncols.add(n);
        std::vector<int32_t> f((size_t)n * H);
        for (int j = 0; j < n; ++j) {
            const uint8_t* c = &seed[(size_t)cols[j] * H];
            int last = -(1 << 20);
            for (int y = 0; y < H; ++y) { if (c[y]) last = y; f[(size_t)j * H + y] = y - last; }
            int nxt = 1 << 20;
            for (int y = H - 1; y >= 0; --y) { if (c[y]) nxt = y; int d = std::min(f[(size_t)j * H + y], nxt - y); f[(size_t)j * H + y] = d > 30000 ? (1 << 30) : d * d; }
        }
        for (int c0 = 0; c0 < H; c0 += R) {
            const int RR = std::min(R, H - c0);
            std::vector<int> own((size_t)RR * W, -1);
            auto solve = [&](int r, int q, int lo, int hi) {
                long best = (1L << 60); int bj = lo;
                for (int j = lo; j <= hi; ++j) { const long d = q - cols[j]; const long v = (long)f[(size_t)j * H + c0 + r] + d * d; if (v < best) { best = v; bj = j; } }
                return bj;
            };
            long chain = 0, tot_union = 0, tot_own = 0, chain_mix_instr = 0, tot_mix_instr = 0;
            // pixel 0: all candidates, split over the waves
            for (int r = 0; r < RR; ++r) own[(size_t)r * W] = solve(r, 0, 0, n - 1);
            chain += (n + NW - 1) / NW; tot_union += n; tot_own += n; chain_mix_instr += 5L * ((n + NW - 1) / NW); tot_mix_instr += 5L * n;
            int level = 0;
            for (int step = P2 / 2; step >= 1; step >>= 1, ++level) {
                std::vector<long> per_wave(NW, 0);
                long lu = 0, lo_sum = 0; int npix = 0;
                std::vector<long> pix_union, pix_cost; long lmax = 0;
                for (int q = step; q < W; q += 2 * step) {
                    int ulo = n, uhi = -1; long own_len = 0, mx = 0;
                    for (int r = 0; r < RR; ++r) {
                        const int lo = own[(size_t)r * W + q - step], hi = q + step < W ? own[(size_t)r * W + q + step] : n - 1;
                        own[(size_t)r * W + q] = solve(r, q, lo, hi);
                        ulo = std::min(ulo, lo); uhi = std::max(uhi, hi); own_len += hi - lo + 1; mx = std::max<long>(mx, hi - lo + 1);
                    }
                    pix_union.push_back(uhi - ulo + 1); lmax += mx; pix_cost.push_back(std::min(5L * (uhi - ulo + 1), 8L * mx + 6));
                    lu += uhi - ulo + 1; lo_sum += own_len; ++npix;
                }
                if (npix >= NW) { for (size_t i = 0; i < pix_union.size(); ++i) per_wave[i % NW] += pix_union[i]; chain += *std::max_element(per_wave.begin(), per_wave.end()); }
                else { for (long u : pix_union) chain += (u * npix + NW - 1) / NW; }  // candidates of a pixel split over NW / npix waves
                { std::vector<long> pw(NW, 0); long tot = 0; for (size_t i = 0; i < pix_cost.size(); ++i) { pw[i % NW] += pix_cost[i]; tot += pix_cost[i]; }
                  if (npix >= NW) chain_mix_instr += *std::max_element(pw.begin(), pw.end()); else chain_mix_instr += (tot + NW - 1) / NW; tot_mix_instr += tot; }
                lvl_max[level].add(lmax); lvl_union[level].add(lu); lvl_own[level].add(lo_sum / RR);
                tot_union += lu; tot_own += lo_sum / RR;
            }
            wg_chain_mix.add(chain_mix_instr); wg_instr_mix.add(tot_mix_instr); wg_chain.add(chain); wg_total_union.add(tot_union); wg_total_own.add(tot_own);
            // the in-place chain: depth of column -> owner of its own pixel -> ... (imgproc.h:126-127)
            long nonself = 0, owners = 0;
            for (int r = 0; r < RR; ++r) {
                std::vector<int> dep(n, -1);
                std::vector<char> is_owner(n, 0);
                for (int q = 0; q < W; ++q) is_owner[own[(size_t)r * W + q]] = 1;
                for (int j = 0; j < n; ++j) {
                    if (!is_owner[j]) continue;
                    ++owners;
                    int d = 0, cur = j;
                    while (own[(size_t)r * W + cols[cur]] != cur) { cur = own[(size_t)r * W + cols[cur]]; ++d; }
                    if (d) ++nonself;
                    chain_hist[std::min(d, 63)]++;
                }
            }
            nonself_frac.add(owners ? nonself * 1000 / owners : 0);
        }
    }
    ncols.print("seeded columns per slice");
    wg_chain.print("evaluations per wave (chain) per workgroup");
    wg_chain_mix.print("INSTRUCTIONS per wave, min(5 union, 8 max-own + 6) per pixel");
    wg_instr_mix.print("INSTRUCTIONS per workgroup, same");
    wg_total_union.print("evaluations per workgroup, union ranges");
    wg_total_own.print("evaluations per row, own ranges");
    for (int l = 0; l < 16; ++l) if (!lvl_union[l].v.empty()) { char nm[64]; snprintf(nm, 64, "  level %d union / own per row", l); lvl_union[l].print(nm); lvl_own[l].print("      own"); lvl_max[l].print("      sum over pixels of max-own over rows"); }
    nonself_frac.print("owners that do not own their own pixel (permille)");
    printf("in-place chain depth of owner columns:"); for (int d = 0; d < 64; ++d) if (chain_hist[d]) printf(" %d:%ld", d, chain_hist[d]); printf("\n");
    return 0;
}
