// sort_pool_bench.cpp -- is libstdc++'s std::sort, run on a few threads, still std::sort?  (diagnostic, not product code)
// The top three partition levels of introsort are handed to a persistent pool, every sub-range then runs
// std::__introsort_loop + std::__final_insertion_sort by itself: the comparisons, hence the moves, are the serial ones.
//   build: g++ -O3 -pthread -o sort_pool_bench sort_pool_bench.cpp ; usage: sort_pool_bench [n = 27025] [ties = 0] [threads = 8]
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>
#include <random>
#include <thread>
#include <vector>
struct M { int i; float s; float t[6]; };
struct K { float s; uint32_t p; };
struct Less { bool operator()(const K& a, const K& b) const { return a.s < b.s; } };
class Pool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_task, cv_done;
    std::deque<std::function<void()>> q;
    int pending = 0;
    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_task.wait(lk, [&] { return !q.empty(); });
            auto f = std::move(q.front()); q.pop_front();
            lk.unlock(); f(); lk.lock();
            if (--pending == 0) cv_done.notify_all();
        }
    }
public:
    explicit Pool(int n) { for (int i = 0; i < n; ++i) th.emplace_back([this] { loop(); }); for (auto& t : th) t.detach(); }
    void submit(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu); ++pending; q.push_back(std::move(f)); } cv_task.notify_one(); }
    void help_and_wait() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            if (!q.empty()) { auto f = std::move(q.front()); q.pop_front(); lk.unlock(); f(); lk.lock(); if (--pending == 0) cv_done.notify_all(); continue; }
            if (pending == 0) return;
            cv_done.wait(lk, [&] { return pending == 0 || !q.empty(); });
        }
    }
};
static void par(K* first, K* last, long depth, int levels, Pool* P) {
    auto cmp = __gnu_cxx::__ops::__iter_comp_iter(Less{});
    while (levels > 0 && last - first > 2048 && depth > 0) {
        --depth; --levels;
        K* cut = std::__unguarded_partition_pivot(first, last, cmp);
        P->submit([=] { par(cut, last, depth, levels, P); });
        last = cut;
    }
    std::__introsort_loop(first, last, depth, cmp);
    std::__final_insertion_sort(first, last, cmp);
}
int main(int argc, char** argv) {
    std::mt19937 g(1); std::uniform_real_distribution<float> d(0, 100);
    int n = argc > 1 ? atoi(argv[1]) : 27025, ties = argc > 2 ? atoi(argv[2]) : 0, nt = argc > 3 ? atoi(argv[3]) : 8;
    std::vector<M> base(n); for (int i = 0; i < n; ++i) { base[i].s = ties ? (float)(int)(d(g) * ties / 100) : d(g); base[i].i = i; }
    Pool* P = new Pool(nt - 1);
    for (int rep = 0; rep < 6; ++rep) {
        auto a = base; auto t0 = std::chrono::steady_clock::now();
        std::sort(a.begin(), a.end(), [](const M& x, const M& y) { return x.s < y.s; });
        auto t1 = std::chrono::steady_clock::now();
        std::vector<K> k(n); for (int i = 0; i < n; ++i) k[i] = {base[i].s, (uint32_t)i};
        par(k.data(), k.data() + n, std::__lg(n) * 2, 3, P);
        P->help_and_wait();
        auto t2 = std::chrono::steady_clock::now();
        std::vector<M> o(n); for (int i = 0; i < n; ++i) o[i] = base[k[i].p];
        auto t3 = std::chrono::steady_clock::now();
        bool same = true; for (int i = 0; i < n; ++i) same &= o[i].i == a[i].i && o[i].s == a[i].s;
        printf("std::sort on records %.3f ms | pool: keys + sort %.3f ms, gather %.3f ms | same permutation: %d\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
               std::chrono::duration<double, std::milli>(t2 - t1).count(), std::chrono::duration<double, std::milli>(t3 - t2).count(), (int)same);
    }
}
