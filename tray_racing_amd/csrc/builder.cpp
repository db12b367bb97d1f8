// builder.cpp — CPU construction of the 80-byte CWBVH (see builder.h).
#include "builder.h"
#include "ploc_gpu.h"
#include "collapse_gpu.h"
#include "reinsert_gpu.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <thread>

#include <sched.h>

namespace trx {

int usable_threads() {
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const int a = CPU_COUNT(&set);
        if (a > 0) n = std::min(n, a);
    }
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota> <period>" or "max <period>"
        char quota[32] = {0};
        long period = 0;
        if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && period > 0 && std::strcmp(quota, "max") != 0) {
            const long q = std::atol(quota);
            if (q > 0) n = std::max(1, std::min(n, (int)((q + period / 2) / period)));
        }
        std::fclose(f);
    }
    return n;
}

namespace {

constexpr float kInf = std::numeric_limits<float>::infinity();

// The builder's large working arrays (hundreds of megabytes on the 4-5 M triangle scenes).  Not a std::vector: that
// value-initialises - one thread writing 300-400 MB of zeroes per array before the cores that fill it get to touch it
// (a fifth of a reference-default build once the searches run on the GPU).  Memory comes from calloc, i.e. for these
// sizes as untouched zero pages: reads before writes still see zeroes, and the first touch happens on the core that
// writes the element.  Elements are trivially copyable records.
template <class T>
class BigVec {
    T *p_ = nullptr;
    size_t n_ = 0, cap_ = 0;

  public:
    BigVec() = default;
    explicit BigVec(size_t n) { resize(n); }
    BigVec(size_t n, const T &v) {
        resize(n);
        std::fill(p_, p_ + n, v);
    }
    BigVec(const BigVec &) = delete;
    BigVec &operator=(const BigVec &) = delete;
    ~BigVec() { std::free(p_); }
    void resize(size_t n) { // (new elements are zero bytes)
        if (n > cap_) {
            T *q = static_cast<T *>(std::calloc(n, sizeof(T)));
            if (!q) throw std::bad_alloc();
            if (n_) std::memcpy(static_cast<void *>(q), static_cast<const void *>(p_), n_ * sizeof(T));
            std::free(p_);
            p_ = q;
            cap_ = n;
        } else if (n > n_) {
            std::memset(static_cast<void *>(p_ + n_), 0, (n - n_) * sizeof(T));
        }
        n_ = n;
    }
    template <class It>
    void assign(It first, It last) {
        resize((size_t)(last - first));
        std::copy(first, last, p_);
    }
    void clear() { n_ = 0; }
    void swap(BigVec &o) {
        std::swap(p_, o.p_);
        std::swap(n_, o.n_);
        std::swap(cap_, o.cap_);
    }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    T *data() { return p_; }
    const T *data() const { return p_; }
    T &operator[](size_t i) { return p_[i]; }
    const T &operator[](size_t i) const { return p_[i]; }
    T *begin() { return p_; }
    T *end() { return p_ + n_; }
    const T *begin() const { return p_; }
    const T *end() const { return p_ + n_; }
};

inline Aabb empty_box() {
    return Aabb{{kInf, kInf, kInf}, {-kInf, -kInf, -kInf}};
}
inline void grow(Aabb &a, const Aabb &b) {
    for (int k = 0; k < 3; k++) {
        a.mn[k] = std::min(a.mn[k], b.mn[k]);
        a.mx[k] = std::max(a.mx[k], b.mx[k]);
    }
}
inline void grow_pt(Aabb &a, const float *p) {
    for (int k = 0; k < 3; k++) {
        a.mn[k] = std::min(a.mn[k], p[k]);
        a.mx[k] = std::max(a.mx[k], p[k]);
    }
}
inline float half_area(const Aabb &b) {
    float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
    return dx * dy + dy * dz + dz * dx;
}

// (a spin-wait hint where the target has one; api_traverse.cpp carries the same helper)
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield");
#endif
}

// The arrangement a binary split leaves its range in: the k-th misplaced element from the left changes places with the k-th
// misplaced element from the right (what a bidirectional std::partition does in libstdc++, written out so that the tree
// does not depend on the standard library: split_parallel() reproduces exactly this arrangement on every core, and a range
// of two is split by position, so the ORDER inside a side is part of the tree's primitive order).  Returns the first
// element of the right side.
template <class Pred>
inline uint32_t *partition_pairs(uint32_t *first, uint32_t *last, Pred goes_left) {
    for (;;) {
        for (;;) {
            if (first == last) return first;
            if (!goes_left(*first)) break;
            ++first;
        }
        --last;
        for (;;) {
            if (first == last) return first;
            if (goes_left(*last)) break;
            --last;
        }
        std::swap(*first, *last);
        ++first;
    }
}

// A few dozen short parallel phases in a row (the top of the BVH2 build: five per split) cost more in thread creation than
// in work when every phase starts its own threads; this keeps the threads and hands them one phase after the other.
// Workers spin briefly, then yield, between phases; the pool lives only as long as the phases do.
struct PhasePool {
    const int n;
    std::vector<std::thread> workers;
    std::atomic<uint64_t> generation{0};
    std::atomic<int> done{0};
    std::function<void(int)> phase;
    bool stop = false;

    explicit PhasePool(int threads) : n(std::max(1, threads)) {
        for (int t = 1; t < n; t++)
            workers.emplace_back([this, t]() {
                uint64_t seen = 0;
                for (;;) {
                    uint64_t g;
                    for (unsigned spins = 0; (g = generation.load(std::memory_order_acquire)) == seen; spins++) {
                        if (spins < 4000) cpu_relax();
                        else std::this_thread::yield();
                    }
                    seen = g;
                    if (stop) return;
                    phase(t);
                    done.fetch_add(1, std::memory_order_acq_rel);
                }
            });
    }
    // f(t) for t = 0 .. n-1, t = 0 on the calling thread; returns when all have returned
    template <class F>
    void run(F f) {
        if (n == 1) {
            f(0);
            return;
        }
        phase = f;
        done.store(0, std::memory_order_relaxed);
        generation.fetch_add(1, std::memory_order_release);
        f(0);
        while (done.load(std::memory_order_acquire) != n - 1) cpu_relax();
    }
    ~PhasePool() {
        stop = true;
        generation.fetch_add(1, std::memory_order_release);
        for (auto &th : workers) th.join();
    }
    PhasePool(const PhasePool &) = delete;
    PhasePool &operator=(const PhasePool &) = delete;
};

// ---- BVH2 ------------------------------------------------------------------
// Nodes are laid out in DFS pre-order: a subtree over n primitives owns exactly
// 2n-1 consecutive nodes, so the layout is independent of the thread schedule.
struct Node2 {
    Aabb box;
    uint32_t left;  // inner: index of left child (== self + 1)
    uint32_t right; // inner: index of right child
    uint32_t prim;  // leaf: primitive id
    uint32_t count; // primitives below this node (1 = leaf)
};

struct Task {
    uint32_t node, begin, end;
};

struct Bvh2Builder {
    const Aabb *boxes;
    BigVec<float> cen; // 3 per primitive
    BigVec<uint32_t> idx;
    BigVec<Node2> nodes;

    static constexpr int kMaxBins = 32;
    static constexpr uint32_t kMaxSweep = 64;
    static constexpr uint32_t kParallelSplitMin = 1u << 15; // ranges from this size up are split on every core (split_parallel)
    int kBins = 32;          // SAH bins per axis (<= kMaxBins)
    uint32_t kSweepMax = 48; // ranges up to this size get the exact sweep (<= kMaxSweep)

    // Exact SAH for small ranges: for every axis sort by centroid and sweep all n-1 splits.
    uint32_t sweep_split(uint32_t begin, uint32_t end, const Aabb &bounds) {
        const uint32_t n = end - begin;
        uint32_t order[3][kMaxSweep];
        float right_area[kMaxSweep];
        float best_cost = kInf;
        int best_axis = -1;
        uint32_t best_k = 0;
        for (int axis = 0; axis < 3; axis++) {
            uint32_t *o = order[axis];
            for (uint32_t i = 0; i < n; i++) o[i] = idx[begin + i];
            std::sort(o, o + n, [&](uint32_t a, uint32_t b) {
                float ca = cen[3 * (size_t)a + axis], cb = cen[3 * (size_t)b + axis];
                return ca < cb || (ca == cb && a < b);
            });
            Aabb acc = empty_box();
            for (uint32_t i = n - 1; i > 0; i--) {
                grow(acc, boxes[o[i]]);
                right_area[i] = half_area(acc);
            }
            acc = empty_box();
            for (uint32_t k = 1; k < n; k++) { // left = o[0..k), right = o[k..n)
                grow(acc, boxes[o[k - 1]]);
                float cost = half_area(acc) * (float)k + right_area[k] * (float)(n - k);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_k = k;
                }
            }
        }
        (void)bounds;
        for (uint32_t i = 0; i < n; i++) idx[begin + i] = order[best_axis][i];
        return begin + best_k;
    }

    // Splits [begin,end) in place; returns the split position (begin < mid < end).
    uint32_t split(uint32_t begin, uint32_t end, Aabb &bounds_out) {
        Aabb bounds = empty_box(), cb = empty_box();
        for (uint32_t i = begin; i < end; i++) {
            uint32_t p = idx[i];
            grow(bounds, boxes[p]);
            grow_pt(cb, &cen[3 * (size_t)p]);
        }
        bounds_out = bounds;
        uint32_t n = end - begin;
        if (n == 2) return begin + 1;
        if (n <= kSweepMax) return sweep_split(begin, end, bounds);

        float best_cost = kInf;
        int best_axis = -1, best_bin = -1;
        for (int axis = 0; axis < 3; axis++) {
            float lo = cb.mn[axis], hi = cb.mx[axis];
            if (!(hi > lo)) continue;
            float scale = (float)kBins / (hi - lo);
            Aabb bb[kMaxBins];
            uint32_t bc[kMaxBins];
            for (int b = 0; b < kBins; b++) {
                bb[b] = empty_box();
                bc[b] = 0;
            }
            for (uint32_t i = begin; i < end; i++) {
                uint32_t p = idx[i];
                int b = (int)((cen[3 * (size_t)p + axis] - lo) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                grow(bb[b], boxes[p]);
                bc[b]++;
            }
            float right_area[kMaxBins];
            uint32_t right_cnt[kMaxBins];
            Aabb acc = empty_box();
            uint32_t cnt = 0;
            for (int b = kBins - 1; b > 0; b--) {
                grow(acc, bb[b]);
                cnt += bc[b];
                right_area[b] = half_area(acc);
                right_cnt[b] = cnt;
            }
            acc = empty_box();
            cnt = 0;
            for (int b = 0; b < kBins - 1; b++) {
                grow(acc, bb[b]);
                cnt += bc[b];
                if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                float cost = half_area(acc) * (float)cnt + right_area[b + 1] * (float)right_cnt[b + 1];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_bin = b;
                }
            }
        }
        uint32_t mid = begin;
        if (best_axis >= 0) {
            float lo = cb.mn[best_axis], hi = cb.mx[best_axis];
            float scale = (float)kBins / (hi - lo);
            uint32_t *it = partition_pairs(idx.begin() + begin, idx.begin() + end, [&](uint32_t p) {
                int b = (int)((cen[3 * (size_t)p + best_axis] - lo) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return b <= best_bin;
            });
            mid = (uint32_t)(it - idx.begin());
        }
        if (mid == begin || mid == end) {
            // all centroids coincide (or binning degenerated): median split on
            // the widest axis of the box, ties broken by primitive id.
            int axis = 0;
            float ext[3] = {bounds.mx[0] - bounds.mn[0], bounds.mx[1] - bounds.mn[1],
                            bounds.mx[2] - bounds.mn[2]};
            if (ext[1] > ext[axis]) axis = 1;
            if (ext[2] > ext[axis]) axis = 2;
            mid = begin + n / 2;
            std::nth_element(idx.begin() + begin, idx.begin() + mid, idx.begin() + end,
                             [&](uint32_t a, uint32_t b) {
                                 float ca = cen[3 * (size_t)a + axis], cb2 = cen[3 * (size_t)b + axis];
                                 return ca < cb2 || (ca == cb2 && a < b);
                             });
        }
        return mid;
    }

    // split() for a LARGE range on `threads` cores (the top of the tree: eight levels of it touch every primitive, and on
    // one core they were 60 % of the BVH2 stage).  Same bins, same costs, same split plane - box unions and counts do not
    // depend on the order they are merged in - and the same arrangement of the primitives inside the two sides as
    // std::partition leaves (below).  The tree is the serial builder's, node for node and primitive for primitive.
    BigVec<uint32_t> scratch_idx;
    PhasePool *top_pool = nullptr; // threads of the parallel splits (run())
    uint32_t split_parallel(uint32_t begin, uint32_t end, Aabb &bounds_out, int threads) {
        const uint32_t n = end - begin;
        struct Part {
            Aabb bounds, cb;
            Aabb bb[3][kMaxBins];
            uint32_t bc[3][kMaxBins];
            uint32_t left = 0;
        };
        std::vector<Part> part((size_t)threads);
        auto chunk = [&](int t) { return std::make_pair(begin + (uint32_t)((uint64_t)n * (uint64_t)t / (uint64_t)threads),
                                                        begin + (uint32_t)((uint64_t)n * (uint64_t)(t + 1) / (uint64_t)threads)); };
        auto on_all = [&](auto f) { top_pool->run(f); };
        on_all([&](int t) {
            Part &p = part[(size_t)t];
            p.bounds = empty_box();
            p.cb = empty_box();
            for (uint32_t i = chunk(t).first; i < chunk(t).second; i++) {
                const uint32_t q = idx[i];
                grow(p.bounds, boxes[q]);
                grow_pt(p.cb, &cen[3 * (size_t)q]);
            }
        });
        Aabb bounds = empty_box(), cb = empty_box();
        for (const Part &p : part) {
            grow(bounds, p.bounds);
            grow(cb, p.cb);
        }
        bounds_out = bounds;
        float lo[3], scale[3];
        bool live[3];
        for (int axis = 0; axis < 3; axis++) {
            lo[axis] = cb.mn[axis];
            live[axis] = cb.mx[axis] > lo[axis];
            scale[axis] = live[axis] ? (float)kBins / (cb.mx[axis] - lo[axis]) : 0.f;
        }
        on_all([&](int t) {
            Part &p = part[(size_t)t];
            for (int axis = 0; axis < 3; axis++)
                for (int b = 0; b < kBins; b++) {
                    p.bb[axis][b] = empty_box();
                    p.bc[axis][b] = 0;
                }
            for (uint32_t i = chunk(t).first; i < chunk(t).second; i++) {
                const uint32_t q = idx[i];
                for (int axis = 0; axis < 3; axis++) {
                    if (!live[axis]) continue;
                    int b = (int)((cen[3 * (size_t)q + axis] - lo[axis]) * scale[axis]);
                    b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                    grow(p.bb[axis][b], boxes[q]);
                    p.bc[axis][b]++;
                }
            }
        });
        float best_cost = kInf;
        int best_axis = -1, best_bin = -1;
        for (int axis = 0; axis < 3; axis++) {
            if (!live[axis]) continue;
            Aabb bb[kMaxBins];
            uint32_t bc[kMaxBins];
            for (int b = 0; b < kBins; b++) {
                bb[b] = empty_box();
                bc[b] = 0;
                for (const Part &p : part) {
                    grow(bb[b], p.bb[axis][b]);
                    bc[b] += p.bc[axis][b];
                }
            }
            float right_area[kMaxBins];
            uint32_t right_cnt[kMaxBins];
            Aabb acc = empty_box();
            uint32_t cnt = 0;
            for (int b = kBins - 1; b > 0; b--) {
                grow(acc, bb[b]);
                cnt += bc[b];
                right_area[b] = half_area(acc);
                right_cnt[b] = cnt;
            }
            acc = empty_box();
            cnt = 0;
            for (int b = 0; b < kBins - 1; b++) {
                grow(acc, bb[b]);
                cnt += bc[b];
                if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                float cost = half_area(acc) * (float)cnt + right_area[b + 1] * (float)right_cnt[b + 1];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_bin = b;
                }
            }
        }
        uint32_t mid = begin;
        if (best_axis >= 0) {
            const float l0 = lo[best_axis], sc = scale[best_axis];
            auto goes_left = [&](uint32_t q) {
                int b = (int)((cen[3 * (size_t)q + best_axis] - l0) * sc);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return b <= best_bin;
            };
            // The arrangement partition_pairs() leaves - the k-th misplaced element from the left changes places with the k-th
            // misplaced element from the right - produced on every core.
            on_all([&](int t) {
                uint32_t c = 0;
                for (uint32_t i = chunk(t).first; i < chunk(t).second; i++) c += goes_left(idx[i]) ? 1u : 0u;
                part[(size_t)t].left = c;
            });
            uint32_t total_left = 0;
            for (int t = 0; t < threads; t++) total_left += part[(size_t)t].left;
            mid = begin + total_left;
            if (mid != begin && mid != end) {
                // holes: positions below mid that hold a right-side element (rising); strays: positions from mid on that
                // hold a left-side element (falling).  There are equally many.
                std::vector<uint32_t> n_hole((size_t)threads, 0u), n_stray((size_t)threads, 0u);
                on_all([&](int t) {
                    uint32_t h = 0, g = 0;
                    for (uint32_t i = chunk(t).first; i < chunk(t).second; i++) {
                        const bool l = goes_left(idx[i]);
                        h += (i < mid && !l) ? 1u : 0u;
                        g += (i >= mid && l) ? 1u : 0u;
                    }
                    n_hole[(size_t)t] = h;
                    n_stray[(size_t)t] = g;
                });
                std::vector<uint32_t> hole_at((size_t)threads), stray_at((size_t)threads);
                uint32_t pairs = 0;
                for (int t = 0; t < threads; t++) {
                    hole_at[(size_t)t] = pairs;
                    pairs += n_hole[(size_t)t];
                }
                uint32_t from_right = 0;
                for (int t = threads - 1; t >= 0; t--) {
                    stray_at[(size_t)t] = from_right;
                    from_right += n_stray[(size_t)t];
                }
                if (scratch_idx.size() < 2 * (size_t)pairs) scratch_idx.resize(2 * (size_t)pairs);
                uint32_t *const hole = scratch_idx.data(), *const stray = scratch_idx.data() + pairs;
                on_all([&](int t) {
                    uint32_t h = hole_at[(size_t)t];
                    for (uint32_t i = chunk(t).first; i < chunk(t).second && i < mid; i++)
                        if (!goes_left(idx[i])) hole[h++] = i;
                    uint32_t g = stray_at[(size_t)t];
                    for (uint32_t i = chunk(t).second; i-- > std::max(chunk(t).first, mid);)
                        if (goes_left(idx[i])) stray[g++] = i;
                });
                on_all([&](int t) {
                    for (uint32_t k = (uint32_t)((uint64_t)pairs * (uint64_t)t / (uint64_t)threads); k < (uint32_t)((uint64_t)pairs * (uint64_t)(t + 1) / (uint64_t)threads); k++)
                        std::swap(idx[hole[k]], idx[stray[k]]);
                });
            }
        }
        if (mid == begin || mid == end) return split(begin, end, bounds_out); // (degenerate: the serial path's fall-back)
        return mid;
    }

    // Builds one node; returns child tasks through l / r (false for a leaf).  threads > 1: a large range, split on every core.
    bool build_node(const Task &t, Task &l, Task &r, int threads = 1) {
        Node2 &nd = nodes[t.node];
        uint32_t n = t.end - t.begin;
        nd.count = n;
        if (n == 1) {
            uint32_t p = idx[t.begin];
            nd.box = boxes[p];
            nd.left = nd.right = 0;
            nd.prim = p;
            return false;
        }
        uint32_t mid = threads > 1 && n >= kParallelSplitMin && n > kSweepMax ? split_parallel(t.begin, t.end, nd.box, threads)
                                                                               : split(t.begin, t.end, nd.box);
        uint32_t nl = mid - t.begin;
        nd.left = t.node + 1;
        nd.right = t.node + 2 * nl;
        nd.prim = 0;
        l = Task{nd.left, t.begin, mid};
        r = Task{nd.right, mid, t.end};
        return true;
    }

    void build_subtree(Task root) {
        std::vector<Task> stack;
        stack.push_back(root);
        while (!stack.empty()) {
            Task t = stack.back();
            stack.pop_back();
            Task l, r;
            if (build_node(t, l, r)) {
                stack.push_back(r);
                stack.push_back(l);
            }
        }
    }

    void run(uint32_t n, int threads) {
        nodes.resize(2 * (size_t)n - 1);
        idx.resize(n);
        for (uint32_t i = 0; i < n; i++) idx[i] = i;
        // Top: ranges of at least kParallelSplitMin primitives are split one after the other, each on every core; what is
        // left goes to a shared queue, where a worker either splits a range once more (still larger than the grain: its
        // halves go back to the queue) or builds the whole subtree - nothing is ever split by one core while the others wait.
        const uint32_t grain = std::max<uint32_t>(4096, n / (uint32_t)(threads * 16));
        std::vector<Task> pending{Task{0, 0, n}}, queue;
        const auto t_top = std::chrono::steady_clock::now();
        {
            // (the pool's threads exist only when a split can run on them: the BLASes of a two-level scene are thousands of
            // small builds)
            const bool parallel_top = threads > 1 && n >= kParallelSplitMin && n > grain;
            PhasePool pool(parallel_top ? threads : 1);
            top_pool = &pool;
            while (!pending.empty()) {
                Task t = pending.back();
                pending.pop_back();
                if (t.end - t.begin < kParallelSplitMin || t.end - t.begin <= grain || threads == 1) {
                    queue.push_back(t);
                    continue;
                }
                Task l, r;
                if (build_node(t, l, r, threads)) {
                    pending.push_back(r);
                    pending.push_back(l);
                }
            }
            top_pool = nullptr;
        }
        if (getenv("TRX_BUILD_VERBOSE") && n > 100000) fprintf(stderr, "[trx build] bvh2 top: %.3f s, %zu ranges\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_top).count(), queue.size());
        std::sort(queue.begin(), queue.end(), [](const Task &a, const Task &b) { return a.end - a.begin < b.end - b.begin; }); // largest last = first out
        if (threads <= 1) {
            for (size_t i = queue.size(); i-- > 0;) build_subtree(queue[i]);
            return;
        }
        std::mutex mu;
        std::atomic<size_t> outstanding{queue.size()}; // ranges in the queue or in a worker's hands
        auto worker = [&]() {
            for (;;) {
                Task t;
                {
                    std::unique_lock<std::mutex> lock(mu);
                    if (queue.empty()) {
                        lock.unlock();
                        if (outstanding.load(std::memory_order_acquire) == 0) return;
                        std::this_thread::yield();
                        continue;
                    }
                    t = queue.back();
                    queue.pop_back();
                }
                Task l, r;
                if (t.end - t.begin > grain) {
                    if (build_node(t, l, r)) {
                        std::lock_guard<std::mutex> lock(mu);
                        queue.push_back(r);
                        queue.push_back(l);
                        outstanding.fetch_add(1, std::memory_order_acq_rel); // one taken, two added
                        continue;
                    }
                } else {
                    build_subtree(t);
                }
                outstanding.fetch_sub(1, std::memory_order_acq_rel);
            }
        };
        std::vector<std::thread> pool;
        for (int i = 0; i < threads; i++) pool.emplace_back(worker);
        for (auto &th : pool) th.join();
    }
};

// Back to DFS pre-order (left child == self + 1, a subtree over k primitives owns 2k-1 consecutive nodes, root at 0),
// which the collapse relies on.  `root` is where the tree starts in the given array.
void relayout_dfs(BigVec<Node2> &nodes, uint32_t root, int threads) {
    const size_t n = nodes.size();
    BigVec<Node2> out(n);
    typedef std::pair<uint32_t, uint32_t> Job; // (old index, new index) of a subtree root
    auto place = [&](Job root, uint32_t stop_below, std::vector<Job> *deferred) {
        std::vector<Job> todo{root};
        while (!todo.empty()) {
            auto [o, w] = todo.back();
            todo.pop_back();
            const Node2 &nd = nodes[o];
            if (deferred && nd.count <= stop_below) { // small enough: a worker places this subtree
                deferred->push_back(Job(o, w));
                continue;
            }
            Node2 &dst = out[w];
            dst.box = nd.box;
            dst.prim = nd.prim;
            dst.count = nd.count;
            if (nd.count > 1) {
                dst.left = w + 1;
                dst.right = w + 2 * nodes[nd.left].count;
                todo.emplace_back(nd.right, dst.right);
                todo.emplace_back(nd.left, dst.left);
            } else {
                dst.left = dst.right = 0;
            }
        }
    };
    std::vector<Job> jobs;
    const uint32_t grain = std::max<uint32_t>(1024u, nodes[root].count / (uint32_t)(std::max(1, threads) * 16));
    if (threads <= 1) {
        place(Job(root, 0u), 0u, nullptr);
    } else {
        place(Job(root, 0u), grain, &jobs);
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t k = next.fetch_add(1); k < jobs.size(); k = next.fetch_add(1)) place(jobs[k], 0u, nullptr);
        };
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(worker);
        for (auto &th : pool) th.join();
    }
    nodes.swap(out);
}

// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner 2018) -----------------------------
// The BVH2 stage of obvhs' ploc builder (build_cwbvh_from_tris, src/cwbvh.rs:97; parameters src/main.rs:571-585):
// primitives are sorted along a Morton curve through their box centres; every round each cluster looks `radius`
// places either side for the neighbour whose union with it has the smallest surface area, clusters that chose each
// other merge under a new node, and the survivors keep their order.  Rounds repeat until one cluster is left.
// Deterministic for any thread count: the searches only read, the merges happen in index order.
struct PlocBuilder {
    typedef unsigned __int128 u128;
    static inline uint64_t spread21(uint64_t x) { // 21 bits -> every third bit
        x &= 0x1fffffull;
        x = (x | x << 32) & 0x1f00000000ffffull;
        x = (x | x << 16) & 0x1f0000ff0000ffull;
        x = (x | x << 8) & 0x100f00f00f00f00full;
        x = (x | x << 4) & 0x10c30c30c30c30c3ull;
        x = (x | x << 2) & 0x1249249249249249ull;
        return x;
    }
    static inline u128 spread42(uint64_t x) { // 42 bits -> every third bit of 126
        return (u128)spread21(x & 0x1fffffull) | ((u128)spread21(x >> 21) << 63);
    }

    template <class Key>
    static void radix_sort(std::vector<Key> &keys, std::vector<uint32_t> &idx, int key_bits) {
        const size_t n = keys.size();
        std::vector<Key> k2(n);
        std::vector<uint32_t> i2(n);
        for (int shift = 0; shift < key_bits; shift += 16) {
            std::vector<uint32_t> count(65537, 0u);
            for (size_t i = 0; i < n; i++) count[(size_t)((keys[i] >> shift) & 0xffffu) + 1]++;
            bool trivial = false;
            for (size_t b = 0; b < 65536; b++) {
                if (count[b + 1] == n) trivial = true; // every key shares this digit
                count[b + 1] += count[b];
            }
            if (trivial) continue;
            for (size_t i = 0; i < n; i++) {
                const uint32_t d = count[(size_t)((keys[i] >> shift) & 0xffffu)]++;
                k2[d] = keys[i];
                i2[d] = idx[i];
            }
            keys.swap(k2);
            idx.swap(i2);
        }
    }

    // nodes: out, 2n-1 entries in DFS pre-order with the root at 0
    static void run(const Aabb *boxes, const float *cen, uint32_t n, uint32_t radius, uint32_t depth_threshold,
                    uint32_t sort_bits, int threads, BigVec<Node2> &nodes) {
        nodes.clear();
        if (n == 0) return;
        nodes.resize(2 * (size_t)n - 1);
        // Morton order of the box centres
        float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
        for (uint32_t i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) {
                lo[k] = std::min(lo[k], cen[3 * (size_t)i + k]);
                hi[k] = std::max(hi[k], cen[3 * (size_t)i + k]);
            }
        std::vector<uint32_t> order(n);
        for (uint32_t i = 0; i < n; i++) order[i] = i;
        const int bits = sort_bits == 128 ? 42 : 21;
        double scale[3];
        for (int k = 0; k < 3; k++) {
            const double ext = (double)hi[k] - (double)lo[k];
            scale[k] = ext > 0.0 ? ((double)((1ull << bits) - 1ull)) / ext : 0.0;
        }
        auto quant = [&](uint32_t i, int k) -> uint64_t {
            const double q = ((double)cen[3 * (size_t)i + k] - (double)lo[k]) * scale[k];
            return q <= 0.0 ? 0ull : (uint64_t)q;
        };
        if (sort_bits == 128) {
            std::vector<u128> keys(n);
            for (uint32_t i = 0; i < n; i++) keys[i] = spread42(quant(i, 0)) | (spread42(quant(i, 1)) << 1) | (spread42(quant(i, 2)) << 2);
            radix_sort(keys, order, 128);
        } else {
            std::vector<uint64_t> keys(n);
            for (uint32_t i = 0; i < n; i++) keys[i] = spread21(quant(i, 0)) | (spread21(quant(i, 1)) << 1) | (spread21(quant(i, 2)) << 2);
            radix_sort(keys, order, 64);
        }
        // leaves, in curve order, fill the first n entries; inner nodes follow in creation order
        std::vector<uint32_t> cur(n), nxt;
        for (uint32_t i = 0; i < n; i++) {
            Node2 &leaf = nodes[i];
            leaf.box = boxes[order[i]];
            leaf.left = leaf.right = 0;
            leaf.prim = order[i];
            leaf.count = 1;
            cur[i] = i;
        }
        uint32_t next_node = n;
        std::vector<uint32_t> nn;
        threads = std::max(1, threads);
        for (uint32_t round = 0; cur.size() > 1; round++) {
            const uint32_t m = (uint32_t)cur.size();
            const uint32_t r = round < depth_threshold ? 1u : std::max(1u, radius);
            nn.resize(m);
            auto search = [&](uint32_t begin, uint32_t end) {
                for (uint32_t i = begin; i < end; i++) {
                    const Aabb &bi = nodes[cur[i]].box;
                    const uint32_t j0 = i > r ? i - r : 0u, j1 = std::min(m - 1u, i + r);
                    float best = kInf;
                    uint32_t best_j = i == 0 ? 1u : i - 1u;
                    for (uint32_t j = j0; j <= j1; j++) {
                        if (j == i) continue;
                        Aabb u = bi;
                        grow(u, nodes[cur[j]].box);
                        const float a = half_area(u);
                        if (a < best) { // first of equals: the lowest index
                            best = a;
                            best_j = j;
                        }
                    }
                    nn[i] = best_j;
                }
            };
            const int use = (int)std::min<uint32_t>((uint32_t)threads, std::max(1u, m / 4096u));
            if (use <= 1) {
                search(0u, m);
            } else {
                std::vector<std::thread> pool;
                for (int t = 0; t < use; t++)
                    pool.emplace_back(search, (uint32_t)((uint64_t)m * t / use), (uint32_t)((uint64_t)m * (t + 1) / use));
                for (auto &th : pool) th.join();
            }
            nxt.clear();
            nxt.reserve(m);
            for (uint32_t i = 0; i < m; i++) {
                const uint32_t j = nn[i];
                if (nn[j] == i) {
                    if (i < j) { // the pair merges where its first member stood
                        Node2 &p = nodes[next_node];
                        p.left = cur[i];
                        p.right = cur[j];
                        p.box = nodes[cur[i]].box;
                        grow(p.box, nodes[cur[j]].box);
                        p.prim = 0;
                        p.count = nodes[cur[i]].count + nodes[cur[j]].count;
                        nxt.push_back(next_node++);
                    }
                } else {
                    nxt.push_back(cur[i]);
                }
            }
            cur.swap(nxt);
        }
        relayout_dfs(nodes, cur[0], threads);
    }
};

// ---- BVH2 reinsertion optimisation ------------------------------------------
// Meister & Bittner 2018, "Parallel Reinsertion for Bounding Volume Hierarchy
// Optimization" (the pass obvhs runs after PLOC, knob `reinsertion_batch_ratio`
// behind the reference's `-r`, src/main.rs:113-118).  The nodes with the largest
// surface area are taken out one at a time and put back where the summed area
// of the inner nodes drops the most; the search walks from the node's old place
// towards the root and explores each sibling subtree with branch-and-bound.
// Candidates are processed sequentially in a fixed order, so the result does
// not depend on the thread count.
struct Reinserter {
    static constexpr uint32_t kNone = 0xffffffffu;
    BigVec<Node2> &nodes;
    BigVec<uint32_t> parent;
    explicit Reinserter(BigVec<Node2> &n) : nodes(n), parent(n.size()) {
        const size_t total = nodes.size();
        const int threads = total < ((size_t)1 << 16) ? 1 : std::max(1, std::min(usable_threads(), 32));
        if (total) parent[0] = kNone; // every other node is some inner node's child
        on_threads(threads, [&](int t) {
            for (size_t i = total * (size_t)t / threads; i < total * (size_t)(t + 1) / threads; i++)
                if (nodes[i].count > 1) parent[nodes[i].left] = parent[nodes[i].right] = (uint32_t)i;
        });
    }
    bool leaf(uint32_t i) const { return nodes[i].count == 1; }
    uint32_t sibling(uint32_t i) const {
        const Node2 &p = nodes[parent[i]];
        return p.left == i ? p.right : p.left;
    }
    double inner_area() const {
        double s = 0.0;
        for (const Node2 &n : nodes)
            if (n.count > 1) s += half_area(n.box);
        return s;
    }

    // Best place below `top` for a box of area `area`; `gain` is what the tree
    // has saved so far by taking the node out (everything above `top`).
    typedef std::vector<std::pair<float, uint32_t>> Scratch;
    typedef std::pair<float, uint32_t> Cand; // {area, node}: a candidate of an iteration (select_candidates)
    void search(Scratch &stack, uint32_t top, float gain, const Aabb &box, float area, uint32_t &best_to,
                float &best_gain) const {
        stack.clear();
        stack.emplace_back(gain, top);
        while (!stack.empty()) {
            auto [g, id] = stack.back();
            stack.pop_back();
            if (g - area <= best_gain) continue; // even a zero-growth insertion cannot win
            const Node2 &dst = nodes[id];
            Aabb merged = dst.box;
            grow(merged, box);
            float here = g - half_area(merged); // new inner node holding {dst, node}
            if (here > best_gain) {
                best_gain = here;
                best_to = id;
            }
            if (dst.count > 1) {
                // going below dst instead grows dst to `merged`
                float below = here + half_area(dst.box);
                stack.emplace_back(below, dst.left);
                stack.emplace_back(below, dst.right);
            }
        }
    }

    bool find(Scratch &stack, uint32_t from, uint32_t &to) const {
        const uint32_t p = parent[from];
        const Aabb box = nodes[from].box;
        const float area = half_area(box);
        float gain = half_area(nodes[p].box); // p disappears
        float best_gain = 0.f;
        uint32_t best_to = kNone;
        uint32_t sib = sibling(from);
        search(stack, sib, gain, box, area, best_to, best_gain);
        Aabb shrunk = nodes[sib].box; // what the path node looks like without `from`
        for (uint32_t cur = p; parent[cur] != kNone; cur = parent[cur]) {
            sib = sibling(cur);
            search(stack, sib, gain, box, area, best_to, best_gain);
            grow(shrunk, nodes[sib].box);
            gain += half_area(nodes[parent[cur]].box) - half_area(shrunk);
        }
        to = best_to;
        return best_to != kNone;
    }

    void replace_child(uint32_t par, uint32_t old_child, uint32_t new_child) {
        if (nodes[par].left == old_child)
            nodes[par].left = new_child;
        else
            nodes[par].right = new_child;
        parent[new_child] = par;
    }
    void refit_up(uint32_t i) {
        for (; i != kNone; i = parent[i]) {
            Aabb b = nodes[nodes[i].left].box;
            grow(b, nodes[nodes[i].right].box);
            if (std::memcmp(&b, &nodes[i].box, sizeof(Aabb)) == 0) break;
            nodes[i].box = b;
        }
    }
    void move(uint32_t from, uint32_t to) {
        const uint32_t p = parent[from], s = sibling(from), g = parent[p];
        replace_child(g, p, s);
        const uint32_t tp = parent[to];
        replace_child(tp, to, p);
        nodes[p].left = to;
        nodes[p].right = from;
        parent[to] = p;
        parent[from] = p;
        nodes[p].box = nodes[to].box;
        grow(nodes[p].box, nodes[from].box);
        // Primitive counts are NOT kept up to date while nodes move: the searches only ask whether a node is a leaf
        // (count == 1, which no move changes; an inner node's stale count stays >= 2), and walking both root paths for
        // every move was a third of the sequential part of a batch.  recount() restores them once, for the re-layout.
        refit_up(g);
        refit_up(tp);
    }

    // Batched variant (Meister & Bittner's parallel formulation): the candidates of a batch search the tree as the
    // previous batch left it, all at once; their moves are then applied in candidate order, skipping a move whose nodes
    // an earlier move of the same batch touched or that would no longer be a legal re-link.  The searches are
    // read-only and the application order is fixed, so the result does not depend on the thread count.
    // Applies the moves found for candidates [begin, begin + count) in candidate order (see run_batched); returns how many.
    uint32_t apply_batch(const BigVec<Cand> &cand, size_t begin, size_t count, const uint32_t *found, std::vector<uint32_t> &touched_at,
                         uint32_t stamp, const uint32_t *ids = nullptr) {
        uint32_t moved_now = 0;
        for (size_t k = 0; k < count; k++) {
            const uint32_t from = ids ? ids[k] : cand[begin + k].second, to = found[k];
            if (to == kNone) continue;
            const uint32_t p = parent[from];
            if (p == 0 || p == kNone || to == p || to == from || parent[to] == kNone) continue;
            const uint32_t s = sibling(from), g = parent[p], tp = parent[to];
            if (to == s) continue; // already its sibling: nothing to gain
            if (touched_at[from] == stamp || touched_at[p] == stamp || touched_at[s] == stamp || touched_at[g] == stamp ||
                touched_at[to] == stamp || touched_at[tp] == stamp)
                continue; // an earlier move of this batch re-linked one of them: the search is stale
            bool inside = false; // `to` must not lie below `from` (an earlier move may have put it there)
            for (uint32_t a = to; a != kNone; a = parent[a])
                if (a == from) {
                    inside = true;
                    break;
                }
            if (inside) {
                below_skips++;
                continue;
            }
            move(from, to);
            touched_at[from] = touched_at[p] = touched_at[s] = touched_at[g] = touched_at[to] = touched_at[tp] = stamp;
            moved_now++;
        }
        return moved_now;
    }
    uint64_t below_skips = 0; // moves apply_batch dropped because their target had come to lie below the node they move

    // The batched pass with WHOLE-ITERATION batches: every candidate of an iteration searches the tree as the previous
    // iteration left it (Meister & Bittner's formulation as the paper states it), the searches on `device` (one thread
    // per candidate, reinsert_gpu.cpp) or, device < 0, on `threads` cores - the same searches, the same found[], the same
    // tree either way (tests/test_gpu_builder.py).  Moves are applied on the host in candidate order, stale ones skipped.
    // More of the searches are stale than with batches of 128, so an iteration gains less and the pass runs more of them.
    uint32_t run_whole_iterations(float batch_ratio, int iterations, int threads, int device, double *device_seconds, bool keep_layout = false) {
        const size_t n = nodes.size();
        uint32_t moved = 0;
        if (n < 8 || batch_ratio <= 0.f) return 0;
        threads = std::max(1, std::min(threads, std::min(usable_threads(), 64)));
        BigVec<Cand> cand, cand_tmp;
        std::vector<uint32_t> ids, found;
        std::vector<uint32_t> touched_at(n, 0u);
        ReinsertDevice *dev = nullptr;
        std::string err;
        if (device >= 0 && !reinsert_dev_open(device, n, &dev, err)) throw std::runtime_error("GPU build stage: " + err);
        struct Closer {
            ReinsertDevice *d;
            ~Closer() { reinsert_dev_close(d); }
        } closer{dev};
        const bool verbose = getenv("TRX_BUILD_VERBOSE") != nullptr && n > 100000;
        auto now = []() { return std::chrono::steady_clock::now(); };
        auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        double t_select = 0, t_search = 0, t_apply = 0;
        int on_host = 0; // iterations of a device pass that the host had to apply
        uint32_t max_rounds = 0;
        const unsigned long long host_apply_mask = getenv("TRX_BUILD_HOST_APPLY") ? std::strtoull(getenv("TRX_BUILD_HOST_APPLY"), nullptr, 0) : 0ull;
        if (dev && !reinsert_dev_upload(dev, nodes.data(), parent.data(), err)) throw std::runtime_error("GPU build stage: " + err);
        for (int it = 0; it < iterations; it++) {
            const auto t0 = now();
            size_t take;
            if (dev) {
                // (the candidates are chosen on the device too: the same nodes in the same order as select_candidates)
                take = n < 3 ? 0 : std::min(n - 3, (size_t)std::max(1.0, (double)n * batch_ratio));
            } else {
                take = select_candidates(cand, cand_tmp, batch_ratio, threads);
            }
            ids.resize(take);
            found.assign(take, kNone);
            if (!dev)
                for (size_t k = 0; k < take; k++) ids[k] = cand[k].second;
            const auto t1 = now();
            t_select += secs(t0, t1);
            uint32_t moved_now = 0;
            if (dev) {
                // The tree stays on the device from the first iteration to the last: selection, searches, the choice of
                // the moves that apply_batch would apply, the re-linking and the boxes are kernels (reinsert_gpu.cpp).
                // An iteration the device cannot settle - a search that outgrew its fixed stack, or a move whose target
                // has come to lie below the node it moves, which only the sequential pass resolves - comes here whole.
                bool to_host = false;
                uint32_t rounds = 0;
                // (TRX_BUILD_HOST_APPLY = bit mask of iterations handed to the host whatever the device finds: how the tests
                // reach the hand-over, which real trees almost never take)
                const bool force = host_apply_mask & (1ull << std::min(it, 63));
                if (!reinsert_dev_iteration_resident(dev, (uint32_t)take, ids.data(), found.data(), &moved_now, &to_host, device_seconds, err, force,
                                                     &rounds))
                    throw std::runtime_error("GPU build stage: " + err);
                max_rounds = std::max(max_rounds, rounds);
                const auto t2 = now();
                t_search += secs(t1, t2);
                if (to_host) {
                    on_host++;
                    if (!reinsert_dev_download(dev, nodes.data(), parent.data(), err)) throw std::runtime_error("GPU build stage: " + err);
                    Scratch scratch;
                    for (size_t k = 0; k < take; k++)
                        if (found[k] == kReinsertOverflow) {
                            found[k] = kNone;
                            (void)find(scratch, ids[k], found[k]);
                        }
                    moved_now = apply_batch(cand, 0, take, found.data(), touched_at, (uint32_t)it + 1u, ids.data());
                    if (!reinsert_dev_upload(dev, nodes.data(), parent.data(), err)) throw std::runtime_error("GPU build stage: " + err);
                    t_apply += secs(t2, now());
                }
            } else {
                std::atomic<size_t> next{0};
                on_threads(threads, [&](int) {
                    Scratch scratch;
                    for (size_t k = next.fetch_add(64); k < take; k = next.fetch_add(64))
                        for (size_t j = k; j < std::min(take, k + 64); j++) {
                            const uint32_t from = ids[j];
                            uint32_t to = kNone;
                            if (parent[from] != 0 && parent[from] != kNone) (void)find(scratch, from, to);
                            found[j] = to;
                        }
                });
                const auto t2 = now();
                t_search += secs(t1, t2);
                moved_now = apply_batch(cand, 0, take, found.data(), touched_at, (uint32_t)it + 1u, ids.data());
                t_apply += secs(t2, now());
            }
            moved += moved_now;
            if (moved_now == 0) break;
        }
        if (dev) {
            const auto t2 = now();
            if (!reinsert_dev_download(dev, nodes.data(), parent.data(), err)) throw std::runtime_error("GPU build stage: " + err);
            t_search += secs(t2, now());
            if (verbose) fprintf(stderr, "[trx build] reinsertion: tree resident on the device, %d iteration(s) applied by the host, at most %u rounds to settle an iteration's moves\n", on_host, max_rounds);
        }
        const auto t3 = now();
        // (the device's collapse stage follows links and counts the primitives itself: no pre-order layout needed)
        if (moved && !keep_layout) relayout(threads);
        if (verbose)
            fprintf(stderr, "[trx build] reinsertion: select %.3f s, search (+ copies) %.3f s, apply %.3f s, re-layout %.3f s; %llu move(s) dropped for a target below the moved node\n",
                    t_select, t_search, t_apply, secs(t3, now()), (unsigned long long)below_skips);
        return moved;
    }

    uint32_t run_batched(float batch_ratio, int iterations, int threads, bool keep_layout = false) {
        const size_t n = nodes.size();
        uint32_t moved = 0;
        if (n < 8 || batch_ratio <= 0.f) return 0;
        constexpr size_t kBatch = 128; // candidates searched against the same tree (2 048 is faster and loses a third of the gain)
        threads = std::max(1, std::min(threads, std::min(usable_threads(), 32))); // the workers spin between batches: never more than the cores
        BigVec<Cand> cand, cand_tmp;
        std::vector<uint32_t> found(kBatch);
        std::vector<uint32_t> touched_at(n, 0u); // batch stamp of the last move that re-linked this node
        uint32_t stamp = 0;
        // persistent workers: the main thread publishes a batch, everyone (it included) takes candidates by ticket
        struct Shared {
            std::atomic<uint32_t> generation{0}, next{0}, done{0};
            std::atomic<bool> quit{false};
            size_t begin = 0, count = 0;
        } sh;
        auto search_some = [&](Scratch &scratch) {
            for (uint32_t k = sh.next.fetch_add(1); k < sh.count; k = sh.next.fetch_add(1)) {
                const uint32_t from = cand[sh.begin + k].second;
                uint32_t to = kNone;
                if (parent[from] != 0 && parent[from] != kNone) (void)find(scratch, from, to);
                found[k] = to;
            }
        };
        std::vector<std::thread> pool;
        Scratch scratch;
        for (int it = 0; it < iterations; it++) {
            // (the workers spin between batches, so they only live while batches are searched: spinning beside the
            // threads of the selection would burn the CPU quota a container grants)
            const size_t take = select_candidates(cand, cand_tmp, batch_ratio, threads);
            sh.quit.store(false, std::memory_order_release);
            for (int t = 1; t < threads; t++)
                pool.emplace_back([&, start = sh.generation.load()]() {
                    Scratch scratch;
                    uint32_t seen = start;
                    for (;;) {
                        uint32_t g;
                        while ((g = sh.generation.load(std::memory_order_acquire)) == seen) {
                            if (sh.quit.load(std::memory_order_acquire)) return;
                            std::this_thread::yield();
                        }
                        seen = g;
                        search_some(scratch);
                        sh.done.fetch_add(1, std::memory_order_acq_rel);
                    }
                });
            uint32_t moved_now = 0;
            for (size_t begin = 0; begin < take; begin += kBatch) {
                sh.begin = begin;
                sh.count = std::min(kBatch, take - begin);
                sh.next.store(0, std::memory_order_relaxed);
                sh.done.store(0, std::memory_order_relaxed);
                sh.generation.fetch_add(1, std::memory_order_release);
                search_some(scratch);
                while (sh.done.load(std::memory_order_acquire) != (uint32_t)pool.size()) std::this_thread::yield();
                stamp++;
                moved_now += apply_batch(cand, begin, sh.count, found.data(), touched_at, stamp);
            }
            sh.quit.store(true, std::memory_order_release);
            for (auto &th : pool) th.join();
            pool.clear();
            moved += moved_now;
            if (moved_now == 0) break;
        }
        if (moved && !keep_layout) relayout(threads);
        return moved;
    }

    uint32_t run(float batch_ratio, int iterations, int threads, bool keep_layout = false) {
        const size_t n = nodes.size();
        uint32_t moved = 0;
        if (n < 8 || batch_ratio <= 0.f) return 0;
        BigVec<Cand> cand, cand_tmp;
        const bool verbose = getenv("TRX_BUILD_VERBOSE") != nullptr && n > 100000;
        auto now = []() { return std::chrono::steady_clock::now(); };
        auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        double t_select = 0, t_search = 0;
        size_t searched = 0;
        for (int it = 0; it < iterations; it++) {
            const auto t0 = now();
            const size_t take = select_candidates(cand, cand_tmp, batch_ratio, std::min(threads, usable_threads()));
            const auto t1 = now();
            t_select += secs(t0, t1);
            // one at a time, each search on the tree as the previous move left it (sequential: the searches
            // cost little next to the memory passes around them, and stale searches lose tree quality)
            uint32_t moved_now = 0;
            Scratch scratch;
            for (size_t c = 0; c < take; c++) {
                uint32_t from = cand[c].second, to;
                if (parent[from] == 0 || parent[from] == kNone) continue; // an earlier move put it under the root
                if (find(scratch, from, to)) {
                    move(from, to);
                    moved_now++;
                }
            }
            searched += take;
            t_search += secs(t1, now());
            moved += moved_now;
            if (moved_now == 0) break;
        }
        const auto t3 = now();
        if (moved && !keep_layout) relayout(threads);
        if (verbose)
            fprintf(stderr, "[trx build] reinsertion: select %.3f s, %zu searches + %u moves %.3f s, re-layout %.3f s\n", t_select, searched, moved,
                    t_search, secs(t3, now()));
        return moved;
    }

    // exact primitive counts of every subtree (the moves left the inner nodes' counts stale): every leaf walks towards the
    // root, the SECOND walker to reach an inner node sums its children's counts and goes on - each node is written once, by
    // whichever walker arrives last, from two final values: the same numbers on any number of threads
    void recount(int threads) {
        const size_t total = nodes.size();
        threads = total < ((size_t)1 << 16) ? 1 : std::max(1, std::min(threads, 64));
        std::unique_ptr<std::atomic<uint8_t>[]> arrived(new std::atomic<uint8_t>[total]);
        on_threads(threads, [&](int t) {
            for (size_t i = total * (size_t)t / threads; i < total * (size_t)(t + 1) / threads; i++) arrived[i].store(0, std::memory_order_relaxed);
        });
        on_threads(threads, [&](int t) {
            for (size_t i = total * (size_t)t / threads; i < total * (size_t)(t + 1) / threads; i++) {
                if (nodes[i].count != 1) continue;
                for (uint32_t p = parent[i]; p != kNone; p = parent[p]) {
                    if (arrived[p].fetch_add(1, std::memory_order_acq_rel) == 0) break; // the other child's walker finishes it
                    nodes[p].count = nodes[nodes[p].left].count + nodes[nodes[p].right].count;
                }
            }
        });
    }
    void relayout(int threads) {
        recount(threads);
        relayout_dfs(nodes, 0u, threads);
    }

    // The `take` candidates of an iteration - the nodes with the largest area, largest first, ties by index - chosen
    // and ordered on `threads` cores: the order is total, so the result is the sequential one whatever the thread count.
    static bool larger(const Cand &a, const Cand &b) { return a.first > b.first || (a.first == b.first && a.second < b.second); }
    template <class F>
    static void on_threads(int threads, F f) {
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; t++) pool.emplace_back(f, t);
        f(0);
        for (auto &th : pool) th.join();
    }
    size_t select_candidates(BigVec<Cand> &cand, BigVec<Cand> &tmp, float batch_ratio, int threads) const {
        const size_t n = nodes.size();
        threads = std::max(1, std::min(threads, 64));
        if (n < (size_t)1 << 16) threads = 1;
        // every node but the root and its children is eligible
        std::vector<size_t> first(threads + 1, 0);
        auto range = [&](int t) { return std::make_pair(1 + (n - 1) * (size_t)t / threads, 1 + (n - 1) * (size_t)(t + 1) / threads); };
        size_t take;
        if (threads == 1) {
            size_t c = 0;
            for (size_t i = 1; i < n; i++) c += parent[i] != 0;
            cand.resize(c);
            size_t w = 0;
            for (size_t i = 1; i < n; i++)
                if (parent[i] != 0) cand[w++] = Cand(half_area(nodes[i].box), (uint32_t)i);
            take = std::min(cand.size(), (size_t)std::max(1.0, (double)n * batch_ratio));
        } else {
            // On every core: a histogram of the areas' upper sixteen bits (an area is a non-negative float: its bits order
            // like its value) finds the bin the take-th largest falls into; only the nodes from that bin up are gathered,
            // and the serial selection below works on a few per cent of the array instead of all of it.
            constexpr size_t kHist = (size_t)1 << 16;
            std::vector<uint32_t> hist((size_t)threads * kHist, 0u);
            on_threads(threads, [&](int t) {
                uint32_t *h = &hist[(size_t)t * kHist];
                size_t c = 0;
                for (size_t i = range(t).first; i < range(t).second; i++) {
                    if (parent[i] == 0) continue;
                    const float a = half_area(nodes[i].box);
                    uint32_t bits;
                    std::memcpy(&bits, &a, 4);
                    h[bits >> 16]++;
                    c++;
                }
                first[t + 1] = c;
            });
            size_t eligible = 0;
            for (int t = 0; t < threads; t++) eligible += first[t + 1];
            take = std::min(eligible, (size_t)std::max(1.0, (double)n * batch_ratio));
            size_t above = 0, bin = kHist; // `above` nodes lie in bins >= bin
            while (bin > 0 && above < take) {
                bin--;
                for (int t = 0; t < threads; t++) above += hist[(size_t)t * kHist + bin];
            }
            for (int t = 0; t < threads; t++) {
                size_t c = 0;
                for (size_t b = bin; b < kHist; b++) c += hist[(size_t)t * kHist + b];
                first[t + 1] = first[t] + c;
            }
            first[0] = 0;
            cand.resize(first[threads]);
            on_threads(threads, [&](int t) {
                size_t w = first[t];
                for (size_t i = range(t).first; i < range(t).second; i++) {
                    if (parent[i] == 0) continue;
                    const float a = half_area(nodes[i].box);
                    uint32_t bits;
                    std::memcpy(&bits, &a, 4);
                    if ((bits >> 16) >= bin) cand[w++] = Cand(a, (uint32_t)i);
                }
            });
        }
        if (take < cand.size()) std::nth_element(cand.begin(), cand.begin() + take, cand.end(), larger);
        if (threads == 1 || take < (size_t)1 << 16) {
            std::sort(cand.begin(), cand.begin() + take, larger);
            return take;
        }
        // sample sort: splitters from an evenly spaced sample, one bucket per thread, buckets sorted concurrently
        const int buckets = threads;
        std::vector<Cand> sample;
        const size_t per = 64, ns = per * (size_t)buckets;
        for (size_t k = 0; k < ns; k++) sample.push_back(cand[take * k / ns]);
        std::sort(sample.begin(), sample.end(), larger);
        std::vector<Cand> split;
        for (int b = 1; b < buckets; b++) split.push_back(sample[per * (size_t)b]);
        auto bucket_of = [&](const Cand &c) { // elements before the first splitter go to bucket 0, and so on
            return (int)(std::upper_bound(split.begin(), split.end(), c, larger) - split.begin());
        };
        auto chunk = [&](int t) { return std::make_pair(take * (size_t)t / threads, take * (size_t)(t + 1) / threads); };
        std::vector<size_t> counts((size_t)threads * buckets, 0);
        on_threads(threads, [&](int t) {
            size_t *c = &counts[(size_t)t * buckets];
            for (size_t i = chunk(t).first; i < chunk(t).second; i++) c[bucket_of(cand[i])]++;
        });
        std::vector<size_t> offset((size_t)threads * buckets, 0), bucket_begin(buckets + 1, 0);
        size_t run = 0;
        for (int b = 0; b < buckets; b++) {
            bucket_begin[b] = run;
            for (int t = 0; t < threads; t++) {
                offset[(size_t)t * buckets + b] = run;
                run += counts[(size_t)t * buckets + b];
            }
        }
        bucket_begin[buckets] = run;
        tmp.resize(take);
        on_threads(threads, [&](int t) {
            size_t *o = &offset[(size_t)t * buckets];
            for (size_t i = chunk(t).first; i < chunk(t).second; i++) tmp[o[bucket_of(cand[i])]++] = cand[i];
        });
        on_threads(threads, [&](int t) {
            std::sort(tmp.begin() + bucket_begin[t], tmp.begin() + bucket_begin[t + 1], larger);
            std::copy(tmp.begin() + bucket_begin[t], tmp.begin() + bucket_begin[t + 1], cand.begin() + bucket_begin[t]);
        });
        return take;
    }
};

// ---- BVH2 -> BVH8 collapse (Ylitie et al. 2017, section 4.2) ------------------
enum : uint8_t { kLeaf = 0, kInternal = 1, kDistribute = 2 };
struct Decision {
    float cost;
    uint8_t type, dl, dr, pad;
};

struct Collapser {
    const BigVec<Node2> &n2;
    BigVec<Decision> dec; // 7 per BVH2 node
    BuildParams params;
    CwBvh &out;

    Collapser(const BigVec<Node2> &nodes, const BuildParams &p, CwBvh &o)
        : n2(nodes), dec(nodes.size() * 7), params(p), out(o) {}

    // Subtrees of at most `grain` primitives met on the way down from the root, in pre-order (disjoint blocks of the
    // pre-order array), and the nodes above them.
    void split_top(uint32_t grain, std::vector<uint32_t> &subtrees, std::vector<uint32_t> &top) const {
        std::vector<uint32_t> todo{0u};
        while (!todo.empty()) {
            const uint32_t ni = todo.back();
            todo.pop_back();
            if (n2[ni].count <= grain) {
                subtrees.push_back(ni);
                continue;
            }
            top.push_back(ni);
            todo.push_back(n2[ni].right);
            todo.push_back(n2[ni].left);
        }
    }

    void compute_costs(int threads) {
        if (threads <= 1 || n2.size() < ((size_t)1 << 16)) {
            cost_range(0, n2.size());
            return;
        }
        std::vector<uint32_t> subtrees, top;
        split_top(std::max<uint32_t>(1024u, n2[0].count / (uint32_t)(threads * 16)), subtrees, top);
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t k = next.fetch_add(1); k < subtrees.size(); k = next.fetch_add(1))
                cost_range(subtrees[k], (size_t)subtrees[k] + 2 * (size_t)n2[subtrees[k]].count - 1);
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; t++) pool.emplace_back(worker);
        worker();
        for (auto &th : pool) th.join();
        std::sort(top.begin(), top.end());
        for (size_t k = top.size(); k-- > 0;) cost_range(top[k], (size_t)top[k] + 1);
    }

    // the decisions of nodes [begin, end) of the pre-order array, last first; every child outside the range is done
    void cost_range(size_t begin, size_t end) {
        // children have larger indices than their parent (pre-order layout)
        for (size_t ni = end; ni-- > begin;) {
            const Node2 &nd = n2[ni];
            Decision *d = &dec[ni * 7];
            float area = half_area(nd.box);
            if (nd.count == 1) {
                for (int i = 0; i < 7; i++) d[i] = Decision{area * params.prim_cost, kLeaf, 0xff, 0xff, 0};
                continue;
            }
            const Decision *dl = &dec[(size_t)nd.left * 7], *dr = &dec[(size_t)nd.right * 7];
            float cost_leaf = nd.count <= params.max_prims_per_leaf
                                  ? area * (float)nd.count * params.prim_cost
                                  : kInf;
            float cost_dist = kInf;
            uint8_t bl = 0xff, br = 0xff;
            for (int k = 0; k < 7; k++) {
                float c = dl[k].cost + dr[6 - k].cost;
                if (c < cost_dist) {
                    cost_dist = c;
                    bl = (uint8_t)k;
                    br = (uint8_t)(6 - k);
                }
            }
            float cost_internal = cost_dist + area * params.traversal_cost;
            if (cost_leaf < cost_internal)
                d[0] = Decision{cost_leaf, kLeaf, bl, br, 0};
            else
                d[0] = Decision{cost_internal, kInternal, bl, br, 0};
            for (int i = 1; i < 7; i++) {
                float best = d[i - 1].cost;
                uint8_t l = 0xff, r = 0xff;
                for (int k = 0; k < i; k++) {
                    float c = dl[k].cost + dr[i - k - 1].cost;
                    if (c < best) {
                        best = c;
                        l = (uint8_t)k;
                        r = (uint8_t)(i - k - 1);
                    }
                }
                if (l != 0xff)
                    d[i] = Decision{best, kDistribute, l, r, 0};
                else
                    d[i] = d[i - 1];
            }
        }
    }

    void get_children(uint32_t ni, int i, uint32_t *children, int &count) const {
        const Node2 &nd = n2[ni];
        if (nd.count == 1) {
            children[count++] = ni;
            return;
        }
        const Decision &d = dec[(size_t)ni * 7 + i];
        if (dec[(size_t)nd.left * 7 + d.dl].type == kDistribute)
            get_children(nd.left, d.dl, children, count);
        else
            children[count++] = nd.left;
        if (dec[(size_t)nd.right * 7 + d.dr].type == kDistribute)
            get_children(nd.right, d.dr, children, count);
        else
            children[count++] = nd.right;
    }

    void collect_prims(uint32_t ni, std::vector<uint32_t> &prims) const {
        // pre-order layout: the leaves below ni are the count==1 nodes of its 2c-1 block
        const Node2 &nd = n2[ni];
        size_t last = (size_t)ni + 2 * (size_t)nd.count - 1;
        for (size_t k = ni; k < last; k++)
            if (n2[k].count == 1) prims.push_back(n2[k].prim);
    }

    struct Child {
        uint32_t n2 = 0;
        bool used = false;
        bool inner = false;
    };

    // Greedy octant-slot assignment, embree/src/bvh_embree.rs:284-349.
    void order_children(const Aabb &box, const uint32_t *children, int count, Child slots[8]) const {
        float pc[3] = {0.5f * (box.mn[0] + box.mx[0]), 0.5f * (box.mn[1] + box.mx[1]),
                       0.5f * (box.mn[2] + box.mx[2])};
        float cost[8][8];
        for (int c = 0; c < count; c++) {
            const Aabb &cb = n2[children[c]].box;
            float d[3] = {0.5f * (cb.mn[0] + cb.mx[0]) - pc[0], 0.5f * (cb.mn[1] + cb.mx[1]) - pc[1],
                          0.5f * (cb.mn[2] + cb.mx[2]) - pc[2]};
            for (int s = 0; s < 8; s++) {
                float sx = (s & 4) ? -1.f : 1.f, sy = (s & 2) ? -1.f : 1.f, sz = (s & 1) ? -1.f : 1.f;
                cost[c][s] = d[0] * sx + d[1] * sy + d[2] * sz;
            }
        }
        int assignment[8];
        bool filled[8] = {false, false, false, false, false, false, false, false};
        for (int c = 0; c < 8; c++) assignment[c] = -1;
        for (;;) {
            float min_cost = std::numeric_limits<float>::max();
            int min_slot = -1, min_index = -1;
            for (int c = 0; c < count; c++) {
                if (assignment[c] != -1) continue;
                for (int s = 0; s < 8; s++) {
                    if (!filled[s] && cost[c][s] < min_cost) {
                        min_cost = cost[c][s];
                        min_slot = s;
                        min_index = c;
                    }
                }
            }
            if (min_slot < 0) break;
            filled[min_slot] = true;
            assignment[min_index] = min_slot;
        }
        for (int c = 0; c < count; c++) {
            int s = assignment[c];
            if (s < 0) { // non-finite centre: first free slot
                for (s = 0; s < 8 && filled[s]; s++) {}
                filled[s] = true;
            }
            slots[s].n2 = children[c];
            slots[s].used = true;
            slots[s].inner = dec[(size_t)children[c] * 7].type == kInternal;
        }
    }

    // Node encoding, embree/src/bvh_embree_to_cwbvh.rs:85-186.
    static float quant_scale(float extent, float lo, float hi_world) {
        (void)lo;
        (void)hi_world;
        float x = std::max(extent, 1e-20f) * (1.0f / 255.0f);
        int k;
        float m = std::frexp(x, &k); // x = m * 2^k, m in [0.5, 1)
        float e = std::ldexp(1.0f, m == 0.5f ? k - 1 : k);
        return e;
    }

    // Where emit writes: the CwBvh itself, or the private buffers of a subtree task.
    struct Sink {
        std::vector<CwbvhNode> *nodes;
        std::vector<uint32_t> *prims;
    };
    struct Task {
        std::vector<CwbvhNode> nodes; // [0] the subtree's root, then its descendants in emission order
        std::vector<uint32_t> prims;
    };
    const std::vector<uint32_t> *task_roots = nullptr; // sorted BVH2 indices of the subtrees emitted as tasks
    std::vector<Task> *tasks = nullptr;

    // The inner (CWBVH-node) children of the node that BVH2 node `ni` becomes, in slot order.
    void inner_children(uint32_t ni, uint32_t *inner, int &n_inner) const {
        uint32_t children[8];
        int count = 0;
        if (n2[ni].count == 1)
            children[count++] = ni;
        else
            get_children(ni, 0, children, count);
        Child slots[8];
        order_children(n2[ni].box, children, count, slots);
        n_inner = 0;
        for (int s = 0; s < 8; s++)
            if (slots[s].used && slots[s].inner) inner[n_inner++] = slots[s].n2;
    }

    // Emission on every core.  The layout emit() produces is sequential by construction (a node's children are
    // allocated when it is visited, its primitives appended then), but a subtree's descendants and primitives end up
    // contiguous, so: subtrees of bounded size are emitted privately and concurrently, with indices relative to their
    // own start, and the sequential pass over the few nodes above them splices each one in where the recursion
    // reaches it, shifting its child_base / primitive_base indices.  Byte-identical to the sequential emission.
    void emit_all(int threads) {
        out.nodes.resize(1);
        Sink global{&out.nodes, &out.primitive_indices};
        const uint32_t grain = std::max<uint32_t>(4096u, n2[0].count / (uint32_t)(std::max(threads, 1) * 16));
        if (threads <= 1 || n2[0].count <= grain || n2.size() < ((size_t)1 << 16)) {
            emit(global, 0, 0);
            return;
        }
        std::vector<uint32_t> roots;
        std::vector<uint32_t> todo{0u};
        while (!todo.empty()) { // the collapsed tree from the root down to subtrees of at most `grain` primitives
            const uint32_t ni = todo.back();
            todo.pop_back();
            if (n2[ni].count <= grain) {
                roots.push_back(ni);
                continue;
            }
            uint32_t inner[8];
            int n_inner = 0;
            inner_children(ni, inner, n_inner);
            for (int k = 0; k < n_inner; k++) todo.push_back(inner[k]);
        }
        std::sort(roots.begin(), roots.end());
        std::vector<Task> done(roots.size());
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t k = next.fetch_add(1); k < roots.size(); k = next.fetch_add(1)) {
                Task &t = done[k];
                t.nodes.resize(1);
                Sink local{&t.nodes, &t.prims};
                emit(local, 0, roots[k]);
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; t++) pool.emplace_back(worker);
        worker();
        for (auto &th : pool) th.join();
        task_roots = &roots;
        tasks = &done;
        emit(global, 0, 0);
        task_roots = nullptr;
        tasks = nullptr;
    }

    // Puts a finished subtree where the sequential emission would have written it: root at out_idx, descendants and
    // primitives appended, indices shifted from the task's own numbering.
    void splice(Sink sink, uint32_t out_idx, const Task &t) {
        const uint32_t node_shift = (uint32_t)sink.nodes->size() - 1u; // task index j >= 1 lands at size + (j - 1)
        const uint32_t prim_shift = (uint32_t)sink.prims->size();
        const size_t at = sink.nodes->size();
        sink.nodes->resize(at + t.nodes.size() - 1);
        for (size_t j = 0; j < t.nodes.size(); j++) {
            CwbvhNode nd = t.nodes[j];
            nd.child_base_idx += node_shift;
            nd.primitive_base_idx += prim_shift;
            (*sink.nodes)[j == 0 ? out_idx : at + j - 1] = nd;
        }
        sink.prims->insert(sink.prims->end(), t.prims.begin(), t.prims.end());
    }

    void emit(Sink sink, uint32_t out_idx, uint32_t ni) {
        if (task_roots && sink.nodes == &out.nodes) {
            auto it = std::lower_bound(task_roots->begin(), task_roots->end(), ni);
            if (it != task_roots->end() && *it == ni) {
                splice(sink, out_idx, (*tasks)[(size_t)(it - task_roots->begin())]);
                return;
            }
        }
        const Node2 &nd = n2[ni];
        uint32_t children[8];
        int count = 0;
        if (nd.count == 1)
            children[count++] = ni; // single-primitive scene: root with one leaf child
        else
            get_children(ni, 0, children, count);
        Child slots[8];
        order_children(nd.box, children, count, slots);

        uint32_t child_base = (uint32_t)sink.nodes->size();
        uint32_t prim_base = (uint32_t)sink.prims->size();

        CwbvhNode node;
        std::memset(&node, 0, sizeof(node));
        float e[3];
        for (int k = 0; k < 3; k++) {
            node.p[k] = nd.box.mn[k];
            e[k] = quant_scale(nd.box.mx[k] - nd.box.mn[k], nd.box.mn[k], nd.box.mx[k]);
            // make sure 255 steps reach the far plane after rounding
            while (std::ceil(((double)nd.box.mx[k] - (double)nd.box.mn[k]) / (double)e[k]) > 255.0)
                e[k] *= 2.0f;
            uint32_t bits;
            std::memcpy(&bits, &e[k], 4);
            node.e[k] = (uint8_t)(bits >> 23);
        }
        node.child_base_idx = child_base;
        node.primitive_base_idx = prim_base;

        uint32_t n_inner = 0, total_tris = 0;
        std::vector<uint32_t> prims;
        for (int s = 0; s < 8; s++) {
            if (!slots[s].used) continue;
            const Aabb &cb = n2[slots[s].n2].box;
            uint8_t *qmin[3] = {node.child_min_x, node.child_min_y, node.child_min_z};
            uint8_t *qmax[3] = {node.child_max_x, node.child_max_y, node.child_max_z};
            for (int k = 0; k < 3; k++) {
                float rcp = 1.0f / e[k];
                float lo = std::floor((cb.mn[k] - node.p[k]) * rcp);
                float hi = std::ceil((cb.mx[k] - node.p[k]) * rcp);
                lo = std::min(std::max(lo, 0.0f), 255.0f);
                hi = std::min(std::max(hi, 0.0f), 255.0f);
                // keep the decoded planes conservative under f32 rounding of (c - p)
                while (lo > 0.0f && (double)node.p[k] + (double)lo * (double)e[k] > (double)cb.mn[k]) lo -= 1.0f;
                while (hi < 255.0f && (double)node.p[k] + (double)hi * (double)e[k] < (double)cb.mx[k]) hi += 1.0f;
                qmin[k][s] = (uint8_t)lo;
                qmax[k][s] = (uint8_t)hi;
            }
            if (slots[s].inner) {
                node.imask |= (uint8_t)(1u << s);
                node.child_meta[s] = (uint8_t)((24 + s) | 0x20);
                n_inner++;
            } else {
                prims.clear();
                collect_prims(slots[s].n2, prims);
                uint32_t np = (uint32_t)prims.size();
                static const uint8_t unary[4] = {0, 0x20, 0x60, 0xE0};
                node.child_meta[s] = (uint8_t)(total_tris | unary[np]);
                total_tris += np;
                for (uint32_t p : prims) sink.prims->push_back(p);
            }
        }
        (*sink.nodes)[out_idx] = node;
        // inner children are stored contiguously in slot order
        sink.nodes->resize(sink.nodes->size() + n_inner);
        uint32_t k = 0;
        for (int s = 0; s < 8; s++) {
            if (slots[s].used && slots[s].inner) {
                emit(sink, child_base + k, slots[s].n2);
                k++;
            }
        }
    }
};

void build_from_boxes(const Aabb *boxes, const float *centroids, uint64_t n, const BuildParams &params_in,
                      CwBvh &out) {
    auto t0 = std::chrono::steady_clock::now();
    BuildParams params = params_in;
    if (params.max_prims_per_leaf < 1) params.max_prims_per_leaf = 1;
    if (params.max_prims_per_leaf > 3) params.max_prims_per_leaf = 3;
    int threads = params.threads > 0 ? params.threads : usable_threads();
    if (threads < 1) threads = 1;

    out.nodes.clear();
    out.primitive_indices.clear();
    out.primitive_boxes.clear();
    out.total_aabb = empty_box();
    if (n == 0) {
        // empty scene: one node with no children; every ray misses
        CwbvhNode node;
        std::memset(&node, 0, sizeof(node));
        node.e[0] = node.e[1] = node.e[2] = 127;
        node.child_base_idx = 1;
        out.nodes.push_back(node);
        out.total_aabb = Aabb{{0, 0, 0}, {0, 0, 0}};
        out.build_seconds = 0.0;
        return;
    }
    Bvh2Builder b2;
    b2.boxes = boxes;
    if (params.ploc_search_distance == 0) b2.cen.assign(centroids, centroids + 3 * n);
    b2.kBins = std::max(2, std::min(params.sah_bins, (int)Bvh2Builder::kMaxBins));
    b2.kSweepMax = std::min<uint32_t>(params.sweep_max, Bvh2Builder::kMaxSweep);
    const bool verbose = getenv("TRX_BUILD_VERBOSE") != nullptr && n > 100000;
    auto lap = [&](const char *what) {
        if (verbose)
            fprintf(stderr, "[trx build] %-12s %.3f s\n", what,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    };
    if (params.ploc_search_distance > 0 && params.ploc_device >= 0 && n >= kDevicePlocMinPrims) {
        // Morton sort + PLOC rounds as kernels; the tree is the one PlocBuilder::run returns
        static_assert(sizeof(Node2) == 40, "Node2 is shared with ploc_gpu.cpp");
        b2.nodes.resize(2 * (size_t)n - 1);
        uint32_t root = 0;
        std::string err;
        double dev_s = 0.0;
        if (!ploc_bvh2_device(params.ploc_device, boxes, centroids, (uint32_t)n, params.ploc_search_distance,
                              params.ploc_search_depth_threshold, params.ploc_sort_bits, b2.nodes.data(), &root, &dev_s, err))
            throw std::runtime_error("GPU build stage: " + err);
        for (size_t i = n; i < b2.nodes.size(); i++) // children precede their parent in creation order
            b2.nodes[i].count = b2.nodes[b2.nodes[i].left].count + b2.nodes[b2.nodes[i].right].count;
        if (verbose) fprintf(stderr, "[trx build] ploc on device %d: %.4f s of kernels\n", params.ploc_device, dev_s);
        relayout_dfs(b2.nodes, root, threads);
    } else if (params.ploc_search_distance > 0)
        PlocBuilder::run(boxes, centroids, (uint32_t)n, params.ploc_search_distance, params.ploc_search_depth_threshold,
                         params.ploc_sort_bits, threads, b2.nodes);
    else
        b2.run((uint32_t)n, threads);
    lap(params.ploc_search_distance > 0 ? "bvh2 (ploc)" : "bvh2");
    out.total_aabb = b2.nodes[0].box;
    if (params.reinsertion_batch_ratio > 0.f && params.reinsertion_iterations > 0) {
        Reinserter opt(b2.nodes);
        const int dev = n >= kDevicePlocMinPrims ? params.ploc_device : -1; // (>= 0: the collapse runs there and needs no pre-order layout)
        if (params.reinsertion_whole_iterations) {
            double dev_s = 0.0;
            const uint32_t moved = opt.run_whole_iterations(params.reinsertion_batch_ratio, params.reinsertion_iterations, threads, dev, &dev_s, dev >= 0);
            if (verbose) fprintf(stderr, "[trx build] reinsertion: %u moves, searches on %s (%.4f s of kernels)\n", moved, dev >= 0 ? "the device" : "the host", dev_s);
        } else if (params.reinsertion_batched)
            opt.run_batched(params.reinsertion_batch_ratio, params.reinsertion_iterations, threads, dev >= 0);
        else
            opt.run(params.reinsertion_batch_ratio, params.reinsertion_iterations, threads, dev >= 0);
        lap(params.reinsertion_whole_iterations ? "reinsertion (whole iterations)" : params.reinsertion_batched ? "reinsertion (batched)" : "reinsertion");
    }

    if (params.ploc_device >= 0 && n >= kDevicePlocMinPrims) {
        // cost table, emission order and node encoding as kernels; the bytes are the ones the host stage below writes
        std::string err;
        double dev_s = 0.0;
        float root_cost = 0.f;
        if (!collapse_encode_device(params.ploc_device, b2.nodes.data(), b2.nodes.size(), params.max_prims_per_leaf, params.traversal_cost,
                                    params.prim_cost, out.nodes, out.primitive_indices, &root_cost, &dev_s, err))
            throw std::runtime_error("GPU build stage: " + err);
        out.sah_cost = root_cost / std::max(half_area(b2.nodes[0].box), 1e-30f);
        if (verbose) fprintf(stderr, "[trx build] collapse + encode on device %d: %.4f s of kernels, sah8=%.3f\n", params.ploc_device, dev_s, out.sah_cost);
        lap("collapse (device)");
        out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return;
    }
    Collapser col(b2.nodes, params, out);
    col.compute_costs(threads);
    lap("collapse dp");
    out.sah_cost = col.dec[0].cost / std::max(half_area(b2.nodes[0].box), 1e-30f);
    if (verbose) fprintf(stderr, "[trx build] n=%llu sah8=%.3f\n", (unsigned long long)n, out.sah_cost);
    if (b2.nodes[0].count > 1) col.dec[0].type = kInternal; // the root is always a node
    out.nodes.reserve(n / 4 + 16);
    out.primitive_indices.reserve(n);
    col.emit_all(threads);
    lap("emit");
    out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

} // namespace

void build_cwbvh_from_aabbs(const Aabb *boxes, uint64_t n, const BuildParams &params, CwBvh &out) {
    std::vector<float> cen(3 * n);
    for (uint64_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) cen[3 * i + k] = 0.5f * (boxes[i].mn[k] + boxes[i].mx[k]);
    build_from_boxes(boxes, cen.data(), n, params, out);
}

namespace {

// ---- pre-splitting (obvhs `pre_split`, the reference's --split, src/main.rs:572) ---------------------
// Early split clipping in the manner of Karras & Aila 2013: triangles whose boxes are mostly empty are
// cut into several references, each with the tight box of the triangle clipped to its cell, and the BVH
// is built over the references.  A triangle then appears once per reference in `primitive_indices`
// (the caller's triangle buffer repeats it, like `bvh.primitive_indices.map(|i| tris[i])` in
// src/rt_gpu/mod.rs:38-43).

struct SplitRef {
    Aabb box;
    uint32_t tri;
};

// Box of the triangle clipped to `cell` (Sutherland-Hodgman against the six planes), padded by a few ulps:
// clipped vertices are interpolated, so the padding keeps the references' union over the triangle.
Aabb clip_tri_to_cell(const float *v, const Aabb &cell) {
    double poly[16][3], tmp[16][3];
    int np = 3;
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) poly[i][k] = v[3 * i + k];
    for (int axis = 0; axis < 3 && np > 0; axis++) {
        for (int side = 0; side < 2 && np > 0; side++) {
            const double plane = side == 0 ? cell.mn[axis] : cell.mx[axis];
            int nt = 0;
            for (int i = 0; i < np; i++) {
                const double *a = poly[i], *b = poly[(i + 1) % np];
                const bool ina = side == 0 ? a[axis] >= plane : a[axis] <= plane;
                const bool inb = side == 0 ? b[axis] >= plane : b[axis] <= plane;
                if (ina) { for (int k = 0; k < 3; k++) tmp[nt][k] = a[k]; nt++; }
                if (ina != inb) {
                    const double t = (plane - a[axis]) / (b[axis] - a[axis]);
                    for (int k = 0; k < 3; k++) tmp[nt][k] = a[k] + t * (b[k] - a[k]);
                    tmp[nt][axis] = plane;
                    nt++;
                }
            }
            np = nt;
            for (int i = 0; i < np; i++)
                for (int k = 0; k < 3; k++) poly[i][k] = tmp[i][k];
        }
    }
    Aabb out = empty_box();
    for (int i = 0; i < np; i++) {
        for (int k = 0; k < 3; k++) {
            const float lo = std::nextafter((float)poly[i][k], -kInf), hi = std::nextafter((float)poly[i][k], kInf);
            out.mn[k] = std::min(out.mn[k], std::nextafter(lo, -kInf));
            out.mx[k] = std::max(out.mx[k], std::nextafter(hi, kInf));
        }
    }
    for (int k = 0; k < 3; k++) { // never outside the cell
        out.mn[k] = std::max(out.mn[k], cell.mn[k]);
        out.mx[k] = std::min(out.mx[k], cell.mx[k]);
    }
    return out;
}

float tri_double_area(const float *v) {
    const float e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
    const float c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    return std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
}

// Splits the emptiest boxes first until `budget` extra references are spent (deterministic: a heap keyed by
// (emptiness, index)).  Cuts are at the spatial median of the box's longest axis.
void pre_split(const float *verts, uint64_t n, float extra_ratio, std::vector<SplitRef> &refs) {
    refs.resize(n);
    // emptiness of a reference: half area of its box minus the area the triangle could account for
    auto emptiness = [&](const SplitRef &r) {
        return half_area(r.box) - 0.5f * tri_double_area(verts + 9 * (size_t)r.tri);
    };
    std::vector<std::pair<float, uint32_t>> heap;
    double scene_area = 0.0;
    Aabb scene = empty_box();
    for (uint64_t i = 0; i < n; i++) {
        const float *v = verts + 9 * i;
        Aabb b = empty_box();
        grow_pt(b, v);
        grow_pt(b, v + 3);
        grow_pt(b, v + 6);
        refs[i] = SplitRef{b, (uint32_t)i};
        grow(scene, b);
    }
    scene_area = half_area(scene);
    const float floor_area = (float)(scene_area * 1e-7); // do not chase slivers
    for (uint64_t i = 0; i < n; i++) {
        const float e = emptiness(refs[i]);
        if (e > floor_area) heap.emplace_back(e, (uint32_t)i);
    }
    auto less = [](const std::pair<float, uint32_t> &a, const std::pair<float, uint32_t> &b) {
        return a.first < b.first || (a.first == b.first && a.second > b.second);
    };
    std::make_heap(heap.begin(), heap.end(), less);
    uint64_t budget = (uint64_t)((double)n * extra_ratio);
    while (budget > 0 && !heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), less);
        const uint32_t ri = heap.back().second;
        heap.pop_back();
        const SplitRef r = refs[ri];
        int axis = 0;
        float ext[3] = {r.box.mx[0] - r.box.mn[0], r.box.mx[1] - r.box.mn[1], r.box.mx[2] - r.box.mn[2]};
        if (ext[1] > ext[axis]) axis = 1;
        if (ext[2] > ext[axis]) axis = 2;
        const float mid = 0.5f * (r.box.mn[axis] + r.box.mx[axis]);
        if (!(mid > r.box.mn[axis] && mid < r.box.mx[axis])) continue;
        Aabb lc = r.box, rc = r.box;
        lc.mx[axis] = mid;
        rc.mn[axis] = mid;
        const Aabb lb = clip_tri_to_cell(verts + 9 * (size_t)r.tri, lc), rb = clip_tri_to_cell(verts + 9 * (size_t)r.tri, rc);
        const bool lok = lb.mn[0] <= lb.mx[0] && lb.mn[1] <= lb.mx[1] && lb.mn[2] <= lb.mx[2];
        const bool rok = rb.mn[0] <= rb.mx[0] && rb.mn[1] <= rb.mx[1] && rb.mn[2] <= rb.mx[2];
        if (!lok || !rok) continue; // the triangle lies in one half: nothing to gain from this cut
        refs[ri].box = lb;
        refs.push_back(SplitRef{rb, r.tri});
        budget--;
        for (uint32_t idx : {ri, (uint32_t)(refs.size() - 1)}) {
            const float e = emptiness(refs[idx]) * 0.5f; // each half accounts for about half the triangle
            if (e > floor_area) {
                heap.emplace_back(e, idx);
                std::push_heap(heap.begin(), heap.end(), less);
            }
        }
    }
}

} // namespace

void build_cwbvh_from_tris(const float *verts, uint64_t n, const BuildParams &params, CwBvh &out) {
    if (params.pre_split_ratio > 0.f && n > 1) {
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<SplitRef> refs;
        pre_split(verts, n, params.pre_split_ratio, refs);
        std::vector<Aabb> rboxes(refs.size());
        std::vector<float> rcen(3 * refs.size());
        for (size_t i = 0; i < refs.size(); i++) {
            rboxes[i] = refs[i].box;
            for (int k = 0; k < 3; k++) rcen[3 * i + k] = 0.5f * (refs[i].box.mn[k] + refs[i].box.mx[k]);
        }
        build_from_boxes(rboxes.data(), rcen.data(), refs.size(), params, out);
        out.primitive_boxes.resize(out.primitive_indices.size());
        for (size_t i = 0; i < out.primitive_indices.size(); i++) {
            const SplitRef &r = refs[out.primitive_indices[i]];
            out.primitive_boxes[i] = r.box;
            out.primitive_indices[i] = r.tri; // a triangle may now appear more than once
        }
        out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return;
    }
    std::vector<Aabb> boxes(n);
    std::vector<float> cen(3 * n);
    for (uint64_t i = 0; i < n; i++) {
        const float *v = verts + 9 * i;
        Aabb b = empty_box();
        grow_pt(b, v);
        grow_pt(b, v + 3);
        grow_pt(b, v + 6);
        boxes[i] = b;
        for (int k = 0; k < 3; k++) cen[3 * i + k] = (v[k] + v[3 + k] + v[6 + k]) * (1.0f / 3.0f);
    }
    build_from_boxes(boxes.data(), cen.data(), n, params, out);
}

} // namespace trx
