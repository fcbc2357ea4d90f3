// pwalign.hip -- the C ABI of include/pwalign.h: contexts, the host-side wave-task scheduler,
// device memory management and kernel launches.  gfx950 only; there is no CPU path.
#include "../../include/pwalign.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <limits>
#include <new>
#include <numeric>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "kernel_table.h"
#include "batch_affine_tb.hip.h"

using namespace pwa;

__global__ void pwa_nop_kernel(int* p) {
    if (p && threadIdx.x == 12345) *p = 0;
}

// Symbols -> codes on the device (build_arena): n16 blocks of 16 bytes at p, every byte through the 256-entry table.  The host then only
// copies raw bytes into the upload buffers (a memcpy instead of a table lookup per byte, which was what bounded a 570 MB arena).
struct RecodeTable {
    uint8_t t[256];
};
__global__ __launch_bounds__(256) void pwa_recode_kernel(uint8_t* p, size_t n16, const RecodeTable tab) {
    __shared__ uint8_t lut[256];
    lut[threadIdx.x] = tab.t[threadIdx.x];
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        uint4 v = reinterpret_cast<const uint4*>(p)[i];
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int d = 0; d < 4; ++d)
            w[d] = (uint32_t)lut[w[d] & 0xffu] | (uint32_t)lut[(w[d] >> 8) & 0xffu] << 8 | (uint32_t)lut[(w[d] >> 16) & 0xffu] << 16 | (uint32_t)lut[w[d] >> 24] << 24;
        reinterpret_cast<uint4*>(p)[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// Host-side source / destination of the library's own host <-> device copies: page-locked, grow-only, kept in the context.
// hipMemcpy from a short-lived pageable vector works, but the runtime registers its pages with the driver for the DMA, and
// when the vector is freed the unmap notifier evicts the process's GPU queues: the NEXT kernel submission then takes 14-24 ms
// [gpu, r02: tools/cold_start.py, PWA_PROBE] -- which is what made the first run of every fresh batch 25 ms late.
struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    hipError_t reserve(size_t n) {   // contents are NOT kept
        if (n <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        n = (n + (n >> 2) + 4095) & ~(size_t)4095;   // 25 % headroom: few regrowths
        const hipError_t e = hipHostMalloc(&p, n, hipHostMallocDefault);
        if (e == hipSuccess) cap = n;
        else p = nullptr;
        return e;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// Test / diagnostic switches (include/pwalign.h, "Environment switches"): environment variables read ONCE, by
// pwa_ctx_create, into the context.  No entry point consults the environment afterwards; a test that wants another
// setting creates a fresh context.
struct Knobs {
    bool debug = false, probe = false;          // PWA_DEBUG, PWA_PROBE: host-side phase times on stderr, nop-kernel probes
    int force_rl = 0, force_w = 0;              // PWA_FORCE_RL (2 | 3 | 4), PWA_FORCE_W (1 | 4): geometry of the stripe engine
    int wg_per_cu = 0;                          // PWA_WG_PER_CU: workgroups per CU of a stripe-engine launch
    bool no_lds_pad = false;                    // PWA_NO_LDS_PAD
    std::string stamps;                         // PWA_STAMPS=<file>: per-stripe time stamps of the fill
    int trace_stripe = -1;                      // PWA_TRACE_STRIPE
    bool no_packed_dist = false;                // PWA_NO_PACKED_DIST: hw4 pass in its two-value form
    int paired = -1;                            // PWA_PAIRED: two-strip tasks as two waves with an LDS hand-off (opt-in)
    int force_lanes = -1;                       // PWA_FORCE_LANES=0: never the per-lane-text kernels
    int force_r = 0, force_mode = -1;           // PWA_FORCE_R, PWA_FORCE_MODE: strip height / kernel form of the strip engine
    uint64_t arena_limit = 0;                   // PWA_ARENA_LIMIT: bytes of sequence arena per run of the one-shot calls (tests)
    uint64_t lane_rows_limit = 0;               // PWA_LANE_ROWS_LIMIT: bytes of per-lane text rows per batch object (tests)
    int mini_per_cu = 0;                        // PWA_MINI_PER_CU: most four-wave workgroups of a mini-stripe fill per CU (experiments; default 2)
    uint64_t range_bytes = 0;                   // PWA_RANGE_BYTES: band + op bytes per range of pwa_align_batch / pwa_overlaps (tests: several ranges on small lists)
    bool no_pair_table = false;                 // PWA_NO_PAIR_TABLE: traceback fills on raw bytes (compare + select)
    bool no_keyed_tb = false;                   // PWA_NO_KEYED_TB: traceback fills in the plain int32 form
    bool no_gap_shift = false;                  // PWA_NO_GAP_SHIFT: global traceback fills in H, not G = H - gap (i + j)
    bool no_tiled_ops = false;                  // PWA_NO_TILED_OPS: op lists through the staging copy
    bool strip_wg1 = false;                     // PWA_STRIP_WG1: strip kernels as single-wave workgroups (r02 form; A/B of the placement effect)
    bool no_pipeline = false;                   // PWA_NO_PIPELINE: the runs of a one-shot score call are processed strictly one after the other
    int pipe_runs = 0;                          // PWA_PIPE_RUNS=N: cut a list that fits one arena into N pipelined runs (experiment; measured slower)
    int scores_route = -1;                      // PWA_SCORES_ROUTE: 0 = every pair on the strip engine, 1 = every pair on the stripe
                                                // engine, unset = by estimated cost (batch_create_impl)
    int tb_engine = -1;                         // PWA_TB_ENGINE: 0 = stripe engine only, 2 = mini-stripe kernels wherever they exist (also one
                                                // pair per wave for 257 .. 1024 rows, however few such pairs), unset = by pattern length and count
    void read() {
        auto flag = [](const char* n) { return std::getenv(n) != nullptr; };
        auto num = [](const char* n, int dflt) { const char* e = std::getenv(n); return e ? std::atoi(e) : dflt; };
        debug = flag("PWA_DEBUG");
        probe = flag("PWA_PROBE");
        force_rl = num("PWA_FORCE_RL", 0);
        force_w = num("PWA_FORCE_W", 0);
        wg_per_cu = num("PWA_WG_PER_CU", 0);
        no_lds_pad = flag("PWA_NO_LDS_PAD");
        if (const char* e = std::getenv("PWA_STAMPS")) stamps = e;
        trace_stripe = num("PWA_TRACE_STRIPE", -1);
        no_packed_dist = flag("PWA_NO_PACKED_DIST");
        paired = num("PWA_PAIRED", -1);
        force_lanes = num("PWA_FORCE_LANES", -1);
        force_r = num("PWA_FORCE_R", 0);
        force_mode = num("PWA_FORCE_MODE", -1);
        if (const char* e = std::getenv("PWA_ARENA_LIMIT")) arena_limit = std::max<uint64_t>(1024, std::strtoull(e, nullptr, 10));
        if (const char* e = std::getenv("PWA_LANE_ROWS_LIMIT")) lane_rows_limit = std::max<uint64_t>(1024, std::strtoull(e, nullptr, 10));
        mini_per_cu = num("PWA_MINI_PER_CU", 0);
        if (const char* e = std::getenv("PWA_RANGE_BYTES")) range_bytes = std::max<uint64_t>(4096, std::strtoull(e, nullptr, 10));
        no_pair_table = flag("PWA_NO_PAIR_TABLE");
        no_keyed_tb = flag("PWA_NO_KEYED_TB");
        no_gap_shift = flag("PWA_NO_GAP_SHIFT");
        no_tiled_ops = flag("PWA_NO_TILED_OPS");
        no_pipeline = flag("PWA_NO_PIPELINE");
        pipe_runs = num("PWA_PIPE_RUNS", 0);
        strip_wg1 = flag("PWA_STRIP_WG1");
        scores_route = num("PWA_SCORES_ROUTE", -1);
        tb_engine = num("PWA_TB_ENGINE", -1);
    }
};

// ------------------------------------------------------------------------------------ context
struct pwa_ctx {
    Knobs knobs;
    int device = 0;
    int num_cu = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    std::string err;
    float fill_ms = 0.f, tb_ms = 0.f;
    uint64_t band_bytes = 0;
    bool score_band = false;   // pwa_ctx_set_score_band: also materialise the int32 score band in HBM
    // Traceback / score band workspaces of pwa_align*, kept between calls (grow-only, at most kBandCacheMax each):
    // hipMalloc of several GiB is sometimes fast (0.3 ms) and sometimes not (0.2 - 1.5 s) depending on the state of the
    // device's memory, and a caller that aligns batch after batch should pay it once.
    void* band_cache = nullptr;
    size_t band_cache_bytes = 0;
    void* sband_cache = nullptr;
    size_t sband_cache_bytes = 0;
    // the strip hand-off workspace of the last destroyed batch (5.2 GB for C3): the next batch takes it over
    void* hand_cache = nullptr;
    size_t hand_cache_bytes = 0;
    // pwa_align* work buffers (sequence arena, op lists, results, pair descriptors, task list, hand-off rows, progress words,
    // per-stripe bests, queue): grow-only, reused by the next call -- a call that aligns a batch costs no hipMalloc / hipFree
    // (each of which also synchronises the device) once the context has seen a batch of that size
    enum { POOL_ARENA, POOL_OPS, POOL_RES, POOL_DESC, POOL_TASKS, POOL_ROWS, POOL_PROGRESS, POOL_BEST, POOL_QUEUE, POOL_N };
    void* pool[POOL_N] = {};
    size_t pool_bytes[POOL_N] = {};
    // page-locked staging of everything the library itself uploads or reads back (see PinnedBuf)
    hipStream_t aux_stream = nullptr;                  // pwa_batch_run: the mini-stripe launches of a split batch run next to its stripe launch
    hipEvent_t aux_ev[2] = {nullptr, nullptr};         // fork / join of that
    hipStream_t copy_stream = nullptr;                 // uploads that overlap host work (build_arena)
    hipEvent_t copy_ev[2] = {nullptr, nullptr};
    // Device buffers of destroyed batch objects, kept for the next one (DevBuf below): a steady stream of batches -- the runs of a
    // one-shot call over a large list, a caller that builds batch after batch -- costs no hipMalloc and, more to the point, no
    // hipFree: hipFree waits for ALL work on the device, i.e. for the kernels of the batch that is still running, and with it the
    // overlap of preparing run k + 1 with computing run k would be gone (scores_in_arena_chunks).
    std::vector<std::pair<void*, size_t>> free_list;
    size_t free_list_bytes = 0;
    enum { PIN_ARENA, PIN_ARENA2, PIN_TASKS, PIN_SLOT0, PIN_SLOT1, PIN_SLOT2, PIN_SLOT3, PIN_SLOT4, PIN_DESC, PIN_TL, PIN_RES, PIN_BOUNCE, PIN_N };
    PinnedBuf pin[PIN_N];
};
constexpr size_t kBandCacheMax = 64ull << 30;   // (288 GB of HBM per GPU: a 4096-pair batch with both bands is 33 GB)

namespace {

constexpr size_t kFreeListMaxBytes = 24ull << 30, kFreeListMaxCount = 64;
struct DevBuf {   // RAII device allocation; with `pool` set, released buffers go to the context's free list and come back from it
    void* p = nullptr;
    size_t bytes = 0;
    pwa_ctx* pool = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) {
            if (pool && pool->free_list.size() < kFreeListMaxCount && pool->free_list_bytes + bytes <= kFreeListMaxBytes) {
                pool->free_list.emplace_back(p, bytes);
                pool->free_list_bytes += bytes;
            } else {
                (void)hipFree(p);
            }
        }
        p = nullptr;
        bytes = 0;
    }
    hipError_t alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        if (pool) {   // best fit among the kept buffers: at least n, at most 2 n + 1 MiB (a 5 GB block is not spent on a 1 KB request)
            size_t best = pool->free_list.size();
            for (size_t i = 0; i < pool->free_list.size(); ++i) {
                const size_t have = pool->free_list[i].second;
                if (have >= n && have <= 2 * n + (1u << 20) && (best == pool->free_list.size() || have < pool->free_list[best].second)) best = i;
            }
            if (best < pool->free_list.size()) {
                p = pool->free_list[best].first;
                bytes = pool->free_list[best].second;
                pool->free_list_bytes -= bytes;
                pool->free_list.erase(pool->free_list.begin() + (long)best);
                return hipSuccess;
            }
        }
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess && pool && !pool->free_list.empty()) {   // out of memory with buffers parked: give them back and try again
            for (auto& f : pool->free_list) (void)hipFree(f.first);
            pool->free_list.clear();
            pool->free_list_bytes = 0;
            (void)hipGetLastError();
            e = hipMalloc(&p, n);
        }
        if (e == hipSuccess) bytes = n;
        else p = nullptr;
        return e;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// Stable LSD radix sort of `idx` by 64-bit keys (16-bit digits; passes whose digit is constant are skipped).
// Sorting a million pairs with std::sort and a comparator that looks lengths up cost ~95 ms per batch [gpu box].
void radix_sort_by_key(std::vector<uint64_t>& key, std::vector<uint32_t>& idx) {
    const size_t n = idx.size();
    std::vector<uint64_t> key2(n);
    std::vector<uint32_t> idx2(n);
    std::vector<size_t> cnt(65536);
    for (int pass = 0; pass < 4; ++pass) {
        const int sh = 16 * pass;
        std::fill(cnt.begin(), cnt.end(), 0);
        for (size_t i = 0; i < n; ++i) ++cnt[(key[i] >> sh) & 0xffff];
        if (n && cnt[(key[0] >> sh) & 0xffff] == n) continue;
        size_t run = 0;
        for (size_t d = 0; d < 65536; ++d) {
            const size_t c = cnt[d];
            cnt[d] = run;
            run += c;
        }
        for (size_t i = 0; i < n; ++i) {
            const size_t o = cnt[(key[i] >> sh) & 0xffff]++;
            key2[o] = key[i];
            idx2[o] = idx[i];
        }
        key.swap(key2);
        idx.swap(idx2);
    }
}

// Stable counting sort of `idx` by key(idx[i]) in [0, n_buckets): two linear passes.
template <class KeyFn>
void counting_sort(std::vector<uint32_t>& idx, std::vector<uint32_t>& tmp, size_t n_buckets, KeyFn key) {
    const size_t n = idx.size();
    if (n >= (1u << 18) && n_buckets <= (1u << 16)) {
        // a million pairs: both passes are scattered memory accesses -- on a few threads, each with its own histogram over its own
        // contiguous part of idx (thread t's elements of a bucket go behind those of the threads before it: still stable)
        const int T = (int)std::min<size_t>({8, std::max(1u, std::thread::hardware_concurrency()), n >> 16});
        std::vector<std::vector<uint32_t>> cnt((size_t)T, std::vector<uint32_t>(n_buckets, 0));
        auto part = [&](int t) { return std::make_pair(n * (size_t)t / (size_t)T, n * (size_t)(t + 1) / (size_t)T); };
        auto run = [&](auto&& fn) {
            std::vector<std::thread> th;
            for (int t = 1; t < T; ++t) th.emplace_back(fn, t);
            fn(0);
            for (auto& x : th) x.join();
        };
        run([&](int t) {
            const auto [a, z] = part(t);
            uint32_t* const c = cnt[(size_t)t].data();
            for (size_t o = a; o < z; ++o) ++c[key(idx[o])];
        });
        uint32_t at = 0;
        for (size_t bkt = 0; bkt < n_buckets; ++bkt)
            for (int t = 0; t < T; ++t) {
                const uint32_t c = cnt[(size_t)t][bkt];
                cnt[(size_t)t][bkt] = at;
                at += c;
            }
        tmp.resize(n);
        run([&](int t) {
            const auto [a, z] = part(t);
            uint32_t* const c = cnt[(size_t)t].data();
            for (size_t o = a; o < z; ++o) tmp[c[key(idx[o])]++] = idx[o];
        });
        idx.swap(tmp);
        return;
    }
    std::vector<uint32_t> cnt(n_buckets + 1, 0);
    for (const uint32_t v : idx) ++cnt[key(v) + 1];
    for (size_t b = 0; b < n_buckets; ++b) cnt[b + 1] += cnt[b];
    tmp.resize(idx.size());
    for (const uint32_t v : idx) tmp[cnt[key(v)]++] = v;
    idx.swap(tmp);
}

// idx by DESCENDING length, stable (equal lengths keep their order): linear passes instead of std::stable_sort's n log n compares
// through two indirections ([cpu] 16 384 pairs of random lengths: 0.95 ms for the merge sort)
template <class LenFn>
void sort_by_length_desc(std::vector<uint32_t>& idx, LenFn len_of) {
    if (idx.size() < 2) return;
    uint64_t lmin = ~0ull, lmax = 0;
    for (const uint32_t v : idx) {
        const uint64_t l = len_of(v);
        lmin = std::min(lmin, l);
        lmax = std::max(lmax, l);
    }
    if (lmin == lmax) return;
    if (lmax - lmin <= 4 * (uint64_t)idx.size() + 65536) {
        std::vector<uint32_t> tmp;
        counting_sort(idx, tmp, (size_t)(lmax - lmin + 1), [&](uint32_t v) { return (size_t)(lmax - len_of(v)); });
    } else {
        std::vector<uint64_t> key(idx.size());
        for (size_t o = 0; o < idx.size(); ++o) key[o] = lmax - len_of(idx[o]);
        radix_sort_by_key(key, idx);
    }
}

// Runs fn(first_seq, last_seq, thread) over the sequences, split into byte-balanced contiguous ranges, on up to
// 16 host threads (one per >= 8 MiB): the host passes over the input (alphabet scan, symbol coding into the
// arena) are memory-bound loops that otherwise dominate the call for inputs of hundreds of MB.
template <class F>
void for_seq_ranges(const uint64_t* seq_off, uint32_t n_seq, F&& fn, int* n_threads_out = nullptr, uint64_t bytes_per_thread = 8ull << 20) {
    const uint64_t total = n_seq ? seq_off[n_seq] - seq_off[0] : 0;
    int T = (int)std::min<uint64_t>({16, total / bytes_per_thread + 1, std::max(1u, std::thread::hardware_concurrency())});
    T = std::max(1, std::min<int>(T, (int)std::max<uint32_t>(n_seq, 1)));
    if (n_threads_out) *n_threads_out = T;
    std::vector<uint32_t> cut((size_t)T + 1, n_seq);
    cut[0] = 0;
    for (int t = 1; t < T; ++t) {
        const uint64_t want = seq_off[0] + total / (uint64_t)T * (uint64_t)t;
        cut[(size_t)t] = (uint32_t)(std::lower_bound(seq_off, seq_off + n_seq, want) - seq_off);
    }
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back([&, t] { fn(cut[(size_t)t], cut[(size_t)t + 1], t); });
    fn(cut[0], cut[1], 0);
    for (auto& x : th) x.join();
}

// The traceback kernels keep H * 4 + priority in int32 (pair_fill.hip.h): |H| has to stay below 2^28.
// (bits = 26: local fills of the mini-stripe kernels, whose first-maximum records hold H * 16 + a step index, mini_fill.hip.h)
bool tb_range_ok(uint64_t n_plus_m, int match, int mismatch, int gap, int bits = 28) {
    const uint64_t amax = (uint64_t)std::max<int64_t>({std::llabs((long long)match), std::llabs((long long)mismatch),
                                                       std::llabs((long long)gap), 1});
    return (n_plus_m + 2) <= (1ull << bits) / amax;
}

// A workspace of `bytes` from the context's cache slot (see pwa_ctx): reused when big enough, regrown otherwise;
// requests beyond kBandCacheMax are served by `fallback` and freed with it.
hipError_t cached_workspace(void*& slot, size_t& slot_bytes, size_t bytes, DevBuf& fallback, void** out) {
    if (bytes > kBandCacheMax) {
        const hipError_t e = fallback.alloc(bytes);
        *out = fallback.p;
        return e;
    }
    if (slot_bytes < bytes) {
        if (slot) (void)hipFree(slot);
        slot = nullptr;
        slot_bytes = 0;
        const hipError_t e = hipMalloc(&slot, bytes);
        if (e != hipSuccess) {
            slot = nullptr;
            return e;
        }
        slot_bytes = bytes;
    }
    *out = slot;
    return hipSuccess;
}

// pageable memory that the library does not own (or that is too large to mirror in page-locked memory): through a bounce buffer
hipError_t upload_via_bounce(pwa_ctx* c, void* dst, const void* src, size_t bytes) {
    constexpr size_t kChunk = 8u << 20;
    hipError_t e = c->pin[pwa_ctx::PIN_BOUNCE].reserve(std::min(bytes, kChunk));
    for (size_t o = 0; e == hipSuccess && o < bytes; o += kChunk) {
        const size_t n = std::min(kChunk, bytes - o);
        std::memcpy(c->pin[pwa_ctx::PIN_BOUNCE].p, static_cast<const uint8_t*>(src) + o, n);
        e = hipMemcpy(static_cast<uint8_t*>(dst) + o, c->pin[pwa_ctx::PIN_BOUNCE].p, n, hipMemcpyHostToDevice);
    }
    return e;
}

// The device arena of a call: every used sequence s at aoff[s] (16-byte aligned) as symbols -- through `table` (256 entries) or
// copied when table == nullptr -- and zeros everywhere else.  It goes up in pieces of ~32 MiB: while piece k is on its way
// (copy stream, from one of two page-locked buffers of the context) piece k + 1 is being coded by several host threads into the
// other -- readFasta's blob is never repacked into a second host copy, and a 570 MB arena costs 2 x 32 MiB of pinned memory.
// r03: arenas of a MiB and more are coded ON THE DEVICE when no sequence holds a NUL byte (`nul_free`: the padding between sequences is
// zeros and has to stay zeros, so the device table maps 0 to 0): the host threads then only copy raw bytes into the pieces and a small
// kernel behind every piece's copy turns them into codes in place.
hipError_t build_arena(pwa_ctx* c, void* d_arena, uint64_t arena_bytes, const uint8_t* seq_bytes, const uint64_t* seq_off, uint32_t n_seq,
                       const std::vector<uint8_t>& is_used, const std::vector<uint64_t>& aoff, const uint8_t* table, bool nul_free = false) {
    constexpr uint64_t kPiece = 32ull << 20;
    const bool on_device = table && nul_free && arena_bytes >= (1ull << 20) && arena_bytes % 16 == 0;
    RecodeTable rt;
    if (on_device) {
        std::memcpy(rt.t, table, 256);
        rt.t[0] = 0;
        table = nullptr;   // the pieces take raw bytes
    }
    std::vector<uint32_t> used;
    for (uint32_t s = 0; s < n_seq; ++s)
        if (is_used[s]) used.push_back(s);
    if (used.empty()) return hipMemset(d_arena, 0, arena_bytes);
    hipError_t e = hipSuccess;
    int piece = 0;
    for (size_t u0 = 0; u0 < used.size() && e == hipSuccess; ++piece) {
        size_t u1 = u0 + 1;
        const uint64_t a0 = u0 == 0 ? 0 : aoff[used[u0]];
        auto end_of = [&](size_t u) { return u < used.size() ? aoff[used[u]] : arena_bytes; };
        while (u1 < used.size() && end_of(u1 + 1) - a0 <= kPiece) ++u1;
        const uint64_t a1 = end_of(u1), bytes = a1 - a0;
        PinnedBuf& pb = c->pin[pwa_ctx::PIN_ARENA + (piece & 1)];
        if (piece >= 2) e = hipEventSynchronize(c->copy_ev[piece & 1]);   // the copy that last read this buffer
        if (e == hipSuccess) e = pb.reserve(bytes);
        if (e != hipSuccess) break;
        uint8_t* const host = pb.as<uint8_t>();
        // sequences u0 .. u1-1 of the piece over a few threads, byte-balanced
        const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>({16, bytes / (1ull << 20) + 1, std::max(1u, std::thread::hardware_concurrency()), (uint64_t)(u1 - u0)}));
        auto work = [&](int t) {
            const uint64_t lo = a0 + bytes / T * t, hi = t + 1 == T ? a1 : a0 + bytes / T * (t + 1);
            // first sequence whose region starts at or after lo (regions are [aoff[s], aoff[next]))
            size_t u = std::lower_bound(used.begin() + u0, used.begin() + u1, lo, [&](uint32_t sidx, uint64_t v) { return aoff[sidx] < v; }) - used.begin();
            if (t == 0) {
                u = u0;
                if (a0 < aoff[used[u0]]) std::memset(host, 0, aoff[used[u0]] - a0);
            }
            for (; u < u1 && aoff[used[u]] < hi; ++u) {
                const uint32_t sidx = used[u];
                const uint64_t len = seq_off[sidx + 1] - seq_off[sidx], r0 = aoff[sidx], r1 = end_of(u + 1);
                uint8_t* dst = host + (r0 - a0);
                const uint8_t* src = seq_bytes + seq_off[sidx];
                if (table)
                    for (uint64_t o = 0; o < len; ++o) dst[o] = table[src[o]];
                else if (len)
                    std::memcpy(dst, src, len);
                std::memset(dst + len, 0, r1 - r0 - len);
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
        e = hipMemcpyAsync(static_cast<uint8_t*>(d_arena) + a0, host, bytes, hipMemcpyHostToDevice, c->copy_stream);
        if (e == hipSuccess) e = hipEventRecord(c->copy_ev[piece & 1], c->copy_stream);
        if (e == hipSuccess && on_device && a0 % 16 == 0 && bytes % 16 == 0) {   // (pieces start and end on sequence regions: multiples of 16)
            const size_t n16 = (size_t)(bytes / 16);
            hipLaunchKernelGGL(pwa_recode_kernel, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 4096)), dim3(256), 0, c->copy_stream,
                               static_cast<uint8_t*>(d_arena) + a0, n16, rt);
            e = hipGetLastError();
        } else if (e == hipSuccess && on_device) {
            e = hipErrorInvalidValue;   // cannot happen: regions are 16-byte aligned
        }
        u0 = u1;
    }
    const hipError_t e2 = hipStreamSynchronize(c->copy_stream);
    return e != hipSuccess ? e : e2;
}

int fail(pwa_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define HIPC(ctx, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            return fail((ctx), e_ == hipErrorOutOfMemory ? PWA_E_NOMEM : PWA_E_HIP,                  \
                        std::string(#call) + ": " + hipGetErrorString(e_));                          \
        }                                                                                            \
    } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

inline int32_t wrap_mul(int64_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }

// Geometry of the wavefront (pair) engine: RL rows per lane (stripe = 64*RL rows) and W compute waves
// per workgroup (a workgroup task = W consecutive stripes + one helper wave).  Pairs of a single
// stripe use W = 1; short multi-stripe pairs get RL = 2 (twice the stripes = twice the waves in flight).
struct PairGeom {
    int rl, w;
};
PairGeom choose_geom(const Knobs& kn, uint64_t max_n, bool keyed = true, bool keyed_tb = false) {
    // (W = 3 -- three stripes + the helper = one wave per SIMD -- was measured in r02: no gain over W = 4, the stripes behind the
    // first workgroup run ~5-9 % slower than the first either way: they run at the edge of what their producer has posted.)
    // RL = 2 up to 32k rows (twice the stripes = twice the waves of a pair in flight), RL = 4 beyond.  (The choice was measured in r01 --
    // 10k x 10k: RL = 2 10 % ahead; 100k x 100k: RL = 4 5 % ahead -- and has held since; today's fills: 1.65 ms / 11.8 - 12.2 ms)
    PairGeom g{max_n <= 32768 ? 2 : 4, 4};
    // 129..256 rows: ONE 256-row stripe (W = 1) instead of two 128-row stripes in a 4-stripe workgroup with two idle waves.  (Since r03
    // only what the mini-stripe engine cannot take comes here with such patterns: alphabets of more than 7 symbols, scores beyond the keys.)
    if (max_n > 128 && max_n <= 256) g.rl = 4;
    if (kn.force_rl) g.rl = kn.force_rl == 2 ? 2 : 4;   // experiments only
    if (!keyed) g.rl = 4;   // the plain int32 traceback form exists for RL = 4 only (pair_kernels.hip)
    if ((max_n + 64 * g.rl - 1) / (64 * g.rl) <= 1) g.w = 1;
    // (W = 8 was built and measured in r02: nine waves on a CU's four SIMDs share issue slots, a step goes from 197 to 317
    // cycles -- a workgroup lives on one CU, so four compute waves is the most that keeps one stripe per SIMD)
    (void)keyed_tb;
    if (kn.force_w) g.w = kn.force_w == 1 ? 1 : 4;
    return g;
}
pair_kernel_t pair_fill_fn(PairGeom g, bool local, bool tb, bool sband, bool perm, bool keyed, bool gap0) {
    return pair_fill_kernel_for(g.rl, g.w, local, tb, sband, perm, keyed, gap0);
}
pair_kernel_t pair_tb_fn(PairGeom g, bool local, int walk) { return pair_traceback_kernel_for(g.rl, local, walk); }

size_t tb_band_bytes(uint64_t n, uint64_t m, int rl) {
    const uint64_t stripes = (n + 64 * rl - 1) / (64 * rl);
    return (size_t)(stripes * band_steps(m) * 64 * rl);
}

// Device-side state of one launch of the wavefront (pair) engine: pair descriptors, the global
// stripe-task list, hand-off rows, progress counters, per-stripe bests.
struct PairLaunch {
    DevBuf desc, tasks, rows, progress, best, queue;
    bool from_pool = false;   // take the six buffers from the context's pool (one launch at a time per context: pwa_align*)
    void *p_desc = nullptr, *p_tasks = nullptr, *p_rows = nullptr, *p_progress = nullptr, *p_best = nullptr, *p_queue = nullptr;
    size_t progress_bytes = 0;
    hipError_t take(pwa_ctx* ctx, DevBuf& own, int slot, size_t bytes, void** out) {
        if (bytes == 0) bytes = 16;
        if (from_pool) return cached_workspace(ctx->pool[slot], ctx->pool_bytes[slot], bytes, own, out);
        const hipError_t e = own.alloc(bytes);
        *out = own.p;
        return e;
    }
    PairParams G{};
    PairGeom geom{4, 4};
    bool mini = false;   // the mini-stripe engine (mini_fill.hip.h): mini_ln lanes per pair, geom.rl rows per lane, 64 / mini_ln pairs per wave
    int mini_ln = 16;
    bool perm = false;   // sequences are coded 0..6 (pad 7) and the key constants fit a byte: table-scoring fill kernels
    bool keyed = true;   // traceback fills keep H * 4 + priority (needs |H| < 2^28); false: plain int32 compare-and-select form
    bool gap0 = false;   // global keyed table-scoring fill in gap-shifted coordinates: build() was given gap 0 and scores s - 2 gap
    uint32_t grid = 0;
    uint64_t row_bytes = 0;
    uint64_t n_stripes = 0;
    DevBuf stamps;   // PWA_STAMPS=<file>: per-stripe time stamps of the fill (debugging the stripe pipeline)

    // pd[q].{pat,txt,n,m,tb,sband,res,ops,ops_cap} filled by the caller; this adds the pipeline fields
    int build(pwa_ctx* ctx, std::vector<PairDesc>& pd, int match, int mismatch, int gap, PairGeom g) {
        geom = g;
        const uint64_t rows_per_stripe = 64ull * g.rl;
        std::vector<StripeTask> tl;
        uint64_t rows_i32 = 0, n_stripes_total = 0;
        for (size_t q = 0; q < pd.size(); ++q) {
            const uint64_t ns = ((uint64_t)pd[q].n + rows_per_stripe - 1) / rows_per_stripe;
            const uint64_t nsup = (ns + g.w - 1) / g.w;
            if (tl.size() + nsup >= 0xffffffffull || n_stripes_total + ns >= 0xffffffffull)
                return fail(ctx, PWA_E_CAPACITY, "too many stripe tasks in one launch");
            pd[q].first_task = (uint32_t)tl.size();
            pd[q].first_stripe = (uint32_t)n_stripes_total;
            pd[q].n_stripes = (uint32_t)ns;
            pd[q].row_stride = (uint32_t)align_up((uint64_t)pd[q].m + 64, 64);
            for (uint64_t st = 0; st < nsup; ++st) tl.push_back({(uint32_t)q, (uint32_t)st});
            rows_i32 += (nsup - 1) * pd[q].row_stride;
            n_stripes_total += ns;
        }
        row_bytes = rows_i32 * sizeof(int32_t);
        HIPC(ctx, take(ctx, rows, pwa_ctx::POOL_ROWS, row_bytes, &p_rows));
        uint64_t ro = 0;
        for (auto& d : pd) {
            d.rows = static_cast<int32_t*>(p_rows) + ro;
            const uint64_t nsup = ((uint64_t)d.n_stripes + g.w - 1) / g.w;
            ro += (nsup - 1) * d.row_stride;
        }
        HIPC(ctx, take(ctx, desc, pwa_ctx::POOL_DESC, pd.size() * sizeof(PairDesc), &p_desc));
        HIPC(ctx, ctx->pin[pwa_ctx::PIN_DESC].reserve(pd.size() * sizeof(PairDesc)));
        std::memcpy(ctx->pin[pwa_ctx::PIN_DESC].p, pd.data(), pd.size() * sizeof(PairDesc));
        HIPC(ctx, hipMemcpy(p_desc, ctx->pin[pwa_ctx::PIN_DESC].p, pd.size() * sizeof(PairDesc), hipMemcpyHostToDevice));
        HIPC(ctx, take(ctx, tasks, pwa_ctx::POOL_TASKS, tl.size() * sizeof(StripeTask), &p_tasks));
        HIPC(ctx, ctx->pin[pwa_ctx::PIN_TL].reserve(tl.size() * sizeof(StripeTask)));
        std::memcpy(ctx->pin[pwa_ctx::PIN_TL].p, tl.data(), tl.size() * sizeof(StripeTask));
        HIPC(ctx, hipMemcpy(p_tasks, ctx->pin[pwa_ctx::PIN_TL].p, tl.size() * sizeof(StripeTask), hipMemcpyHostToDevice));
        progress_bytes = align_up(tl.size() * sizeof(uint32_t), 16);
        HIPC(ctx, take(ctx, progress, pwa_ctx::POOL_PROGRESS, progress_bytes, &p_progress));
        HIPC(ctx, take(ctx, best, pwa_ctx::POOL_BEST, std::max<uint64_t>(n_stripes_total, 1) * sizeof(StripeBest), &p_best));
        HIPC(ctx, take(ctx, queue, pwa_ctx::POOL_QUEUE, 64, &p_queue));
        G.pairs = static_cast<PairDesc*>(p_desc);
        G.tasks = static_cast<StripeTask*>(p_tasks);
        G.n_pairs = (uint32_t)pd.size();
        G.n_tasks = (uint32_t)tl.size();
        G.queue = static_cast<uint32_t*>(p_queue);
        G.progress = static_cast<uint32_t*>(p_progress);
        G.best = static_cast<StripeBest*>(p_best);
        G.match = match;
        G.mismatch = mismatch;
        G.gap = gap;
        G.dash = 0x100;   // no symbol: set by the callers that walk for overlaps
        G.stamps = nullptr;
        G.trace_stripe = -1;
        G.trace_base = 0;
        n_stripes = n_stripes_total;
        // Tasks come off the queue in global order, so correctness does not depend on how many workgroups
        // are resident.  One workgroup = W compute waves + 1 helper wave.
        // [gpu] single-stripe batches (4096 pairs 150 x 10k, NW + band): 8 workgroups per CU 4.11 ms, 12 or 16: 3.76 ms (three
        // compute waves per SIMD fill the issue slots two leave open); the HBM-bound SW + score-band batch does not care
        int per_cu = g.w == 1 ? 12 : 3;
        if (ctx->knobs.wg_per_cu > 0) per_cu = ctx->knobs.wg_per_cu;   // experiments only
        grid = (uint32_t)std::min<uint64_t>(tl.size(), (uint64_t)ctx->num_cu * per_cu);
        return PWA_OK;
    }
    // mini-stripe engine: pd = the real pairs first (n_real of them), then empty patterns up to a multiple of four; task t = the
    // pairs 4t .. 4t+3 (the caller orders them so that a task's texts are about equally long)
    int build_mini(pwa_ctx* ctx, std::vector<PairDesc>& pd, uint32_t n_real, int match, int mismatch, int gap, int rl, int ln = 16) {
        mini = true;
        mini_ln = ln;
        geom = PairGeom{rl, 1};
        const size_t ppw = (size_t)(64 / ln);   // pairs per wave: 4, or 1 (one pair per wave: 512- / 1024-row single stripes)
        if (pd.empty() || pd.size() % ppw || pd.size() >= 0xffffffffull || n_real > pd.size() || n_real + ppw - 1 < pd.size())
            return fail(ctx, PWA_E_INVALID, "internal: mini-stripe task list");
        for (size_t q = 0; q < pd.size(); ++q) {
            pd[q].first_task = (uint32_t)(q / ppw);
            pd[q].first_stripe = (uint32_t)q;
            pd[q].n_stripes = 1;
            pd[q].row_stride = 0;
            pd[q].rows = nullptr;
        }
        HIPC(ctx, take(ctx, desc, pwa_ctx::POOL_DESC, pd.size() * sizeof(PairDesc), &p_desc));
        HIPC(ctx, ctx->pin[pwa_ctx::PIN_DESC].reserve(pd.size() * sizeof(PairDesc)));
        std::memcpy(ctx->pin[pwa_ctx::PIN_DESC].p, pd.data(), pd.size() * sizeof(PairDesc));
        HIPC(ctx, hipMemcpy(p_desc, ctx->pin[pwa_ctx::PIN_DESC].p, pd.size() * sizeof(PairDesc), hipMemcpyHostToDevice));
        HIPC(ctx, take(ctx, best, pwa_ctx::POOL_BEST, pd.size() * sizeof(StripeBest), &p_best));
        HIPC(ctx, take(ctx, queue, pwa_ctx::POOL_QUEUE, 64, &p_queue));
        progress_bytes = 0;
        row_bytes = 0;
        G = PairParams{};
        G.pairs = static_cast<PairDesc*>(p_desc);
        G.n_pairs = n_real;
        G.n_tasks = (uint32_t)(pd.size() / ppw);
        G.queue = static_cast<uint32_t*>(p_queue);
        G.best = static_cast<StripeBest*>(p_best);
        G.match = match;
        G.mismatch = mismatch;
        G.gap = gap;
        G.dash = 0x100;
        G.trace_stripe = -1;
        n_stripes = pd.size();
        grid = G.n_tasks;   // (clamped to what the chip holds at launch time, where the kernel is known)
        return PWA_OK;
    }
    // enqueue: zero the queue / progress words, fill, then the walk (or only the end-cell pick)
    int launch(pwa_ctx* ctx, hipStream_t st, bool local, bool tb, int walk, hipEvent_t after_fill, bool sband = false) {
        HIPC(ctx, hipMemsetAsync(p_queue, 0, 16, st));
        if (mini) {
            const pair_kernel_t fill = mini_fill_kernel_for(geom.rl, local, sband, gap0 && !sband && !local, tb, mini_ln);   // tb = false: no band at all
            const pair_kernel_t walk_fn = mini_traceback_kernel_for(geom.rl, local, tb ? walk : (int)WALK_NONE, mini_ln);
            if (!fill || !walk_fn || !perm || !keyed) return fail(ctx, PWA_E_INVALID, "internal: no mini-stripe kernel for this form");
            // Workgroups of four waves (one task each per round); `per_cu` of them per CU, enforced through the dynamic LDS request, so
            // that no CU gets more than its share whatever ran before (mini_fill.hip.h): with ceil(tasks / 4) workgroups for 256 CUs,
            // per_cu = ceil(workgroups / CUs), at most 2; longer task lists run in rounds ([gpu] pairs 150 x 10k: 8192 of them at two
            // waves per SIMD 2.68 ms, 16384 at four 6.61 ms -- 16 k concurrent write streams get 4.0 instead of 4.9 TB/s out of HBM).
            const uint32_t n_wg = (G.n_tasks + kMiniWaves - 1) / kMiniWaves;
            // (band-less fills have no write streams to thin out: four per CU -- [gpu] scores with end cells 2 - 3 % faster than at two)
            const uint32_t cap_per_cu = ctx->knobs.mini_per_cu > 0 ? (uint32_t)std::min(ctx->knobs.mini_per_cu, 5) : (tb ? 2u : 4u);
            const uint32_t per_cu = std::min<uint32_t>(cap_per_cu, (n_wg + (uint32_t)ctx->num_cu - 1) / (uint32_t)ctx->num_cu);
            static const uint32_t kPadKiB[6] = {0, 96, 64, 48, 36, 30};   // more than 160 KiB / (per_cu + 1), at most 160 KiB / per_cu
            const size_t pad_lds = (size_t)kPadKiB[per_cu] * 1024;
            HIPC(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fill), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad_lds));
            const uint32_t g = std::min<uint32_t>(n_wg, (uint32_t)ctx->num_cu * per_cu);
            if (ctx->knobs.debug) std::fprintf(stderr, "[pwa] mini fill: %u tasks, %u workgroups of %d waves, %u per CU (%zu KiB of LDS each)\n", G.n_tasks, g, kMiniWaves, per_cu, pad_lds >> 10);
            hipLaunchKernelGGL(fill, dim3(g), dim3(64 * kMiniWaves), pad_lds, st, G);
            HIPC(ctx, hipGetLastError());
            if (after_fill) HIPC(ctx, hipEventRecord(after_fill, st));
            hipLaunchKernelGGL(walk_fn, dim3(G.n_pairs), dim3(64), 0, st, G);   // one wave per pair
            HIPC(ctx, hipGetLastError());
            return PWA_OK;
        }
        HIPC(ctx, hipMemsetAsync(p_progress, 0, progress_bytes, st));
        if (!ctx->knobs.stamps.empty()) {
            HIPC(ctx, stamps.alloc(n_stripes * 32 + 4 * 8192 * 8));
            HIPC(ctx, hipMemsetAsync(stamps.p, 0, n_stripes * 32 + 4 * 8192 * 8, st));
            G.stamps = stamps.as<unsigned long long>();
            G.trace_base = (uint32_t)(n_stripes * 4);
            G.trace_stripe = ctx->knobs.trace_stripe;
        }
        // (scores / end cells only over a coded arena, keys in range: the keyed chunk without a band -- batch_create_impl sets perm for that)
        const bool noband = !tb && perm && keyed && !sband;
        const pair_kernel_t fill = noband ? pair_fill_kernel_for(geom.rl, geom.w, local, true, false, true, true, gap0 && !local, false)
                                          : pair_fill_fn(geom, local, tb, sband, perm && tb && keyed, keyed, gap0 && tb && keyed && perm && !sband && !local);
        if (!fill) return fail(ctx, PWA_E_INVALID, "internal: no fill kernel for this geometry");
        // A launch with no more multi-stripe workgroups than CUs asks for enough (unused) dynamic LDS that only ONE workgroup
        // fits a CU: a stripe is one wave alone on its SIMD, and every stripe of a pair moves at the pace of the slowest --
        // two workgroups sharing a CU's four SIMDs would slow the whole pipeline
        size_t pad_lds = 0;
        if (geom.w > 1 && grid <= (uint32_t)ctx->num_cu && !ctx->knobs.no_lds_pad) pad_lds = 96 * 1024;   // static (<= 16 KiB) + 96 KiB > half of the CU's 160 KiB
        if (pad_lds) HIPC(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fill), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad_lds));
        hipLaunchKernelGGL(fill, dim3(grid), dim3(64 * (geom.w + 1)), pad_lds, st, G);
        HIPC(ctx, hipGetLastError());
        if (after_fill) HIPC(ctx, hipEventRecord(after_fill, st));
        hipLaunchKernelGGL(pair_tb_fn(geom, local, walk), dim3(G.n_pairs), dim3(64), 0, st, G);   // one wave per pair
        HIPC(ctx, hipGetLastError());
        return PWA_OK;
    }
    // after the stream has been synchronised: did a bounded spin give up?
    int check(pwa_ctx* ctx) {
        if (const char* path = ctx->knobs.stamps.c_str(); !ctx->knobs.stamps.empty() && stamps.p) {
            std::vector<unsigned long long> h(n_stripes * 4);
            HIPC(ctx, hipMemcpy(h.data(), stamps.p, n_stripes * 32, hipMemcpyDeviceToHost));
            if (FILE* f = std::fopen(path, "w")) {
                for (uint64_t k = 0; k < n_stripes; ++k)
                    std::fprintf(f, "%llu %llu %llu %llu %llu\n", (unsigned long long)k, h[4 * k] - h[0], h[4 * k + 3] - h[0], h[4 * k + 1] - h[0], h[4 * k + 2] - h[0]);
                std::fclose(f);
            }
            if (G.trace_stripe >= 0) {
                std::vector<unsigned long long> tr(4 * 8192);
                HIPC(ctx, hipMemcpy(tr.data(), stamps.as<unsigned long long>() + G.trace_base, tr.size() * 8, hipMemcpyDeviceToHost));
                if (FILE* f = std::fopen((std::string(path) + ".trace").c_str(), "w")) {
                    for (int c = 0; c < 8192; ++c)
                        std::fprintf(f, "%d %llu %llu %llu %llu\n", c, tr[c] - h[0], tr[8192 + c] - h[0], tr[2 * 8192 + c] - h[0], tr[3 * 8192 + c] - h[0]);
                    std::fclose(f);
                }
            }
        }
        uint32_t q[2] = {0, 0};
        HIPC(ctx, hipMemcpy(q, p_queue, sizeof q, hipMemcpyDeviceToHost));
        if (q[1] != 0) return fail(ctx, PWA_E_HIP, "stripe pipeline timed out waiting for the stripe above");
        return PWA_OK;
    }
};

}  // namespace

// ------------------------------------------------------------------------------------ batch
struct pwa_batch {
    pwa_ctx* ctx = nullptr;
    int mode = 0;
    uint64_t n_pairs = 0;
    uint64_t cells = 0, padded_cells = 0;
    bool want_end = false;
    // engine 1: register-strip kernels
    bool use_strips = false;        // some pairs run on the register-strip kernels
    bool use_pairs = false;         // some (or, with end cells / scorings the strips cannot pad for, all) pairs run on the stripe engine
    const BatchKernelEntry* kern = nullptr;
    BatchParams bp{};
    bool affine = false, nwdist = false, single_strip = false, paired = false, lanes = false;
    int32_t aff_go = 0, aff_ge = 0, aff_neg = 0;
    uint32_t grid = 0;
    DevBuf arena, tasks, slot_poff, slot_plen, slot_out, slot_toff, slot_tlen, lane_text, hand, queue, scores;
    // engine 2: wavefront kernels without traceback band (exact end cells, any scoring)
    PairLaunch pl;
    std::vector<std::unique_ptr<PairLaunch>> mini;   // engine 3: mini-stripe launches without a band, one per row class (short patterns routed off the strips)
    DevBuf pair_res;
    uint64_t n_live = 0;            // pairs that reach a kernel (n > 0 and m > 0)
    std::vector<uint32_t> live_idx; // pairs off the strips: pair index of result slot q (stripe engine's pairs first, then the mini launches')
    std::vector<int32_t> host_scores;   // trivial pairs resolved on the host
    std::vector<uint32_t> host_end_i, host_end_j;
    std::string kernel_name;
    int32_t* ext_scores = nullptr;      // caller-owned device score vector (pwa_batch_set_d_scores)
    static constexpr int kRing = 64;          // event pairs of the most recent runs
    hipEvent_t ev0[kRing] = {}, ev1[kRing] = {};
    uint64_t n_runs = 0;
    bool ran = false;
};

extern "C" {

const char* pwa_version(void) { return "pwalign 0.1 gfx950"; }

// Host-only checks of the scheduler's sorting helpers (include/pwalign.h): tests call this on machines without a GPU.
int pwa_selftest_host(uint32_t seed) try {
    uint64_t x = 0x9e3779b97f4a7c15ull ^ seed;
    auto rnd = [&]() {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        return x;
    };
    int check = 0;
    for (const size_t n : {size_t(0), size_t(1), size_t(2), size_t(1000), size_t(70000), size_t(300000), size_t(1) << 20}) {
        for (const size_t buckets : {size_t(1), size_t(3), size_t(4352), size_t(65536), size_t(200000)}) {
            ++check;
            std::vector<uint32_t> key(n), idx(n), tmp, want(n);
            for (size_t i = 0; i < n; ++i) {
                key[i] = (uint32_t)(rnd() % buckets);
                idx[i] = (uint32_t)i;
            }
            want = idx;
            std::stable_sort(want.begin(), want.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
            counting_sort(idx, tmp, buckets, [&](uint32_t v) { return (size_t)key[v]; });   // (threaded from 2^18 elements, <= 2^16 buckets)
            if (idx != want) return check;
        }
        for (const uint64_t span : {uint64_t(1), uint64_t(7), uint64_t(3000), uint64_t(1) << 33}) {   // the last one takes the radix path
            ++check;
            std::vector<uint64_t> len(n);
            std::vector<uint32_t> idx(n), want(n);
            for (size_t i = 0; i < n; ++i) {
                len[i] = 5 + rnd() % span;
                idx[i] = (uint32_t)i;
            }
            want = idx;
            std::stable_sort(want.begin(), want.end(), [&](uint32_t a, uint32_t b) { return len[a] > len[b]; });
            sort_by_length_desc(idx, [&](uint32_t v) { return len[v]; });
            if (idx != want) return check;
        }
        {
            ++check;
            std::vector<uint64_t> key(n);
            std::vector<uint32_t> idx(n), want(n);
            for (size_t i = 0; i < n; ++i) {
                key[i] = rnd() >> (rnd() % 50);
                idx[i] = (uint32_t)i;
            }
            want = idx;
            const std::vector<uint64_t> key0 = key;
            std::stable_sort(want.begin(), want.end(), [&](uint32_t a, uint32_t b) { return key0[a] < key0[b]; });
            radix_sort_by_key(key, idx);
            if (idx != want) return check;
        }
    }
    return 0;
} catch (...) {
    return -1;
}

const char* pwa_strerror(int code) {
    switch (code) {
        case PWA_OK: return "ok";
        case PWA_E_INVALID: return "invalid argument";
        case PWA_E_NODEVICE: return "no usable gfx950 device";
        case PWA_E_HIP: return "HIP runtime error";
        case PWA_E_NOMEM: return "out of memory";
        case PWA_E_CAPACITY: return "capacity exceeded";
        case PWA_E_IO: return "cannot open or read file";
        default: return "unknown error";
    }
}

int pwa_ctx_create(int device, pwa_ctx** out) {
    if (!out) return PWA_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return PWA_E_NODEVICE;
    if (device < 0 || device >= count) return PWA_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return PWA_E_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return PWA_E_NODEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return PWA_E_NODEVICE;   // kernels exist for gfx950 only
    pwa_ctx* c = new (std::nothrow) pwa_ctx();
    if (!c) return PWA_E_NOMEM;
    c->knobs.read();   // the only place the library's switches are read from the environment
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return PWA_E_HIP;
    }
    for (auto& e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            pwa_ctx_destroy(c);
            return PWA_E_HIP;
        }
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->copy_ev[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->copy_ev[1], hipEventDisableTiming) != hipSuccess || hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->aux_ev[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->aux_ev[1], hipEventDisableTiming) != hipSuccess) {
        pwa_ctx_destroy(c);
        return PWA_E_HIP;
    }
    *out = c;
    return PWA_OK;
}

void pwa_ctx_destroy(pwa_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (auto& e : c->copy_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (auto& e : c->aux_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->band_cache) (void)hipFree(c->band_cache);
    if (c->sband_cache) (void)hipFree(c->sband_cache);
    if (c->hand_cache) (void)hipFree(c->hand_cache);
    for (void* q : c->pool)
        if (q) (void)hipFree(q);
    for (auto& f : c->free_list) (void)hipFree(f.first);
    delete c;
}

const char* pwa_last_error(const pwa_ctx* c) { return c ? c->err.c_str() : "null context"; }

int pwa_ctx_set_score_band(pwa_ctx* c, int on) {
    if (!c) return PWA_E_INVALID;
    c->score_band = on != 0;
    return PWA_OK;
}

// ---------------------------------------------------------------------------- batch: create
} // extern "C" (reopened below): the shared implementation has C++ linkage
enum { KIND_LINEAR = 0, KIND_AFFINE = 1, KIND_NWDIST = 2 };
static int batch_create_impl(pwa_ctx* ctx, int mode, int match, int mismatch, int gap, int kind, int gap_extend,
                             const uint8_t* seq_bytes, const uint64_t* seq_off, uint32_t n_seq, const uint32_t* pair_a,
                             const uint32_t* pair_b, uint64_t n_pairs, int want_end_cells, pwa_batch** out) try {
    if (!ctx || !out) return PWA_E_INVALID;
    *out = nullptr;
    const bool affine = kind == KIND_AFFINE, nwdist = kind == KIND_NWDIST;
    if ((affine || nwdist) && want_end_cells) return fail(ctx, PWA_E_INVALID, "end cells are not defined for this pass");
    if (mode != PWA_MODE_NW && mode != PWA_MODE_SW) return fail(ctx, PWA_E_INVALID, "unknown mode");
    if (!seq_off || (n_pairs && (!pair_a || !pair_b))) return fail(ctx, PWA_E_INVALID, "null input");
    if (n_seq && !seq_bytes && seq_off[n_seq] != 0) return fail(ctx, PWA_E_INVALID, "null seq_bytes");
    if (n_pairs >= 0xffffffffull) return fail(ctx, PWA_E_CAPACITY, "more than 2^32-2 pairs in one batch");
    for (uint32_t s = 0; s < n_seq; ++s)
        if (seq_off[s + 1] < seq_off[s]) return fail(ctx, PWA_E_INVALID, "seq_off not monotone");
    HIPC(ctx, hipSetDevice(ctx->device));
    const bool dbg = ctx->knobs.debug;
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {   // PWA_DEBUG: host-side time between marks
        if (!dbg) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pwa] create: %-24s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
        if (ctx->knobs.probe) {   // how long does a kernel submission take at this point?
            hipLaunchKernelGGL(pwa_nop_kernel, dim3(1), dim3(64), 0, ctx->stream, (int*)nullptr);
            (void)hipStreamSynchronize(ctx->stream);
            const auto t2 = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[pwa]     probe after it: nop kernel + sync %8.3f ms\n", std::chrono::duration<double, std::milli>(t2 - now).count());
            t_last = t2;
        }
    };

    pwa_batch* b = new (std::nothrow) pwa_batch();
    if (!b) return fail(ctx, PWA_E_NOMEM, "host allocation");
    struct Guard {
        pwa_batch* b;
        ~Guard() { if (b) pwa_batch_destroy(b); }
    } guard{b};
    b->ctx = ctx;
    for (DevBuf* d : {&b->arena, &b->tasks, &b->slot_poff, &b->slot_plen, &b->slot_out, &b->slot_toff, &b->slot_tlen, &b->lane_text, &b->hand, &b->queue,
                      &b->scores, &b->pair_res, &b->pl.desc, &b->pl.tasks, &b->pl.rows, &b->pl.progress, &b->pl.best, &b->pl.queue})
        d->pool = ctx;   // released buffers are kept for the next batch of this context
    b->mode = mode;
    b->n_pairs = n_pairs;
    b->want_end = want_end_cells != 0;
    const bool local = mode == PWA_MODE_SW;

    auto slen = [&](uint32_t s) -> uint64_t { return seq_off[s + 1] - seq_off[s]; };

    // ---- one pass over the pair list: index check, and pairs with an empty side never reach a kernel (hw2.cpp: loops
    // 138/205 do not run) -- they are resolved here
    b->host_scores.assign(n_pairs, 0);
    if (b->want_end) {
        b->host_end_i.assign(n_pairs, 0);
        b->host_end_j.assign(n_pairs, 0);
    }
    std::vector<uint32_t> live(n_pairs);
    uint64_t max_n = 0, max_m = 0, n_live = 0;
    bool any_trivial_score = false;
    for (uint64_t k = 0; k < n_pairs; ++k) {
        if (pair_a[k] >= n_seq || pair_b[k] >= n_seq) return fail(ctx, PWA_E_INVALID, "pair index out of range");
        const uint64_t n = slen(pair_a[k]), m = slen(pair_b[k]);
        if (n == 0 || m == 0) {
            if (nwdist) {   // hw4.cpp:21-28 + 146-152: an all-gap alignment, every column counts
                b->host_scores[k] = (int32_t)(n + m);
            } else if (affine) {   // hw3.cpp:39-52: V[0][0] = 0, F[n][0] = Go + Ge(n-1), E[0][m] = Go + Ge(m-1)
                b->host_scores[k] = (n + m == 0) ? 0 : (int32_t)((uint32_t)gap + (uint32_t)wrap_mul((int64_t)(n + m - 1), gap_extend));
            } else if (!local) b->host_scores[k] = wrap_mul((int64_t)(n + m), gap);   // dp[n][0] / dp[0][m], hw2.cpp:125-136
            any_trivial_score = any_trivial_score || b->host_scores[k] != 0;
            if (b->want_end && !local) {
                b->host_end_i[k] = (uint32_t)n;
                b->host_end_j[k] = (uint32_t)m;
            }
            continue;
        }
        if (n > 0x7fffffc0ull || m > 0x7fffffc0ull) return fail(ctx, PWA_E_CAPACITY, "sequence longer than 2^31");
        live[n_live++] = (uint32_t)k;
        b->cells += n * m;
        max_n = std::max(max_n, n);
        max_m = std::max(max_m, m);
    }
    live.resize(n_live);
    b->n_live = live.size();

    HIPC(ctx, b->scores.alloc(std::max<uint64_t>(n_pairs, 1) * sizeof(int32_t)));
    if (any_trivial_score) HIPC(ctx, upload_via_bounce(ctx, b->scores.p, b->host_scores.data(), n_pairs * sizeof(int32_t)));
    else {   // (on the copy stream and waited for: a run may be enqueued on any stream afterwards -- and the context's own stream may be
             // busy with the previous batch's run, which preparing this one must not wait for)
        HIPC(ctx, hipMemsetAsync(b->scores.p, 0, std::max<uint64_t>(n_pairs, 1) * sizeof(int32_t), ctx->copy_stream));
        HIPC(ctx, hipStreamSynchronize(ctx->copy_stream));
    }
    HIPC(ctx, b->queue.alloc(64));   // (the event ring of the runs is created run by run: pwa_batch_run)
    if (live.empty()) {
        b->kernel_name = "none";
        guard.b = nullptr;
        *out = b;
        return PWA_OK;
    }

    // ---- which sequences play which role; text alphabet
    std::vector<uint8_t> is_text(n_seq, 0), is_used(n_seq, 0);
    for (uint32_t k : live) {
        is_text[pair_b[k]] = 1;
        is_used[pair_a[k]] = is_used[pair_b[k]] = 1;
    }
    bool present[256] = {false};
    {
        bool part[16][256] = {};
        for_seq_ranges(seq_off, n_seq, [&](uint32_t s0, uint32_t s1, int t) {
            bool* mine = part[t];
            for (uint32_t s = s0; s < s1; ++s)
                if (is_text[s])
                    for (uint64_t o = seq_off[s]; o < seq_off[s + 1]; ++o) mine[seq_bytes[o]] = true;
        });
        for (int t = 0; t < 16; ++t)
            for (int v = 0; v < 256; ++v) present[v] |= part[t][v];
    }
    int n_alpha = 0;
    int code_of[256];
    int absent_byte = -1;
    for (int v = 0; v < 256; ++v) {
        if (present[v]) code_of[v] = n_alpha++;
        else {
            code_of[v] = -1;
            if (absent_byte < 0) absent_byte = v;
        }
    }

    // LANES kernels also pad TEXTS (columns past a lane's own text): that symbol must match no pattern symbol and
    // must differ from the pattern pad, or padded rows would "match" padded columns.
    bool pattern_has[256] = {false};
    int text_pad_byte = -1;   // raw-byte (SC_CMP) form; the coded (SC_PERM) form uses code 6 when the alphabet leaves it free
    {
        bool in_pattern[256] = {false};
        std::vector<uint8_t> is_pat(n_seq, 0);
        for (uint64_t k = 0; k < n_pairs; ++k) is_pat[pair_a[k]] = 1;
        bool part[16][256] = {};
        for_seq_ranges(seq_off, n_seq, [&](uint32_t s0, uint32_t s1, int t) {
            bool* mine = part[t];
            for (uint32_t s = s0; s < s1; ++s)
                if (is_pat[s])
                    for (uint64_t o = seq_off[s]; o < seq_off[s + 1]; ++o) mine[seq_bytes[o]] = true;
        });
        for (int t = 0; t < 16; ++t)
            for (int v = 0; v < 256; ++v) in_pattern[v] |= part[t][v];
        for (int v = 255; v >= 0 && text_pad_byte < 0; --v)
            if (!in_pattern[v] && v != absent_byte) text_pad_byte = v;
        for (int v = 0; v < 256; ++v) pattern_has[v] = in_pattern[v];
    }

    // ---- engine choice.  The strip engine pads short patterns with rows that match nothing; for SW
    // those rows can only hold values <= real rows if mismatch <= 0 and gap <= 0.
    const bool strips_ok = affine || nwdist || (!b->want_end && (!local || (mismatch <= 0 && gap <= 0)));
    b->affine = affine;
    b->nwdist = nwdist;
    auto fits8 = [](int v) { return v >= -128 && v <= 127; };
    b->use_strips = strips_ok;

    // ---- device arena: every used sequence, 16-byte aligned, as symbols of the chosen coding
    int score_path = SC_CMP;
    int kmode = local ? BM_SW : BM_NW;
    int tab_match = match, tab_mismatch = mismatch;
    if (nwdist) {
        kmode = BM_DIST;
        tab_match = match - gap;       // the rows store H + gap: the diagonal consumer takes the gap back out
        tab_mismatch = mismatch - gap;
        if (n_alpha <= 7 && fits8(tab_match) && fits8(tab_mismatch)) score_path = SC_PERM;
        if (score_path == SC_CMP && absent_byte < 0)
            return fail(ctx, PWA_E_INVALID, "distance pass: the texts use all 256 byte values, no padding symbol left");
        // packed (H, dist) keys: dist in 12 bits, H in the 18 above (batch_nwdist.hip.h)
        const int64_t amax = std::max<int64_t>({std::llabs((long long)match), std::llabs((long long)mismatch),
                                                std::llabs((long long)gap)});
        if (max_n + max_m <= 4000 && (int64_t)(max_n + max_m + 2) * amax < 65536 && !ctx->knobs.no_packed_dist) {
            kmode = BM_DISTP;
            tab_match = match;       // the packed kernel takes hw4's own three scores
            tab_mismatch = mismatch;
        }
    } else if (affine) {
        // Ge(i+j)-shifted form when every value stays far inside int32 (the sentinels are -2^29 there)
        const int64_t amax = std::max<int64_t>({std::llabs((long long)match), std::llabs((long long)mismatch),
                                                std::llabs((long long)gap), std::llabs((long long)gap_extend)});
        const bool shift_ok = (int64_t)(max_n + max_m + 4) * amax * 4 < (1ll << 27);
        kmode = shift_ok ? BM_AFFS : BM_AFF;
        b->aff_go = gap;
        b->aff_ge = gap_extend;
        b->aff_neg = shift_ok ? -(1 << 29) : std::numeric_limits<int32_t>::min() / 2;   // hw3.cpp:16
        if (shift_ok) {
            tab_match = match - 2 * gap_extend;
            tab_mismatch = mismatch - 2 * gap_extend;
        }
        if (n_alpha <= 7 && fits8(tab_match) && fits8(tab_mismatch)) score_path = SC_PERM;
        if (score_path == SC_CMP && absent_byte < 0)
            return fail(ctx, PWA_E_INVALID, "affine pass: the texts use all 256 byte values, no padding symbol left");
    } else if (b->use_strips) {
        if (!local) {
            // gap-shifted NW: G = H - g(i+j) needs every |value| to stay far inside int32
            const int64_t amax = std::max<int64_t>({std::llabs((long long)match), std::llabs((long long)mismatch),
                                                    std::llabs((long long)gap)});
            const int64_t s_match = (int64_t)match - 2 * (int64_t)gap, s_mis = (int64_t)mismatch - 2 * (int64_t)gap;
            if ((int64_t)(max_n + max_m + 4) * amax * 4 < (1ll << 30) && fits8((int)s_match) && fits8((int)s_mis)) {
                kmode = BM_NWG;
                tab_match = (int)s_match;
                tab_mismatch = (int)s_mis;
            }
        }
        if (n_alpha <= 7 && fits8(tab_match) && fits8(tab_mismatch)) score_path = SC_PERM;
        else if (kmode == BM_NWG && n_alpha > 7) { /* compare path works in G space too */ }
        if (score_path == SC_CMP && absent_byte < 0) b->use_strips = false;   // no byte left to pad with
    }

    std::vector<uint64_t> aoff(n_seq, 0);
    uint64_t arena_bytes = 0;
    for (uint32_t s = 0; s < n_seq; ++s)
        if (is_used[s]) {
            aoff[s] = arena_bytes;
            arena_bytes += align_up(slen(s) + 1, 16);
        }
    arena_bytes += 512;   // slack: strips and text words are over-read, never over-used
    if (arena_bytes >= 0xffffffffull) return fail(ctx, PWA_E_CAPACITY, "sequence arena exceeds 4 GiB");
    // (a scores pass that wants end cells runs wholly off the strips: its arena is coded whenever the alphabet allows, for the mini-stripe
    // kernels -- the stripe engine's compare form is the same on codes, a pattern-only symbol is code 7 and equals no text code)
    const bool code_for_end_cells = b->want_end && !affine && !nwdist && n_alpha <= 7 && ctx->knobs.tb_engine != 0;
    const bool arena_coded = (b->use_strips && score_path == SC_PERM) || code_for_end_cells;
    // Short patterns that a scores pass routes away from the strips run on the mini-stripe engine WITHOUT a band (mini_fill.hip.h,
    // BAND = false: four pairs per wave) where it applies: coded arena, keyed cells in range, table constants in a byte.
    bool mini_scores = false, mini_gap0 = false;
    if (arena_coded && !affine && !nwdist && ctx->knobs.tb_engine != 0 && tb_range_ok(max_n + max_m, match, mismatch, gap, local ? 26 : 28)) {
        const int64_t kdm = ((int64_t)match - gap) * 4 + 2, kdx = ((int64_t)mismatch - gap) * 4 + 2;
        mini_scores = kdm <= 127 && kdm >= -126 && kdx <= 127 && kdx >= -126;
        if (mini_scores && !local) {
            const int64_t km = ((int64_t)match - 2 * (int64_t)gap) * 4 + 1, kx = ((int64_t)mismatch - 2 * (int64_t)gap) * 4 + 1;
            const int64_t amax = std::max<int64_t>({std::llabs((long long)match), std::llabs((long long)mismatch), std::llabs((long long)gap), 1});
            mini_gap0 = km <= 127 && km >= -126 && kx <= 127 && kx >= -126 && (max_n + max_m + 2) <= (1ull << 27) / (uint64_t)amax;
        }
    }
    auto mini_rl_of = [&](uint64_t n) -> int {   // rows per lane of the mini-stripe class that holds an n-row pattern, 0: none
        if (mini_scores && n <= 256)
            for (const int rl : kMiniRL)
                if (n <= (uint64_t)(16 * rl)) return rl;
        return 0;
    };
    {
        const bool coded = arena_coded;
        uint8_t code8[256];
        for (int v = 0; v < 256; ++v) code8[v] = (uint8_t)(code_of[v] >= 0 ? code_of[v] : 7);
        HIPC(ctx, b->arena.alloc(arena_bytes));
        HIPC(ctx, build_arena(ctx, b->arena.p, arena_bytes, seq_bytes, seq_off, n_seq, is_used, aoff, coded ? code8 : nullptr,
                              !present[0] && !pattern_has[0]));
    }

    mark("validate + arena upload");
    b->padded_cells = 0;
    std::vector<uint32_t> live_pairs_engine;   // the pairs that run on the stripe engine (all of them when the strips cannot serve the list)
    if (!b->use_strips) live_pairs_engine = live;
    if (b->use_strips) {
        // ---- wave tasks: pairs grouped by text, patterns sorted by length, 64 per wave
        std::vector<uint32_t> order(live);   // ascending pair index: the stable sorts below keep it as the last key
        {
            // text ascending, pattern length descending.  Stable counting sorts, least significant key first: two linear passes
            // per key, and the length pass is skipped when every pattern has the same length (a million-pair cross product:
            // ~3 ms [gpu box] against 14 ms for the 64-bit radix sort, which stays as the fallback for huge key ranges)
            uint64_t lmin = ~0ull, lmax = 0;
            for (const uint32_t k : order) {
                const uint64_t l = slen(pair_a[k]);
                lmin = std::min(lmin, l);
                lmax = std::max(lmax, l);
            }
            // (the histograms have n_seq + 1 and lmax - lmin + 1 entries whatever the list's size: a short list over a large
            // FASTA index, or with one outlier length, is cheaper through the radix sort)
            if (lmax - lmin < (1u << 22) && n_seq <= (1u << 24) && (uint64_t)n_seq <= 4 * (uint64_t)order.size() + 65536 &&
                lmax - lmin <= 4 * (uint64_t)order.size() + 65536) {
                std::vector<uint32_t> tmp;
                if (lmax != lmin) counting_sort(order, tmp, (size_t)(lmax - lmin + 1), [&](uint32_t k) { return (size_t)(lmax - slen(pair_a[k])); });
                counting_sort(order, tmp, (size_t)n_seq, [&](uint32_t k) { return (size_t)pair_b[k]; });
            } else {
                std::vector<uint64_t> key(order.size());   // text ascending, pattern length descending (lengths < 2^31)
                for (size_t o = 0; o < order.size(); ++o)
                    key[o] = ((uint64_t)pair_b[order[o]] << 32) | (uint64_t)(0x7fffffffu - (uint32_t)slen(pair_a[order[o]]));
                radix_sort_by_key(key, order);
            }
        }
        struct HostTask {
            uint32_t text, first, count;
            uint64_t maxlen;   // longest pattern of the task
            uint64_t m;        // text length (LANES: the longest text of the task)
        };
        std::vector<HostTask> ht;
        for (size_t p = 0; p < order.size();) {
            size_t q = p;
            while (q < order.size() && q - p < 64 && pair_b[order[q]] == pair_b[order[p]]) ++q;
            ht.push_back({pair_b[order[p]], (uint32_t)p, (uint32_t)(q - p), slen(pair_a[order[p]]), slen(pair_b[order[p]])});
            p = q;
        }
        // ---- lists whose pairs share few texts (the reference's own loop pairs pattern i with reference i,
        // hw2.cpp:328-338) would leave most lanes of a text-grouped wave empty: give every lane its own text
        // instead (LANES kernels, local alignment only).  Pairs are sorted so that a wave's 64 pairs need about
        // the same number of strips and columns; a wave runs max(strips) x max(columns) of its lanes.
        bool kernels_have_lanes = false;
        size_t n_kernels = 0;
        const BatchKernelEntry* const kernels = batch_kernel_table(&n_kernels);
        for (size_t ki = 0; ki < n_kernels; ++ki) kernels_have_lanes |= (kernels[ki].fn_lanes != nullptr && kernels[ki].score == score_path);
        const bool text_pad_ok = score_path == SC_PERM ? n_alpha <= 6 : text_pad_byte >= 0;
        const bool underfilled = ht.size() * 64 > order.size() * 3 / 2 + 64;
        b->lanes = kmode == BM_SW && !affine && !nwdist && kernels_have_lanes && text_pad_ok && underfilled &&
                   ctx->knobs.paired < 0;   // (the opt-in experiment wins)
        // global alignment: right-aligned texts, front-padded with a code the table scores like a gap (batch_scores.hip.h).
        // Needs the gap-shifted form, a coded alphabet with codes 4..7 free, every pattern symbol inside it (a
        // pattern-only symbol shares code 7 with the pad rows), g <= 0 and -g in a table byte.
        bool patterns_inside = true;
        for (int v = 0; v < 256; ++v) patterns_inside = patterns_inside && (!pattern_has[v] || present[v]);
        const bool lanes_nw = kmode == BM_NWG && !affine && !nwdist && score_path == SC_PERM && n_alpha <= 4 && patterns_inside &&
                              gap <= 0 && fits8(-gap) && underfilled;
        b->lanes = b->lanes || lanes_nw;
        if (ctx->knobs.force_lanes >= 0) b->lanes = b->lanes && ctx->knobs.force_lanes != 0;   // experiments only
        if (b->lanes) {
            order = live;
            std::vector<uint64_t> key(order.size());   // nominal strips descending, then text length descending
            for (size_t o = 0; o < order.size(); ++o) {
                const uint64_t strips = (slen(pair_a[order[o]]) + 75) / 76;
                key[o] = ((0x7fffffffull - strips) << 32) | (uint64_t)(0x7fffffffu - (uint32_t)slen(pair_b[order[o]]));
            }
            radix_sort_by_key(key, order);
            ht.clear();
            for (size_t p = 0; p < order.size(); p += 64) {
                const size_t q = std::min(order.size(), p + 64);
                uint64_t mn = 0, mm = 0;
                for (size_t o = p; o < q; ++o) {
                    mn = std::max(mn, slen(pair_a[order[o]]));
                    mm = std::max(mm, slen(pair_b[order[o]]));
                }
                ht.push_back({pair_b[order[p]], (uint32_t)p, (uint32_t)(q - p), mn, mm});
            }
        }
        mark("sort + group pairs");
        // ---- strip height: least padded work, ties to the taller strip
        int bestR = 0, best_mode = kmode;
        const int kmode_asked = kmode;
        auto choose_strip_height = [&]() {
        bestR = 0;
        best_mode = kmode_asked;
        long double best_cost = -1;
        const int force = ctx->knobs.force_r, force_mode = ctx->knobs.force_mode;   // experiments only
        for (size_t ki = 0; ki < n_kernels; ++ki) {
            const BatchKernelEntry& e = kernels[ki];
            // SW has two forms: BM_SW (R registers per lane, 5.0 VALU per cell) and BM_SWS (2R registers, 4.06)
            const bool mode_ok = e.mode == kmode_asked || (kmode_asked == BM_SW && e.mode == BM_SWS);
            if (!mode_ok || e.score != score_path) continue;
            if (b->lanes && !e.fn_lanes) continue;
            const int R = e.R;
            if (force && force != R) continue;
            if (force_mode >= 0 && force_mode != e.mode) continue;
            long double w = e.mode == BM_SWS ? 4.06L : (e.mode == BM_SW ? 5.02L : 1.0L);   // VALU per cell
            // affine strips of more than 40 rows run 2 instead of 3 waves per SIMD: [gpu] all pairs of 1024 x 1000 take
            // 93.2 ms at R = 52 against 89.2 ms at R = 32 for the same padded cells
            if ((e.mode == BM_AFF || e.mode == BM_AFFS) && R > 40) w *= 1.045L;
            // evaluated cells + the strip hand-off priced at ~2 cells per column and strip boundary ([gpu]: the
            // 1000-row affine pass is equally fast at R = 32 and 52 but moves 37 % fewer HBM bytes at 52)
            long double cost = 0;
            for (const auto& t : ht) {
                const uint64_t strips = (t.maxlen + R - 1) / R;
                cost += (long double)(strips * R + 2 * (strips - 1)) * (long double)t.m * 64.0L;
            }
            cost *= w;
            if (best_cost < 0 || cost < best_cost || (cost == best_cost && R > bestR)) {
                best_cost = cost;
                bestR = R;
                best_mode = e.mode;
            }
        }
        };
        choose_strip_height();
        if (bestR == 0) return fail(ctx, PWA_E_INVALID, "internal: no kernel instantiation");

        // ---- work-aware routing (r03).  The strip engine is the cheaper one per cell (lane = pair, 2.5 - 5 VALU per cell) but a wave
        // task is one wave running strips x columns on its own: a list of few long pairs -- ONE 10k x 10k pair is 105 strips x 2500
        // column blocks x 1560 instructions on one lane of one wave, [gpu] 779 ms against 1.67 ms on the stripe engine -- or of few
        // wave tasks (4096 pairs 150 x 10k = 64 tasks: 11.6 ms against 4.3 ms) leaves the chip idle.  The stripe engine spreads a
        // pair over ceil(n / (64 RL)) waves that sweep anti-diagonals (~100 ns per step of 64 RL cells per SIMD).  Both costs are
        // estimated from the task list with constants measured on the GPU (profiles/r03_route_probe.txt), tasks are moved to the
        // stripe engine in two candidate orders (largest strip task first; cheapest-to-move per unit of strip work first) and
        // the split with the smallest estimated total -- the two launches run one after the other on the run's stream -- wins.
        // Pairs are independent (hw2.cpp:328-338) and both engines are exact, so a split changes no result.
        std::vector<uint32_t> pair_list;   // pair indices routed to the stripe engine
        if (!affine && !nwdist && ctx->knobs.scores_route != 0) {
            const size_t nt0 = ht.size();
            const double vpc = (best_mode == BM_SWS ? 4.06 : best_mode == BM_SW ? 5.02 : best_mode == BM_NWG ? 2.53 : 4.5) + (score_path == SC_CMP ? 2.0 : 0.0);
            constexpr double kLoneNs = 1.9, kSimdNs = 1.63, kSimds = 1024.0;                // ns per wave instruction: one wave alone / a SIMD with two
            // stripe engine: per step and SIMD; per stripe of pipeline lag; per step of a pair alone -- the keyed chunk without a band (coded
            // arena, keys in range) or the plain step ([gpu] r03_route_probe.txt: one 10k x 10k pair 1.26 / 1.01 ms, 64 pairs 3.45 / 2.05 ms)
            const double step_ns = mini_scores ? (local ? 70.0 : 42.0) : (local ? 105.0 : 75.0), lag_us = 9.0,
                         lone_step_ns = mini_scores ? (local ? 55.0 : 30.0) : 100.0;
            std::vector<double> I(nt0), S(nt0), L(nt0);   // strip instructions / stripe-side work (ns x SIMD) / longest single-pair latency (us) of a task
            double I_total = 0, I_max = 0;
            uint64_t filled = 0;
            for (size_t t = 0; t < nt0; ++t) {
                const uint64_t strips = (ht[t].maxlen + bestR - 1) / bestR;
                I[t] = (double)strips * (double)((ht[t].m + 3) / 4) * (4.0 * bestR * vpc);
                I_total += I[t];
                I_max = std::max(I_max, I[t]);
                filled += ht[t].count;
            }
            // nothing to route when the strips' waves are many, full, and none of them dominates: per cell the strips are the cheapest engine
            // by 2x and more, so no task can gain by leaving (and the per-pair estimates below cost ~3 ms for a million pairs)
            const bool strips_fit = nt0 >= 4096 && filled * 10 >= (uint64_t)nt0 * 64 * 9 && I_max * kLoneNs * 4 < I_total / kSimds * kSimdNs &&
                                    ctx->knobs.scores_route < 0;
            for (size_t t = 0; t < nt0 && !strips_fit; ++t) {
                double steps = 0, lat = 0;
                for (uint32_t l = 0; l < ht[t].count; ++l) {
                    const uint32_t k = order[ht[t].first + l];
                    const uint64_t n = slen(pair_a[k]), m = slen(pair_b[k]);
                    if (const int mrl = mini_rl_of(n)) {   // mini-stripe engine, no band: a quarter of a wave, (17 + 5 | 7.3 RL) instructions per step
                        const double mstep = (17.0 + (local ? 7.3 : 5.0) * mrl) * 1.37;   // [gpu] 92 ns per step for RL = 10, global (tools/probes/mini_mix.hip)
                        steps += (double)(m + 15) * mstep / 4.0 / 1.5;                    // (two waves per SIMD: ~1.9 x one wave's throughput)
                        lat = std::max(lat, (double)(m + 15) * mstep * 1e-3);
                        continue;
                    }
                    const PairGeom g = choose_geom(ctx->knobs, n);
                    const double stripes = (double)((n + 64 * g.rl - 1) / (64 * g.rl));
                    steps += stripes * (double)(m + 63) * step_ns;
                    lat = std::max(lat, stripes * lag_us + (double)(m + 63) * lone_step_ns * 1e-3);
                }
                S[t] = steps;
                L[t] = lat;
            }
            auto evaluate = [&](const std::vector<uint32_t>& ord, size_t& best_k) -> double {
                // tasks ord[0 .. k-1] move; suffix maxima of I over the tasks that stay
                std::vector<double> sufmax(nt0 + 1, 0.0);
                for (size_t k = nt0; k-- > 0;) sufmax[k] = std::max(sufmax[k + 1], I[ord[k]]);
                double best_t = -1, moved_I = 0, moved_S = 0, moved_L = 0;
                for (size_t k = 0; k <= nt0; ++k) {
                    const double ts = std::max(sufmax[k] * kLoneNs, (I_total - moved_I) / kSimds * kSimdNs) * 1e-3;              // us
                    const double tp = k ? std::max(moved_L, moved_S / kSimds * 1e-3) + 15.0 : 0.0;                                 // us (+ two more launches)
                    const double tt = ts + tp;
                    if (best_t < 0 || tt < best_t) {
                        best_t = tt;
                        best_k = k;
                    }
                    if (k < nt0) {
                        moved_I += I[ord[k]];
                        moved_S += S[ord[k]];
                        moved_L = std::max(moved_L, L[ord[k]]);
                    }
                }
                return best_t;
            };
            std::vector<uint32_t> ordA(nt0), ordB(nt0);
            std::iota(ordA.begin(), ordA.end(), 0u);
            ordB = ordA;
            if (!strips_fit) {
                std::stable_sort(ordA.begin(), ordA.end(), [&](uint32_t x, uint32_t y) { return I[x] > I[y]; });
                std::stable_sort(ordB.begin(), ordB.end(), [&](uint32_t x, uint32_t y) { return S[x] * I[y] < S[y] * I[x]; });   // S / I ascending
            }
            size_t kA = 0, kB = 0;
            const double tA = strips_fit ? 0.0 : evaluate(ordA, kA), tB = strips_fit ? 0.0 : evaluate(ordB, kB);
            const std::vector<uint32_t>& ord = tA <= tB ? ordA : ordB;
            size_t kmove = strips_fit ? 0 : (tA <= tB ? kA : kB);
            if (ctx->knobs.scores_route == 1) kmove = nt0;   // tests: everything on the stripe engine
            if (dbg) std::fprintf(stderr, "[pwa] route: %zu of %zu wave tasks to the stripe engine (estimates: all on strips %.1f us, split %.1f us)\n",
                                  kmove, nt0, std::max(*std::max_element(I.begin(), I.end()) * kLoneNs, I_total / kSimds * kSimdNs) * 1e-3, std::min(tA, tB));
            if (kmove) {
                std::vector<uint8_t> moved(nt0, 0);
                for (size_t k = 0; k < kmove; ++k) moved[ord[k]] = 1;
                std::vector<HostTask> keep;
                for (size_t t = 0; t < nt0; ++t) {
                    if (!moved[t]) {
                        keep.push_back(ht[t]);
                        continue;
                    }
                    for (uint32_t l = 0; l < ht[t].count; ++l) pair_list.push_back(order[ht[t].first + l]);
                }
                ht.swap(keep);
                std::sort(pair_list.begin(), pair_list.end());
                if (!ht.empty()) choose_strip_height();   // the strips that stay may prefer another height
            }
        }
        if (ht.empty()) b->use_strips = false;
        live_pairs_engine.swap(pair_list);
      if (b->use_strips) {
        kmode = best_mode;
        const int R = bestR;
        for (const auto& t : ht) b->padded_cells += (t.maxlen + bestR - 1) / bestR * bestR * (b->lanes ? (t.m + 3) / 4 * 4 : t.m) * 64;
        b->kern = find_batch_kernel(R, kmode, score_path);
        if (!b->kern) return fail(ctx, PWA_E_INVALID, "internal: no kernel instantiation");
        b->kernel_name = b->kern->name;
        std::sort(ht.begin(), ht.end(), [&](const HostTask& x, const HostTask& y) {   // longest first
            const uint64_t cx = (x.maxlen + R - 1) / R * x.m, cy = (y.maxlen + R - 1) / R * y.m;
            if (cx != cy) return cx > cy;
            return x.first < y.first;
        });
        const size_t nt = ht.size();
        // task list and lane slots are built in page-locked buffers of the context and uploaded from there (see PinnedBuf)
        HIPC(ctx, ctx->pin[pwa_ctx::PIN_TASKS].reserve(nt * sizeof(BatchTask)));
        for (int q = 0; q < (b->lanes ? 5 : 3); ++q) HIPC(ctx, ctx->pin[pwa_ctx::PIN_SLOT0 + q].reserve(nt * 64 * sizeof(uint32_t)));
        BatchTask* const tasks = ctx->pin[pwa_ctx::PIN_TASKS].as<BatchTask>();
        uint32_t* const spoff = ctx->pin[pwa_ctx::PIN_SLOT0].as<uint32_t>();
        uint32_t* const splen = ctx->pin[pwa_ctx::PIN_SLOT1].as<uint32_t>();
        uint32_t* const sout = ctx->pin[pwa_ctx::PIN_SLOT2].as<uint32_t>();
        uint32_t* const stoff = b->lanes ? ctx->pin[pwa_ctx::PIN_SLOT3].as<uint32_t>() : nullptr;   // empty lanes: no text, no pattern
        uint32_t* const stlen = b->lanes ? ctx->pin[pwa_ctx::PIN_SLOT4].as<uint32_t>() : nullptr;
        std::memset(tasks, 0, nt * sizeof(BatchTask));
        std::memset(spoff, 0, nt * 64 * sizeof(uint32_t));
        std::memset(splen, 0, nt * 64 * sizeof(uint32_t));
        std::memset(sout, 0xff, nt * 64 * sizeof(uint32_t));
        if (b->lanes) {
            std::memset(stoff, 0, nt * 64 * sizeof(uint32_t));
            std::memset(stlen, 0, nt * 64 * sizeof(uint32_t));
        }
        uint32_t max_strips = 1;
        size_t two_strip_tasks = 0;
        {   // (a million slots through three indirections each: on a few threads for long task lists)
            const int T = (int)std::max<size_t>(1, std::min<size_t>({8, std::max(1u, std::thread::hardware_concurrency()), nt >> 11}));
            std::vector<uint32_t> part_max((size_t)T, 1);
            std::vector<size_t> part_two((size_t)T, 0);
            auto work = [&](int th) {
                const size_t a = nt * (size_t)th / (size_t)T, z = nt * (size_t)(th + 1) / (size_t)T;
                for (size_t t = a; t < z; ++t) {
                    tasks[t].text_off = (uint32_t)aoff[ht[t].text];
                    tasks[t].text_len = (uint32_t)ht[t].m;
                    tasks[t].slot0 = (uint32_t)(t * 64);
                    tasks[t].n_strips = (uint32_t)((ht[t].maxlen + R - 1) / R);
                    part_max[(size_t)th] = std::max(part_max[(size_t)th], tasks[t].n_strips);
                    part_two[(size_t)th] += tasks[t].n_strips == 2;
                    for (uint32_t l = 0; l < ht[t].count; ++l) {
                        const uint32_t k = order[ht[t].first + l];
                        spoff[t * 64 + l] = (uint32_t)aoff[pair_a[k]];
                        splen[t * 64 + l] = (uint32_t)slen(pair_a[k]);
                        sout[t * 64 + l] = k;
                        if (b->lanes) {
                            stoff[t * 64 + l] = (uint32_t)aoff[pair_b[k]];
                            stlen[t * 64 + l] = (uint32_t)slen(pair_b[k]);
                        }
                    }
                }
            };
            std::vector<std::thread> pool;
            for (int th = 1; th < T; ++th) pool.emplace_back(work, th);
            work(0);
            for (auto& x : pool) x.join();
            for (int th = 0; th < T; ++th) {
                max_strips = std::max(max_strips, part_max[(size_t)th]);
                two_strip_tasks += part_two[(size_t)th];
            }
        }
        if (b->lanes && kmode == BM_NWG) {
            // right-aligned, front-padded (code 4) text rows: task t, lane l at t_base + l * M_t, M_t = 4 * ceil(max m / 4)
            uint64_t total = 0;
            std::vector<uint64_t> tbase(nt);
            for (size_t t = 0; t < nt; ++t) {
                tbase[t] = total;
                total += 64ull * ((ht[t].m + 3) / 4 * 4);
            }
            if (total + 64 >= (ctx->knobs.lane_rows_limit ? ctx->knobs.lane_rows_limit : 0xffffffffull))
                return fail(ctx, PWA_E_CAPACITY, "per-lane text rows exceed 4 GiB");   // (one-shot calls halve the run and retry)
            // The rows are built straight in the context's two page-locked arena buffers, in pieces of whole tasks (~32 MiB), by several host
            // threads, while the previous piece is on its way (copy stream) -- like build_arena.  (Until r03: one thread, byte by byte into a
            // heap buffer, then through the bounce buffer: 156 ms of a 159 ms call for 131 072 pairs 150 x 2000.)
            HIPC(ctx, b->lane_text.alloc(total + 64));
            uint8_t code8[256];
            for (int v = 0; v < 256; ++v) code8[v] = (uint8_t)code_of[v];
            constexpr uint64_t kPiece = 32ull << 20;
            auto task_end = [&](size_t t) { return t + 1 < nt ? tbase[t + 1] : total + 64; };   // (the slack after the last task is pad as well)
            int piece = 0;
            for (size_t t0 = 0; t0 < nt; ++piece) {
                size_t t1 = t0 + 1;
                while (t1 < nt && task_end(t1) - tbase[t0] <= kPiece) ++t1;
                const uint64_t base = tbase[t0], bytes = task_end(t1 - 1) - base;
                PinnedBuf& pb = ctx->pin[pwa_ctx::PIN_ARENA + (piece & 1)];
                if (piece >= 2) HIPC(ctx, hipEventSynchronize(ctx->copy_ev[piece & 1]));   // the copy that last read this buffer
                HIPC(ctx, pb.reserve(bytes));
                uint8_t* const host = pb.as<uint8_t>();
                const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>({16, bytes / (1ull << 20) + 1, std::max(1u, std::thread::hardware_concurrency()), (uint64_t)(t1 - t0)}));
                auto work = [&](int th) {
                    const size_t a = t0 + (t1 - t0) * (size_t)th / (size_t)T, z = t0 + (t1 - t0) * (size_t)(th + 1) / (size_t)T;
                    for (size_t t = a; t < z; ++t) {
                        const uint64_t M = (ht[t].m + 3) / 4 * 4;
                        std::memset(host + (tbase[t] - base), 4, task_end(t) - tbase[t]);
                        tasks[t].text_len = (uint32_t)M;
                        for (uint32_t l = 0; l < ht[t].count; ++l) {
                            const uint32_t k = order[ht[t].first + l];
                            const uint8_t* src = seq_bytes + seq_off[pair_b[k]];
                            const uint64_t len = slen(pair_b[k]);
                            uint8_t* dst = host + (tbase[t] - base) + (uint64_t)l * M + (M - len);
                            for (uint64_t o = 0; o < len; ++o) dst[o] = code8[src[o]];
                            stoff[t * 64 + l] = (uint32_t)(tbase[t] + (uint64_t)l * M);
                        }
                    }
                };
                std::vector<std::thread> pool;
                for (int th = 1; th < T; ++th) pool.emplace_back(work, th);
                work(0);
                for (auto& x : pool) x.join();
                HIPC(ctx, hipMemcpyAsync(b->lane_text.as<uint8_t>() + base, host, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
                HIPC(ctx, hipEventRecord(ctx->copy_ev[piece & 1], ctx->copy_stream));
                t0 = t1;
            }
            HIPC(ctx, hipStreamSynchronize(ctx->copy_stream));
        }
        if (b->lanes) {
            HIPC(ctx, b->slot_toff.alloc(nt * 64 * 4));
            HIPC(ctx, hipMemcpy(b->slot_toff.p, stoff, nt * 64 * 4, hipMemcpyHostToDevice));
            HIPC(ctx, b->slot_tlen.alloc(nt * 64 * 4));
            HIPC(ctx, hipMemcpy(b->slot_tlen.p, stlen, nt * 64 * 4, hipMemcpyHostToDevice));
        }
        mark("choose R + slot arrays");
        HIPC(ctx, b->tasks.alloc(nt * sizeof(BatchTask)));
        HIPC(ctx, hipMemcpy(b->tasks.p, tasks, nt * sizeof(BatchTask), hipMemcpyHostToDevice));
        HIPC(ctx, b->slot_poff.alloc(nt * 64 * 4));
        HIPC(ctx, hipMemcpy(b->slot_poff.p, spoff, nt * 64 * 4, hipMemcpyHostToDevice));
        HIPC(ctx, b->slot_plen.alloc(nt * 64 * 4));
        HIPC(ctx, hipMemcpy(b->slot_plen.p, splen, nt * 64 * 4, hipMemcpyHostToDevice));
        HIPC(ctx, b->slot_out.alloc(nt * 64 * 4));
        HIPC(ctx, hipMemcpy(b->slot_out.p, sout, nt * 64 * 4, hipMemcpyHostToDevice));

        b->single_strip = !affine && !nwdist && max_strips == 1 && b->kern->fn_single != nullptr;
        // Opt-in (PWA_PAIRED=1): two-strip tasks (the C3 shape: 150-row patterns in 76-row strips) as two waves of one
        // workgroup that pass the boundary row through an LDS ring instead of HBM.  It removes the hand-off traffic
        // (84 GB per C3 launch -> none) but couples the two waves' progress: [gpu] 172.5 ms against 161.9 ms for the
        // HBM hand-off form, which already runs at the VALU issue limit -- hence not the default.
        b->paired = false;
        if (ctx->knobs.paired >= 0)
            b->paired = ctx->knobs.paired != 0 && !affine && !nwdist && max_strips == 2 && b->kern->fn_pair != nullptr &&
                        two_strip_tasks * 8 >= nt * 7;   // at most 1 task in 8 may leave the second wave idle
        if (b->lanes) {
            b->paired = false;
            b->kernel_name = std::string(b->kernel_name).insert(b->kernel_name.size() - 1, ",LANES");
        }
        const void* kfn = b->lanes        ? reinterpret_cast<const void*>(max_strips == 1 ? b->kern->fn_lanes_single : b->kern->fn_lanes)
                          : b->single_strip ? reinterpret_cast<const void*>(b->kern->fn_single)
                          : b->paired     ? reinterpret_cast<const void*>(b->kern->fn_pair)
                          : nwdist        ? reinterpret_cast<const void*>(b->kern->dfn)
                                 : (affine ? reinterpret_cast<const void*>(b->kern->afn) : reinterpret_cast<const void*>(b->kern->fn));
        if (b->paired) b->kernel_name = std::string("batch_scores_pair_kernel") + (b->kernel_name.c_str() + std::strlen("batch_scores_kernel"));
        int per_cu = 0;
        HIPC(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, b->paired ? 128 : 64, 0));
        per_cu = std::max(1, std::min(per_cu, 32));
        b->grid = (uint32_t)std::min<uint64_t>(nt, (uint64_t)ctx->num_cu * per_cu);
        // int32 per half: one (affine: two) int4 per lane per 4-column block
        const int hand_vals = (affine || (nwdist && kmode != BM_DISTP)) ? 2 : 1;   // int4 per lane per 4-column block
        const uint64_t half = ((max_strips > 1 && !b->paired) ? ((max_m + 3) / 4 + 1) * 256 : 256) * hand_vals;
        // Strip s reads the half written by strip s-1 and writes the other one.  With at most two strips per task
        // the second half is only ever the parked dummy block (stride 0), so it is one block long: for C3 that
        // turns a 10.5 GB workspace (0.24 s of hipMalloc, profiles/r01_malloc_probe.txt) into 5.2 GB (0.3 ms).
        const uint64_t block_ints = 256 * hand_vals;
        const uint64_t second = (max_strips > 2 || b->paired) ? half : block_ints;
        {   // very long texts: fewer workgroups rather than a workspace that does not fit (tasks come off a queue,
            // any grid is correct)
            size_t free_b = 0, total_b = 0;
            HIPC(ctx, hipMemGetInfo(&free_b, &total_b));
            const uint64_t per_wg = (half + second) * sizeof(int32_t);
            const uint64_t fit = std::max<uint64_t>(1, (uint64_t)(free_b * 0.6) / per_wg);
            b->grid = (uint32_t)std::min<uint64_t>(b->grid, fit);
            b->grid = (b->grid + 3u) & ~3u;   // whole four-wave workgroups (pwa_batch_run): every wave has a hand-off region of its own
        }
        {
            const size_t hand_bytes = (size_t)b->grid * (half + second) * sizeof(int32_t);
            if (ctx->hand_cache && ctx->hand_cache_bytes >= hand_bytes) {   // left behind by an earlier batch of this context
                b->hand.p = ctx->hand_cache;
                b->hand.bytes = ctx->hand_cache_bytes;
                ctx->hand_cache = nullptr;
                ctx->hand_cache_bytes = 0;
            } else {
                HIPC(ctx, b->hand.alloc(hand_bytes));
            }
        }

        mark("uploads + workspace");
        BatchParams& P = b->bp;
        P.arena = b->arena.as<uint8_t>();
        P.tasks = b->tasks.as<BatchTask>();
        P.slot_poff = b->slot_poff.as<uint32_t>();
        P.slot_plen = b->slot_plen.as<uint32_t>();
        P.slot_out = b->slot_out.as<uint32_t>();
        P.scores = b->scores.as<int32_t>();
        P.hand = b->hand.as<int32_t>();
        P.hand_stride = half + second;
        P.hand_half = (uint32_t)half;
        P.queue = b->queue.as<uint32_t>();
        P.n_tasks = (uint32_t)nt;
        P.match = tab_match;
        P.mismatch = tab_mismatch;
        P.gap = gap;
        const uint32_t bm = (uint8_t)(int8_t)tab_match, bx = (uint8_t)(int8_t)tab_mismatch;
        P.tab_lo = bm | (bx << 8) | (bx << 16) | (bx << 24);   // selector 0 -> match
        P.tab_hi = bx * 0x01010101u;                           // selectors 4..7 -> mismatch
        const uint32_t pad = (score_path == SC_PERM) ? 7u : (uint32_t)absent_byte;
        P.pad_word = pad * 0x01010101u;
        P.tpad_word = ((score_path == SC_PERM) ? 6u : (uint32_t)std::max(text_pad_byte, 0)) * 0x01010101u;
        P.lane_text = b->lane_text.as<uint8_t>();
        if (b->lanes && kmode == BM_NWG) P.tab_hi = (uint32_t)(uint8_t)(int8_t)(-gap) * 0x01010101u;   // selectors 4..7: front pad = a gap column
        P.slot_toff = b->slot_toff.as<uint32_t>();
        P.slot_tlen = b->slot_tlen.as<uint32_t>();
      }   // strips that stay
    }
    if (!live_pairs_engine.empty()) {
        // ---- the pairs that do not run on strips: short patterns over a coded arena on the mini-stripe engine (no band, four pairs per
        // wave, one launch per row class), everything else on the stripe engine (no band: exact first-maximum end cells, any scoring)
        std::vector<uint32_t> plist;
        std::vector<std::pair<int, std::vector<uint32_t>>> mini_lists;   // (rows per lane, pairs)
        for (const uint32_t k : live_pairs_engine) {
            const int mrl = mini_rl_of(slen(pair_a[k]));
            if (!mrl) {
                plist.push_back(k);
                continue;
            }
            size_t c = 0;
            while (c < mini_lists.size() && mini_lists[c].first != mrl) ++c;
            if (c == mini_lists.size()) mini_lists.emplace_back(mrl, std::vector<uint32_t>());
            mini_lists[c].second.push_back(k);
        }
        const size_t nl = live_pairs_engine.size();
        HIPC(ctx, b->pair_res.alloc(nl * sizeof(PairResult)));
        HIPC(ctx, hipMemset(b->pair_res.p, 0, nl * sizeof(PairResult)));
        size_t q_next = 0;
        auto describe = [&](uint32_t k, size_t q) {
            PairDesc d;
            std::memset(&d, 0, sizeof d);
            d.pat = b->arena.as<uint8_t>() + aoff[pair_a[k]];
            d.txt = b->arena.as<uint8_t>() + aoff[pair_b[k]];
            d.n = (int32_t)slen(pair_a[k]);
            d.m = (int32_t)slen(pair_b[k]);
            d.res = b->pair_res.as<PairResult>() + q;
            d.out_index = k;
            return d;
        };
        std::string names;
        if (!plist.empty()) {
            b->use_pairs = true;
            b->live_idx = plist;
            std::vector<PairDesc> pd;
            pd.reserve(plist.size());
            uint64_t pe_max_n = 0;
            for (const uint32_t k : plist) pe_max_n = std::max(pe_max_n, slen(pair_a[k]));
            PairGeom geom = choose_geom(ctx->knobs, pe_max_n);
            {   // RL = 2 buys a pair more waves in flight -- which a list that fills the chip anyway does not need: [gpu, r03] SW scores of
                // 10k x 10k pairs, RL = 2 / RL = 4: 8 pairs 1.73 / 1.79 ms, 64 pairs 5.35 / 4.93 ms, 256 pairs 17.3 / 12.8 ms
                uint64_t stripes2 = 0;
                for (const uint32_t k : plist) stripes2 += (slen(pair_a[k]) + 127) / 128;
                if (geom.rl == 2 && geom.w == 4 && !ctx->knobs.force_rl && stripes2 >= 2048) geom.rl = 4;
            }
            // a coded arena with keys in range (what the band-less mini kernels ask for as well): the keyed chunk without a band -- table
            // scoring, one v_max3 per cell, global fills gap-shifted -- instead of the plain compare-and-select step: [gpu, r03] SW scores of
            // 64 pairs 10k x 10k 4.8 -> 3.45 ms, NW 4.06 -> 2.05 ms; one pair 1.67 -> 1.26 / 1.55 -> 1.01 ms
            const bool keyed_scores = mini_scores && !ctx->knobs.no_keyed_tb && !ctx->knobs.no_pair_table;
            const bool gap0_scores = keyed_scores && mini_gap0 && !ctx->knobs.no_gap_shift;
            for (const uint32_t k : plist) {
                PairDesc d = describe(k, q_next++);
                d.score_bias = gap0_scores ? wrap_mul((int64_t)(slen(pair_a[k]) + slen(pair_b[k])), gap) : 0;
                pd.push_back(d);
                b->padded_cells += (slen(pair_a[k]) + 64 * geom.rl - 1) / (64 * geom.rl) * (64 * geom.rl) * slen(pair_b[k]);
            }
            b->pl.perm = keyed_scores;
            b->pl.keyed = true;
            b->pl.gap0 = gap0_scores;
            const int rc = b->pl.build(ctx, pd, gap0_scores ? match - 2 * gap : match, gap0_scores ? mismatch - 2 * gap : mismatch, gap0_scores ? 0 : gap, geom);
            if (rc != PWA_OK) return rc;
            b->pl.G.scores_out = b->scores.as<int32_t>();   // the device score vector is complete after run()
            names = std::string("pair_fill_kernel<RL=") + std::to_string(geom.rl) + (local ? ",SW" : (gap0_scores ? ",NW,GAP0" : ",NW")) + (keyed_scores ? ",keyed,no-band>" : ",no-traceback>");
        }
        for (auto& cls : mini_lists) {
            const int rl = cls.first;
            std::vector<uint32_t>& lst = cls.second;
            sort_by_length_desc(lst, [&](uint32_t x) { return slen(pair_b[x]); });   // a wave's four texts about equally long
            std::vector<PairDesc> pd;
            pd.reserve(lst.size() + 3);
            for (const uint32_t k : lst) {
                PairDesc d = describe(k, q_next++);
                b->live_idx.push_back(k);
                d.score_bias = mini_gap0 ? wrap_mul((int64_t)(slen(pair_a[k]) + slen(pair_b[k])), gap) : 0;
                pd.push_back(d);
                b->padded_cells += (uint64_t)(16 * rl) * slen(pair_b[k]);
            }
            const uint32_t n_real = (uint32_t)pd.size();
            while (pd.size() % 4) {   // empty patterns fill the last wave (their results go nowhere: no row of theirs is row n)
                PairDesc d = pd[n_real - 1];
                d.n = 0;
                pd.push_back(d);
            }
            b->mini.emplace_back(new PairLaunch());
            PairLaunch& ml = *b->mini.back();
            for (DevBuf* d : {&ml.desc, &ml.tasks, &ml.rows, &ml.progress, &ml.best, &ml.queue}) d->pool = ctx;
            ml.perm = ml.keyed = true;
            ml.gap0 = mini_gap0;
            const int rc = ml.build_mini(ctx, pd, n_real, mini_gap0 ? match - 2 * gap : match, mini_gap0 ? mismatch - 2 * gap : mismatch, mini_gap0 ? 0 : gap, rl);
            if (rc != PWA_OK) return rc;
            ml.G.scores_out = b->scores.as<int32_t>();
            names += std::string(names.empty() ? "" : " + ") + "mini_fill_kernel<RL=" + std::to_string(rl) + (local ? ",SW" : (mini_gap0 ? ",NW,GAP0" : ",NW")) + ",no-band>";
        }
        b->kernel_name = b->use_strips ? b->kernel_name + " + " + names : names;   // (the strip kernel first: bench.py prices its instruction mix)
    }
    if (dbg) {
        mark("engine setup");
        HIPC(ctx, hipDeviceSynchronize());
        mark("hipDeviceSynchronize");
        hipLaunchKernelGGL(pwa_nop_kernel, dim3(1), dim3(64), 0, ctx->stream, (int*)nullptr);
        HIPC(ctx, hipStreamSynchronize(ctx->stream));
        mark("nop kernel + sync");
        hipLaunchKernelGGL(pwa_nop_kernel, dim3(1), dim3(64), 0, ctx->stream, (int*)nullptr);
        HIPC(ctx, hipStreamSynchronize(ctx->stream));
        mark("nop kernel + sync again");
    }
    guard.b = nullptr;
    *out = b;
    return PWA_OK;
} catch (const std::bad_alloc&) {
    return fail(ctx, PWA_E_NOMEM, "host allocation failed");
} catch (...) {
    return fail(ctx, PWA_E_HIP, "unexpected C++ exception");   // nothing may propagate across the C ABI
}

extern "C" {

int pwa_batch_create(pwa_ctx* ctx, int mode, int match, int mismatch, int gap, const uint8_t* seq_bytes,
                     const uint64_t* seq_off, uint32_t n_seq, const uint32_t* pair_a, const uint32_t* pair_b,
                     uint64_t n_pairs, int want_end_cells, pwa_batch** out) {
    return batch_create_impl(ctx, mode, match, mismatch, gap, KIND_LINEAR, 0, seq_bytes, seq_off, n_seq, pair_a, pair_b, n_pairs,
                             want_end_cells, out);
}

int pwa_affine_batch_create(pwa_ctx* ctx, int match, int mismatch, int gap_open, int gap_extend, const uint8_t* seq_bytes,
                            const uint64_t* seq_off, uint32_t n_seq, const uint32_t* pair_a, const uint32_t* pair_b,
                            uint64_t n_pairs, pwa_batch** out) {
    return batch_create_impl(ctx, PWA_MODE_NW, match, mismatch, gap_open, KIND_AFFINE, gap_extend, seq_bytes, seq_off, n_seq, pair_a,
                             pair_b, n_pairs, 0, out);
}

int pwa_nwdist_batch_create(pwa_ctx* ctx, int match, int mismatch, int gap, const uint8_t* seq_bytes, const uint64_t* seq_off,
                            uint32_t n_seq, const uint32_t* pair_a, const uint32_t* pair_b, uint64_t n_pairs, pwa_batch** out) {
    return batch_create_impl(ctx, PWA_MODE_NW, match, mismatch, gap, KIND_NWDIST, 0, seq_bytes, seq_off, n_seq, pair_a, pair_b,
                             n_pairs, 0, out);
}

// hw3.cpp:261-283: full affine-gap alignments (score + op list) of a pair list.  Pairs are grouped by string1 (for
// the center-star step every pair has the center there): it becomes the wave's shared text and every lane runs its
// own string2 down the rows (batch_affine_tb.hip.h).  Raw bytes, compare path, 32-row strips: the pass covers N-1
// pairs next to the all-pairs score pass over N(N-1)/2, so it is built for exactness, not for speed.
int pwa_align_affine_batch(pwa_ctx* ctx, int match, int mismatch, int gap_open, int gap_extend, const uint8_t* seq_bytes,
                           const uint64_t* seq_off, uint32_t n_seq, const uint32_t* pair_a, const uint32_t* pair_b,
                           uint64_t n_pairs, int32_t* score_out, uint8_t* ops, const uint64_t* ops_off, uint64_t* n_ops) try {
    if (!ctx) return PWA_E_INVALID;
    if (!seq_off || !score_out || !ops || !ops_off || !n_ops || (n_pairs && (!pair_a || !pair_b)))
        return fail(ctx, PWA_E_INVALID, "null input");
    if (n_seq && !seq_bytes && seq_off[n_seq] != 0) return fail(ctx, PWA_E_INVALID, "null seq_bytes");
    if (n_pairs >= 0xffffffffull) return fail(ctx, PWA_E_CAPACITY, "more than 2^32-2 pairs in one batch");
    for (uint64_t k = 0; k < n_pairs; ++k)
        if (pair_a[k] >= n_seq || pair_b[k] >= n_seq) return fail(ctx, PWA_E_INVALID, "pair index out of range");
    HIPC(ctx, hipSetDevice(ctx->device));
    constexpr int R = 32, Q = R / 4;
    auto slen = [&](uint32_t s) -> uint64_t { return seq_off[s + 1] - seq_off[s]; };

    // ---- pairs with an empty side: the reference's boundary walk (hw3.cpp:42-53, 105-131) -- all 'D' or all 'I'
    std::vector<uint32_t> live;
    uint64_t max_m = 0;
    for (uint64_t k = 0; k < n_pairs; ++k) {
        const uint64_t n1 = slen(pair_a[k]), n2 = slen(pair_b[k]);
        if (n1 > 0x3fffffffull || n2 > 0x3fffffffull) return fail(ctx, PWA_E_CAPACITY, "sequence longer than 2^30");
        if (n1 == 0 || n2 == 0) {
            score_out[k] = (n1 + n2 == 0) ? 0 : (int32_t)((uint32_t)gap_open + (uint32_t)wrap_mul((int64_t)(n1 + n2 - 1), gap_extend));
            std::memset(ops + ops_off[k], n1 ? 'D' : 'I', n1 + n2);
            n_ops[k] = n1 + n2;
            continue;
        }
        live.push_back((uint32_t)k);
        max_m = std::max(max_m, n1);
    }
    if (live.empty()) return PWA_OK;

    // ---- arena: raw bytes; pad byte = one that no string1 (text) contains
    std::vector<uint8_t> is_used(n_seq, 0);
    bool in_text[256] = {false};
    for (uint32_t k : live) {
        is_used[pair_a[k]] = is_used[pair_b[k]] = 1;
    }
    {
        std::vector<uint8_t> is_text(n_seq, 0);
        for (uint32_t k : live) is_text[pair_a[k]] = 1;
        for (uint32_t s = 0; s < n_seq; ++s)
            if (is_text[s])
                for (uint64_t o = seq_off[s]; o < seq_off[s + 1]; ++o) in_text[seq_bytes[o]] = true;
    }
    int pad_byte = -1;
    for (int v = 255; v >= 0 && pad_byte < 0; --v)
        if (!in_text[v]) pad_byte = v;
    if (pad_byte < 0) return fail(ctx, PWA_E_CAPACITY, "the first sequences of the pairs use all 256 byte values: no padding symbol left");
    std::vector<uint64_t> aoff(n_seq, 0);
    uint64_t arena_bytes = 0;
    for (uint32_t s = 0; s < n_seq; ++s)
        if (is_used[s]) {
            aoff[s] = arena_bytes;
            arena_bytes += align_up(slen(s) + 1, 16);
        }
    arena_bytes += 512;
    if (arena_bytes >= 0xffffffffull) return fail(ctx, PWA_E_CAPACITY, "sequence arena exceeds 4 GiB");
    DevBuf arena;
    {
        std::vector<uint8_t> host_arena(arena_bytes, 0);
        for (uint32_t s = 0; s < n_seq; ++s)
            if (is_used[s] && slen(s)) std::memcpy(host_arena.data() + aoff[s], seq_bytes + seq_off[s], slen(s));
        HIPC(ctx, arena.alloc(arena_bytes));
        HIPC(ctx, upload_via_bounce(ctx, arena.p, host_arena.data(), arena_bytes));
    }

    // ---- wave tasks: pairs grouped by string1, string2 sorted by length (descending), 64 per wave
    std::vector<uint32_t> order(live);
    {
        std::vector<uint64_t> key(order.size());
        for (size_t o = 0; o < order.size(); ++o)
            key[o] = ((uint64_t)pair_a[order[o]] << 32) | (uint64_t)(0x7fffffffu - (uint32_t)slen(pair_b[order[o]]));
        radix_sort_by_key(key, order);
    }
    struct HostTask {
        uint32_t first, count;
        uint64_t strips, m, tb_dwords;
    };
    std::vector<HostTask> ht;
    for (size_t p = 0; p < order.size();) {
        size_t q = p;
        while (q < order.size() && q - p < 64 && pair_a[order[q]] == pair_a[order[p]]) ++q;
        const uint64_t strips = (slen(pair_b[order[p]]) + R - 1) / R, m = slen(pair_a[order[p]]);
        ht.push_back({(uint32_t)p, (uint32_t)(q - p), strips, m, strips * m * Q * 64});
        p = q;
    }

    size_t free_b = 0, total_b = 0;
    HIPC(ctx, hipMemGetInfo(&free_b, &total_b));
    const uint64_t tb_budget_dw = std::max<uint64_t>(std::min<uint64_t>((uint64_t)(free_b * 0.6), 6ull << 30) / 4, 1);
    DevBuf d_scores, d_ops, d_nops, d_queue, d_hand;
    HIPC(ctx, d_scores.alloc(n_pairs * sizeof(int32_t)));
    HIPC(ctx, d_nops.alloc(n_pairs * sizeof(uint32_t)));
    HIPC(ctx, d_queue.alloc(64));
    uint64_t ops_total = 0;
    std::vector<uint64_t> dev_ops_off(n_pairs, 0);
    for (uint32_t k : live) {
        dev_ops_off[k] = ops_total;
        ops_total += align_up(slen(pair_a[k]) + slen(pair_b[k]) + 1, 16);
    }
    HIPC(ctx, d_ops.alloc(ops_total));
    const uint64_t half = ((max_m + 3) / 4 + 1) * 192 * 4;   // int32 per half: three int4 per lane per 4-column block
    const uint32_t grid_cap = (uint32_t)std::min<uint64_t>(ht.size(), (uint64_t)ctx->num_cu * 2);
    HIPC(ctx, d_hand.alloc((size_t)grid_cap * 2 * half * sizeof(int32_t)));

    // ---- chunks of tasks whose code bands fit the budget
    for (size_t t0 = 0; t0 < ht.size();) {
        size_t t1 = t0;
        uint64_t dw = 0;
        while (t1 < ht.size() && (t1 == t0 || dw + ht[t1].tb_dwords <= tb_budget_dw)) dw += ht[t1++].tb_dwords;
        if (dw * 4 > (uint64_t)(free_b * 0.9)) return fail(ctx, PWA_E_NOMEM, "traceback codes of one wave task exceed free HBM");
        const size_t nt = t1 - t0;
        std::vector<BatchTask> tasks(nt);
        std::vector<uint32_t> spoff(nt * 64, 0), splen(nt * 64, 0), sout(nt * 64, 0xffffffffu);
        std::vector<uint64_t> tboff(nt);
        std::vector<AffineWalkPair> wp;
        uint64_t at = 0;
        for (size_t t = 0; t < nt; ++t) {
            const HostTask& h = ht[t0 + t];
            const uint32_t text = pair_a[order[h.first]];
            tasks[t].text_off = (uint32_t)aoff[text];
            tasks[t].text_len = (uint32_t)h.m;
            tasks[t].slot0 = (uint32_t)(t * 64);
            tasks[t].n_strips = (uint32_t)h.strips;
            tboff[t] = at;
            for (uint32_t l = 0; l < h.count; ++l) {
                const uint32_t k = order[h.first + l];
                spoff[t * 64 + l] = (uint32_t)aoff[pair_b[k]];
                splen[t * 64 + l] = (uint32_t)slen(pair_b[k]);
                sout[t * 64 + l] = k;
                wp.push_back({at, dev_ops_off[k], l, (uint32_t)h.m, (uint32_t)slen(pair_b[k]), k});
            }
            at += h.tb_dwords;
        }
        DevBuf d_tb, d_tasks, d_spoff, d_splen, d_sout, d_tboff, d_wp;
        HIPC(ctx, d_tb.alloc(dw * 4));
        HIPC(ctx, d_tasks.alloc(nt * sizeof(BatchTask)));
        HIPC(ctx, upload_via_bounce(ctx, d_tasks.p, tasks.data(), nt * sizeof(BatchTask)));
        HIPC(ctx, d_spoff.alloc(nt * 64 * 4));
        HIPC(ctx, upload_via_bounce(ctx, d_spoff.p, spoff.data(), nt * 64 * 4));
        HIPC(ctx, d_splen.alloc(nt * 64 * 4));
        HIPC(ctx, upload_via_bounce(ctx, d_splen.p, splen.data(), nt * 64 * 4));
        HIPC(ctx, d_sout.alloc(nt * 64 * 4));
        HIPC(ctx, upload_via_bounce(ctx, d_sout.p, sout.data(), nt * 64 * 4));
        HIPC(ctx, d_tboff.alloc(nt * sizeof(uint64_t)));
        HIPC(ctx, upload_via_bounce(ctx, d_tboff.p, tboff.data(), nt * sizeof(uint64_t)));
        HIPC(ctx, d_wp.alloc(wp.size() * sizeof(AffineWalkPair)));
        HIPC(ctx, upload_via_bounce(ctx, d_wp.p, wp.data(), wp.size() * sizeof(AffineWalkPair)));

        AffineTbParams T;
        std::memset(&T, 0, sizeof T);
        BatchParams& P = T.a.b;
        P.arena = arena.as<uint8_t>();
        P.tasks = d_tasks.as<BatchTask>();
        P.slot_poff = d_spoff.as<uint32_t>();
        P.slot_plen = d_splen.as<uint32_t>();
        P.slot_out = d_sout.as<uint32_t>();
        P.scores = d_scores.as<int32_t>();
        P.hand = d_hand.as<int32_t>();
        P.hand_stride = 2 * half;
        P.hand_half = (uint32_t)half;
        P.queue = d_queue.as<uint32_t>();
        P.n_tasks = (uint32_t)nt;
        P.match = match;
        P.mismatch = mismatch;
        P.gap = gap_open;
        P.pad_word = (uint32_t)pad_byte * 0x01010101u;
        T.a.go = gap_open;
        T.a.ge = gap_extend;
        T.a.neg = std::numeric_limits<int32_t>::min() / 2;   // hw3.cpp:16
        T.tb = d_tb.as<uint32_t>();
        T.task_tb_off = d_tboff.as<uint64_t>();
        const uint32_t grid = (uint32_t)std::min<uint64_t>(nt, grid_cap);
        HIPC(ctx, hipMemsetAsync(d_queue.p, 0, 16, ctx->stream));
        hipLaunchKernelGGL((batch_affine_tb_kernel<R, SC_CMP>), dim3(grid), dim3(64), 0, ctx->stream, T);
        HIPC(ctx, hipGetLastError());
        hipLaunchKernelGGL((affine_walk_kernel<R>), dim3((uint32_t)((wp.size() + 63) / 64)), dim3(64), 0, ctx->stream,
                           d_wp.as<AffineWalkPair>(), (uint32_t)wp.size(), d_tb.as<uint32_t>(), d_ops.as<uint8_t>(),
                           d_nops.as<uint32_t>());
        HIPC(ctx, hipGetLastError());
        HIPC(ctx, hipStreamSynchronize(ctx->stream));
        t0 = t1;
    }
    std::vector<int32_t> h_scores(n_pairs);
    std::vector<uint32_t> h_nops(n_pairs);
    std::vector<uint8_t> h_ops(ops_total);
    HIPC(ctx, hipMemcpy(h_scores.data(), d_scores.p, n_pairs * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIPC(ctx, hipMemcpy(h_nops.data(), d_nops.p, n_pairs * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIPC(ctx, hipMemcpy(h_ops.data(), d_ops.p, ops_total, hipMemcpyDeviceToHost));
    for (uint32_t k : live) {
        score_out[k] = h_scores[k];
        n_ops[k] = h_nops[k];
        if (h_nops[k] > slen(pair_a[k]) + slen(pair_b[k])) return fail(ctx, PWA_E_CAPACITY, "internal: traceback longer than n+m");
        std::memcpy(ops + ops_off[k], h_ops.data() + dev_ops_off[k], h_nops[k]);
    }
    return PWA_OK;
} catch (const std::bad_alloc&) {
    return fail(ctx, PWA_E_NOMEM, "host allocation failed");
} catch (...) {
    return fail(ctx, PWA_E_HIP, "unexpected C++ exception");   // nothing may propagate across the C ABI
}

int pwa_batch_run(pwa_batch* b, void* stream_v) {
    if (!b) return PWA_E_INVALID;
    pwa_ctx* ctx = b->ctx;
    HIPC(ctx, hipSetDevice(ctx->device));   // the caller's thread may have another device current
    hipStream_t st = stream_v ? static_cast<hipStream_t>(stream_v) : ctx->stream;
    const int slot = (int)(b->n_runs % pwa_batch::kRing);
    if (!b->ev0[slot] || !b->ev1[slot]) {   // both or neither: a half-created pair is torn down and created again
        if (b->ev0[slot]) (void)hipEventDestroy(b->ev0[slot]);
        if (b->ev1[slot]) (void)hipEventDestroy(b->ev1[slot]);
        b->ev0[slot] = b->ev1[slot] = nullptr;
        HIPC(ctx, hipEventCreate(&b->ev0[slot]));
        HIPC(ctx, hipEventCreate(&b->ev1[slot]));
    }
    HIPC(ctx, hipEventRecord(b->ev0[slot], st));
    if (b->n_live) {
        if (b->use_strips) {
            HIPC(ctx, hipMemsetAsync(b->queue.p, 0, 16, st));
            if (b->nwdist) {
                NwDistParams dp;
                dp.b = b->bp;
                dp.tab2_lo = 0x000000ffu;   // selector 0 (symbols equal) -> -1, every other selector -> 0
                dp.tab2_hi = 0u;
                dp.scores2 = nullptr;
                hipLaunchKernelGGL(b->kern->dfn, dim3(b->grid), dim3(64), 0, st, dp);
            } else if (b->affine) {
                AffineParams ap;
                ap.b = b->bp;
                ap.go = b->aff_go;
                ap.ge = b->aff_ge;
                ap.neg = b->aff_neg;
                hipLaunchKernelGGL(b->kern->afn, dim3(b->grid), dim3(64), 0, st, ap);
            } else {
                if (b->paired) {
                    hipLaunchKernelGGL(b->kern->fn_pair, dim3(b->grid), dim3(128), 0, st, b->bp);
                } else {
                    // b->grid waves as workgroups of four, with an LDS request that admits exactly their share per CU: a balanced
                    // placement whatever ran before (batch_scores.hip.h); PWA_STRIP_WG1: single-wave workgroups (A/B)
                    const batch_kernel_t fn = b->lanes ? (b->single_strip ? b->kern->fn_lanes_single : b->kern->fn_lanes) : (b->single_strip ? b->kern->fn_single : b->kern->fn);
                    if (ctx->knobs.strip_wg1) {
                        hipLaunchKernelGGL(fn, dim3(b->grid), dim3(64), 0, st, b->bp);
                    } else {
                        const uint32_t n_wg = (b->grid + 3) / 4, per_cu = (n_wg + (uint32_t)ctx->num_cu - 1) / (uint32_t)ctx->num_cu;
                        static const uint32_t kPadKiB[6] = {0, 96, 64, 48, 36, 30};
                        const size_t pad_lds = per_cu <= 5 ? (size_t)kPadKiB[per_cu] * 1024 : 0;
                        HIPC(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad_lds));
                        hipLaunchKernelGGL(fn, dim3(n_wg), dim3(256), pad_lds, st, b->bp);
                    }
                }
            }
            HIPC(ctx, hipGetLastError());
        }
        // the stripe launch of a split batch is often a few long pairs -- tens of waves bound by their own pipeline latency -- so the
        // mini-stripe launches run NEXT to it, on the context's auxiliary stream, forked after the strips and joined before the end
        const bool fork = b->use_pairs && !b->mini.empty();
        hipStream_t ms = fork ? ctx->aux_stream : st;
        if (fork) {
            HIPC(ctx, hipEventRecord(ctx->aux_ev[0], st));
            HIPC(ctx, hipStreamWaitEvent(ms, ctx->aux_ev[0], 0));
        }
        if (b->use_pairs) {
            const int rc = b->pl.launch(ctx, st, b->mode == PWA_MODE_SW, false, false, nullptr);
            if (rc != PWA_OK) return rc;
        }
        for (auto& ml : b->mini) {
            const int rc = ml->launch(ctx, ms, b->mode == PWA_MODE_SW, false, WALK_NONE, nullptr);
            if (rc != PWA_OK) return rc;
        }
        if (fork) {
            HIPC(ctx, hipEventRecord(ctx->aux_ev[1], ms));
            HIPC(ctx, hipStreamWaitEvent(st, ctx->aux_ev[1], 0));
        }
    }
    HIPC(ctx, hipEventRecord(b->ev1[slot], st));
    ++b->n_runs;
    b->ran = true;
    return PWA_OK;
}

int32_t* pwa_batch_d_scores(pwa_batch* b) {
    if (!b) return nullptr;
    return b->ext_scores ? b->ext_scores : b->scores.as<int32_t>();
}

int pwa_batch_set_d_scores(pwa_batch* b, int32_t* d_scores) {
    if (!b || !d_scores) return PWA_E_INVALID;
    pwa_ctx* ctx = b->ctx;
    HIPC(ctx, hipSetDevice(ctx->device));
    // carry over what is already there (pairs with an empty side are resolved at create time)
    HIPC(ctx, hipMemcpy(d_scores, pwa_batch_d_scores(b), b->n_pairs * sizeof(int32_t), hipMemcpyDeviceToDevice));
    b->ext_scores = d_scores;
    b->bp.scores = d_scores;
    b->pl.G.scores_out = d_scores;
    for (auto& ml : b->mini) ml->G.scores_out = d_scores;
    return PWA_OK;
}

int pwa_batch_last_ms(pwa_batch* b, float* ms) {
    if (!b || !ms || !b->ran) return PWA_E_INVALID;
    HIPC(b->ctx, hipSetDevice(b->ctx->device));
    const int slot = (int)((b->n_runs - 1) % pwa_batch::kRing);
    HIPC(b->ctx, hipEventSynchronize(b->ev1[slot]));
    HIPC(b->ctx, hipEventElapsedTime(ms, b->ev0[slot], b->ev1[slot]));
    return PWA_OK;
}

int pwa_batch_run_times(pwa_batch* b, float* ms_out, int cap, int* n_out) {
    if (!b || !ms_out || !n_out || cap < 0) return PWA_E_INVALID;
    HIPC(b->ctx, hipSetDevice(b->ctx->device));
    const uint64_t have = std::min<uint64_t>(b->n_runs, pwa_batch::kRing);
    const int n = (int)std::min<uint64_t>(have, (uint64_t)cap);
    for (int k = 0; k < n; ++k) {   // oldest of the last n first
        const int slot = (int)((b->n_runs - n + k) % pwa_batch::kRing);
        HIPC(b->ctx, hipEventSynchronize(b->ev1[slot]));
        HIPC(b->ctx, hipEventElapsedTime(&ms_out[k], b->ev0[slot], b->ev1[slot]));
    }
    *n_out = n;
    return PWA_OK;
}

int pwa_batch_info(const pwa_batch* b, uint64_t* cells, uint64_t* padded_cells, uint64_t* n_tasks,
                   const char** kernel_name) {
    if (!b) return PWA_E_INVALID;
    if (cells) *cells = b->cells;
    if (padded_cells) *padded_cells = b->padded_cells;
    if (n_tasks) {
        *n_tasks = (b->use_strips ? b->bp.n_tasks : 0) + (b->use_pairs ? (uint64_t)b->pl.G.n_pairs : 0);
        for (const auto& ml : b->mini) *n_tasks += ml->G.n_tasks;
    }
    if (kernel_name) *kernel_name = b->kernel_name.c_str();
    return PWA_OK;
}

int pwa_batch_fetch(pwa_batch* b, int32_t* score_out, uint32_t* end_i_out, uint32_t* end_j_out) try {
    if (!b || !score_out) return PWA_E_INVALID;
    pwa_ctx* ctx = b->ctx;
    if ((end_i_out || end_j_out) && !b->want_end) return fail(ctx, PWA_E_INVALID, "batch was created without end cells");
    if (!b->ran) return fail(ctx, PWA_E_INVALID, "pwa_batch_run has not been called");
    HIPC(ctx, hipSetDevice(ctx->device));
    HIPC(ctx, hipEventSynchronize(b->ev1[(b->n_runs - 1) % pwa_batch::kRing]));
    if (b->use_pairs) {   // the bounded spins of the stripe pipeline: a workgroup that gave up says so here
        const int rc = b->pl.check(ctx);
        if (rc != PWA_OK) return rc;
    }
    if (!b->want_end || b->n_live == 0) {   // both engines write their pairs' scores into the device score vector
        if (b->paired && b->n_live) {   // the LDS hand-off spins are bounded; a wave that gave up says so here
            uint32_t q[2] = {0, 0};
            HIPC(ctx, hipMemcpy(q, b->queue.p, sizeof q, hipMemcpyDeviceToHost));
            if (q[1] != 0) return fail(ctx, PWA_E_HIP, "strip hand-off timed out inside batch_scores_pair_kernel");
        }
        HIPC(ctx, hipMemcpy(score_out, pwa_batch_d_scores(b), b->n_pairs * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (b->want_end) {   // only reachable with no live pairs
            if (end_i_out) std::memcpy(end_i_out, b->host_end_i.data(), b->n_pairs * 4);
            if (end_j_out) std::memcpy(end_j_out, b->host_end_j.data(), b->n_pairs * 4);
        }
        return PWA_OK;
    }
    std::vector<PairResult> res(b->n_live);   // end cells asked for: every live pair ran off the strips (stripe or mini-stripe engine)
    HIPC(ctx, hipMemcpy(res.data(), b->pair_res.p, b->n_live * sizeof(PairResult), hipMemcpyDeviceToHost));
    std::memcpy(score_out, b->host_scores.data(), b->n_pairs * sizeof(int32_t));
    if (b->want_end) {
        if (end_i_out) std::memcpy(end_i_out, b->host_end_i.data(), b->n_pairs * 4);
        if (end_j_out) std::memcpy(end_j_out, b->host_end_j.data(), b->n_pairs * 4);
    }
    for (uint64_t q = 0; q < b->n_live; ++q) {
        const uint32_t k = b->live_idx[q];
        score_out[k] = res[q].score;
        if (end_i_out) end_i_out[k] = res[q].end_i;
        if (end_j_out) end_j_out[k] = res[q].end_j;
    }
    return PWA_OK;
} catch (const std::bad_alloc&) {
    return fail(b ? b->ctx : nullptr, PWA_E_NOMEM, "host allocation failed");
} catch (...) {
    return fail(b ? b->ctx : nullptr, PWA_E_HIP, "unexpected C++ exception");   // nothing may propagate across the C ABI
}

void pwa_batch_destroy(pwa_batch* b) {
    if (!b) return;
    if (b->ctx) (void)hipSetDevice(b->ctx->device);
    // the device buffers go back to the context's free list, not to hipFree (which would wait for the whole device): wait here for
    // this batch's own last run, so that the next batch cannot be handed memory a kernel is still using
    if (b->ran && b->n_runs) {
        const int slot = (int)((b->n_runs - 1) % pwa_batch::kRing);
        if (b->ev1[slot]) (void)hipEventSynchronize(b->ev1[slot]);
    }
    for (int e = 0; e < pwa_batch::kRing; ++e) {
        if (b->ev0[e]) (void)hipEventDestroy(b->ev0[e]);
        if (b->ev1[e]) (void)hipEventDestroy(b->ev1[e]);
    }
    // the hand-off workspace goes back to the context if it is the larger one (and not outrageous): the next batch of
    // the same shape then skips a multi-GB hipMalloc
    if (b->ctx && b->hand.p && b->hand.bytes > b->ctx->hand_cache_bytes && b->hand.bytes <= kBandCacheMax) {
        if (b->ctx->hand_cache) (void)hipFree(b->ctx->hand_cache);
        b->ctx->hand_cache = b->hand.p;
        b->ctx->hand_cache_bytes = b->hand.bytes;
        b->hand.p = nullptr;
        b->hand.bytes = 0;
    }
    delete b;
}

} // extern "C"
// The strip engine addresses its sequence arena with 32-bit offsets (4 GiB per batch object).  The one-shot entry points
// take pair lists of any size: the list is cut into runs of consecutive pairs whose sequences fit one arena, each run is
// one batch object, results land in the caller's vectors at the run's offset (pairs are independent, hw2.cpp:328-338).
static uint64_t arena_limit(const pwa_ctx* ctx) {
    if (ctx->knobs.arena_limit) return ctx->knobs.arena_limit;   // tests
    return 0xffffffffull - (1ull << 20);
}
// r03 (overlap H2D with compute): the runs are PIPELINED -- while the kernels of run k execute, run k + 1 is validated,
// scheduled, coded and uploaded (copy stream, page-locked pieces) and its kernels are queued behind; the host then collects run k.
// Destroyed runs hand their device buffers to the context's free list (no hipFree: it would wait for the run in flight).
// Cutting a list that FITS one arena into several runs, so that the first kernels start before all of the input is on the device, was
// built and measured (PWA_PIPE_RUNS=6) and is not the default: [gpu] hw2_amd -l on 262 144 pairs 150 x 2000 (569 MB): scores pass 89 ms in
// one run, 137 ms in six -- the kernels of that input take 9 ms, the rest is host work per run (alphabet scans, sorts, uploads with their
// synchronisations), which six runs pay six times; 4.5 GB (two arenas): 357 ms in two pipelined runs, 493 ms in six
// (profiles/r03_cli_scale.txt).
template <class Create>
static int scores_in_arena_chunks(pwa_ctx* ctx, const uint64_t* seq_off, uint32_t n_seq, const uint32_t* pair_a, const uint32_t* pair_b,
                                  uint64_t n_pairs, int32_t* score_out, uint32_t* end_i_out, uint32_t* end_j_out, Create&& create) try {
    if (!seq_off || (n_pairs && (!pair_a || !pair_b))) return fail(ctx, PWA_E_INVALID, "null input");
    uint64_t limit = arena_limit(ctx);
    std::vector<uint64_t> stamp(n_seq, 0);
    uint64_t chunk = 0;
    if (!ctx->knobs.arena_limit && ctx->knobs.pipe_runs > 1) {   // experiment: cut a list that fits one arena into PWA_PIPE_RUNS runs
        ++chunk;
        uint64_t total = 512;
        for (uint64_t k = 0; k < n_pairs; ++k)
            for (const uint32_t sidx : {pair_a[k], pair_b[k]}) {
                if (sidx >= n_seq) return fail(ctx, PWA_E_INVALID, "pair index out of range");
                if (stamp[sidx] != chunk) total += align_up(seq_off[sidx + 1] - seq_off[sidx] + 1, 16);
                stamp[sidx] = chunk;
            }
        limit = std::min<uint64_t>(limit, std::max<uint64_t>(1ull << 20, total / (uint64_t)ctx->knobs.pipe_runs + (1ull << 20)));
    }
    struct InFlight {
        pwa_batch* b = nullptr;
        uint64_t k0 = 0;
    } prev;
    auto collect = [&](InFlight& f) -> int {   // wait for the run's own event, copy its results out, recycle its buffers
        if (!f.b) return PWA_OK;
        const int rc = pwa_batch_fetch(f.b, score_out + f.k0, end_i_out ? end_i_out + f.k0 : nullptr, end_j_out ? end_j_out + f.k0 : nullptr);
        pwa_batch_destroy(f.b);
        f.b = nullptr;
        return rc;
    };
    uint64_t k0 = 0, n_runs = 0;
    int rc = PWA_OK;
    const auto t_begin = std::chrono::steady_clock::now();
    while (k0 < n_pairs && rc == PWA_OK) {
        ++chunk;
        ++n_runs;
        uint64_t k1 = k0, bytes = 512;
        for (; k1 < n_pairs; ++k1) {
            uint64_t add = 0;
            for (const uint32_t sidx : {pair_a[k1], pair_b[k1]}) {
                if (sidx >= n_seq) {
                    (void)collect(prev);
                    return fail(ctx, PWA_E_INVALID, "pair index out of range");
                }
                if (stamp[sidx] != chunk) add += align_up(seq_off[sidx + 1] - seq_off[sidx] + 1, 16);
            }
            if (pair_a[k1] == pair_b[k1] && stamp[pair_a[k1]] != chunk) add /= 2;
            if (bytes + add > limit && k1 > k0) break;
            bytes += add;
            stamp[pair_a[k1]] = stamp[pair_b[k1]] = chunk;
        }
        pwa_batch* b = nullptr;
        rc = create(pair_a + k0, pair_b + k0, k1 - k0, &b);   // host work + uploads: overlaps the kernels of the previous run
        // the arena estimate above is not the only 32-bit limit inside a batch object (per-lane text rows of the LANES form, slot
        // arrays): a run that is refused for its size is halved until it fits -- pairs are independent (hw2.cpp:328-338)
        while (rc == PWA_E_CAPACITY && k1 - k0 > 1) {
            k1 = k0 + (k1 - k0) / 2;
            rc = create(pair_a + k0, pair_b + k0, k1 - k0, &b);
        }
        if (rc == PWA_OK) rc = pwa_batch_run(b, nullptr);      // queued behind the previous run on the context's stream
        const int rc_prev = collect(prev);                      // ... whose results are copied out meanwhile
        if (rc == PWA_OK) {
            prev.b = b;
            prev.k0 = k0;
            rc = rc_prev;
            if (ctx->knobs.no_pipeline && rc == PWA_OK) rc = collect(prev);   // A/B: every run is collected before the next one is prepared
        } else if (b) {
            pwa_batch_destroy(b);
        }
        k0 = k1;
    }
    const int rc_last = collect(prev);
    if (ctx->knobs.debug)
        std::fprintf(stderr, "[pwa] scores pass: %llu pairs in %llu pipelined run(s) of <= %llu MB of sequences, %.3f ms in all\n", (unsigned long long)n_pairs,
                     (unsigned long long)n_runs, (unsigned long long)(limit >> 20), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    return rc != PWA_OK ? rc : rc_last;
} catch (const std::bad_alloc&) {
    return fail(ctx, PWA_E_NOMEM, "host allocation failed");
}

extern "C" {
int pwa_scores(pwa_ctx* ctx, int mode, int match, int mismatch, int gap, const uint8_t* seq_bytes,
               const uint64_t* seq_off, uint32_t n_seq, const uint32_t* pair_a, const uint32_t* pair_b,
               uint64_t n_pairs, int32_t* score_out, uint32_t* end_i_out, uint32_t* end_j_out) {
    if (!ctx || !score_out) return PWA_E_INVALID;
    return scores_in_arena_chunks(ctx, seq_off, n_seq, pair_a, pair_b, n_pairs, score_out, end_i_out, end_j_out,
                                  [&](const uint32_t* a, const uint32_t* b, uint64_t n, pwa_batch** out) {
                                      return pwa_batch_create(ctx, mode, match, mismatch, gap, seq_bytes, seq_off, n_seq, a, b, n,
                                                              (end_i_out || end_j_out) ? 1 : 0, out);
                                  });
}

int pwa_distances(pwa_ctx* ctx, int match, int mismatch, int gap, const uint8_t* seq_bytes, const uint64_t* seq_off, uint32_t n_seq,
                  const uint32_t* pair_a, const uint32_t* pair_b, uint64_t n_pairs, int32_t* dist_out) {
    if (!ctx || !dist_out) return PWA_E_INVALID;
    return scores_in_arena_chunks(ctx, seq_off, n_seq, pair_a, pair_b, n_pairs, dist_out, nullptr, nullptr,
                                  [&](const uint32_t* a, const uint32_t* b, uint64_t n, pwa_batch** out) {
                                      return pwa_nwdist_batch_create(ctx, match, mismatch, gap, seq_bytes, seq_off, n_seq, a, b, n, out);
                                  });
}

int pwa_scores_affine(pwa_ctx* ctx, int match, int mismatch, int gap_open, int gap_extend, const uint8_t* seq_bytes,
                      const uint64_t* seq_off, uint32_t n_seq, const uint32_t* pair_a, const uint32_t* pair_b, uint64_t n_pairs,
                      int32_t* score_out) {
    if (!ctx || !score_out) return PWA_E_INVALID;
    return scores_in_arena_chunks(ctx, seq_off, n_seq, pair_a, pair_b, n_pairs, score_out, nullptr, nullptr,
                                  [&](const uint32_t* a, const uint32_t* b, uint64_t n, pwa_batch** out) {
                                      return pwa_affine_batch_create(ctx, match, mismatch, gap_open, gap_extend, seq_bytes, seq_off, n_seq,
                                                                     a, b, n, out);
                                  });
}
}  // extern "C"

// ------------------------------------------------------------------------- full alignments
// Full alignments of a pair list.  With `ops` the op lists come back (pwa_align_batch); without, only the
// per-pair scores and -- with `overlap_out` -- the overlap lengths computed by the walk itself (pwa_overlaps).
//
// Every pair gets the geometry ITS pattern asks for (r03; hw2.cpp:328-338: the reference's loop has no coupling between
// pairs): patterns of up to 256 rows run on the mini-stripe engine (16 lanes per pair, RL = 4 .. 16 rows per lane: the
// smallest RL that holds the pattern), longer ones on the stripe engine with their own (RL, W).  The list is cut into
// RANGES of consecutive pairs whose bands fit the chunk budget; inside a range the pairs of each class form one launch
// (fill + walk); the device op buffer mirrors the caller's regions of the whole range, so the op lists of all its classes
// come back with one copy.
namespace {
struct TbClass {
    bool mini;
    int rl, w;   // mini: rows per lane (w unused); stripe engine: its PairGeom
    bool operator==(const TbClass& o) const { return mini == o.mini && rl == o.rl && w == o.w; }
};
}  // namespace

static int align_batch_impl(pwa_ctx* ctx, int mode, int match, int mismatch, int gap, const uint8_t* seq_bytes,
                            const uint64_t* seq_off, uint32_t n_seq, const uint32_t* pair_a, const uint32_t* pair_b,
                            uint64_t n_pairs, int32_t* score_out, uint8_t* ops, const uint64_t* ops_off, uint64_t* n_ops,
                            uint64_t* end_cells, uint64_t* start_cells, int32_t* overlap_out) try {
    if (!ctx) return PWA_E_INVALID;
    if (mode != PWA_MODE_NW && mode != PWA_MODE_SW) return fail(ctx, PWA_E_INVALID, "unknown mode");
    const bool want_ops = ops != nullptr;
    if (!seq_off || !score_out || (want_ops && (!ops_off || !n_ops)) || (!want_ops && !overlap_out) ||
        (n_pairs && (!pair_a || !pair_b)))
        return fail(ctx, PWA_E_INVALID, "null input");
    if (n_pairs >= 0xffffffffull) return fail(ctx, PWA_E_CAPACITY, "more than 2^32-2 pairs in one batch");
    for (uint64_t k = 0; k < n_pairs; ++k)
        if (pair_a[k] >= n_seq || pair_b[k] >= n_seq) return fail(ctx, PWA_E_INVALID, "pair index out of range");
    HIPC(ctx, hipSetDevice(ctx->device));
    const bool local = mode == PWA_MODE_SW;
    auto slen = [&](uint32_t s) -> uint64_t { return seq_off[s + 1] - seq_off[s]; };
    ctx->fill_ms = ctx->tb_ms = 0.f;
    ctx->band_bytes = 0;
    const bool dbg = ctx->knobs.debug;
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {   // PWA_DEBUG: host-side time between marks
        if (!dbg) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pwa] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };

    // raw-byte arena of the used sequences
    std::vector<uint8_t> is_used(n_seq, 0);
    for (uint64_t k = 0; k < n_pairs; ++k) is_used[pair_a[k]] = is_used[pair_b[k]] = 1;
    std::vector<uint64_t> aoff(n_seq, 0);
    uint64_t arena_bytes = 0;
    for (uint32_t s = 0; s < n_seq; ++s)
        if (is_used[s]) {
            aoff[s] = arena_bytes;
            arena_bytes += align_up(slen(s) + 1, 16);
        }
    arena_bytes += 256;
    // Alphabets of at most 7 symbols (DNA, DNA + N, ...) are stored as codes 0..6 -- equality is all the recurrence
    // ever asks of a symbol (hw2.cpp:142, 208) -- so that the fill can score four rows with one byte-table lookup
    // (pair_fill.hip.h, PERM).  The table holds the two diagonal key constants: both must fit a signed byte.
    bool coded = false;
    uint8_t code_of[256];
    bool dash_seen = false, nul_seen = false;
    {
        bool seen[256] = {false};
        {   // one pass over every used byte: on a few threads once the input reaches megabytes
            bool part[16][256] = {};
            for_seq_ranges(seq_off, n_seq, [&](uint32_t s0, uint32_t s1, int t) {
                bool* mine = part[t];
                for (uint32_t s = s0; s < s1; ++s)
                    if (is_used[s])
                        for (uint64_t o = seq_off[s]; o < seq_off[s + 1]; ++o) mine[seq_bytes[o]] = true;
            }, nullptr, 1ull << 20);
            for (int t = 0; t < 16; ++t)
                for (int v = 0; v < 256; ++v) seen[v] |= part[t][v];
        }
        int n_alpha = 0;
        for (int v = 0; v < 256; ++v) {
            code_of[v] = (uint8_t)std::min(n_alpha, 7);
            if (seen[v]) ++n_alpha;
        }
        const int64_t kd_match = ((int64_t)match - gap) * 4 + 2, kd_mismatch = ((int64_t)mismatch - gap) * 4 + 2;
        coded = n_alpha <= 7 && kd_match <= 127 && kd_match >= -126 && kd_mismatch <= 127 && kd_mismatch >= -126 &&
                !ctx->knobs.no_pair_table;
        dash_seen = seen[(unsigned char)'-'];
        nul_seen = seen[0];
    }
    // overlapLongestExactMatch (hw2.cpp:269) does not count a column whose symbols are '-' -- also when the '-' is part
    // of the input sequence itself: the walk needs the arena's value for that byte
    const int32_t dash_sym = !dash_seen ? 0x100 : (coded ? (int32_t)code_of[(unsigned char)'-'] : (int32_t)'-');
    mark("validate + alphabet scan");
    DevBuf arena_own;
    void* p_arena = nullptr;
    {
        HIPC(ctx, cached_workspace(ctx->pool[pwa_ctx::POOL_ARENA], ctx->pool_bytes[pwa_ctx::POOL_ARENA], arena_bytes, arena_own, &p_arena));
        HIPC(ctx, build_arena(ctx, p_arena, arena_bytes, seq_bytes, seq_off, n_seq, is_used, aoff, coded ? code_of : nullptr, !nul_seen));
    }
    uint8_t* const arena_base = static_cast<uint8_t*>(p_arena);
    mark("arena upload");

    uint64_t longest_sum = 0;
    for (uint64_t k = 0; k < n_pairs; ++k) longest_sum = std::max(longest_sum, slen(pair_a[k]) + slen(pair_b[k]));
    // scores x lengths beyond the packed keys' 2^28: the plain int32 form, exact for anything the reference's int holds
    const bool keyed = tb_range_ok(longest_sum, match, mismatch, gap, local ? 26 : 28) && !ctx->knobs.no_keyed_tb;   // (local: H * 16 in the first-maximum records)
    // Global alignments with table scoring run in gap-shifted coordinates G = H - gap (i + j): the same recurrence with gap 0 and
    // scores s - 2 gap, identical comparisons and codes, one instruction less per cell (pair_fill.hip.h, GAP0).  |G| <= |H| +
    // |gap| (n + m): twice the range; both shifted diagonal constants must fit the byte table.
    bool gap0 = false;
    if (!local && coded && keyed && !ctx->score_band && !ctx->knobs.no_gap_shift) {
        const int64_t sm = (int64_t)match - 2 * (int64_t)gap, sx = (int64_t)mismatch - 2 * (int64_t)gap;
        const int64_t km = sm * 4 + 1, kx = sx * 4 + 1;   // (s' - 0) * 4 + prio(diag) - prio(left)
        const int64_t amax = std::max<int64_t>({std::llabs((long long)match), std::llabs((long long)mismatch), std::llabs((long long)gap), 1});
        gap0 = km <= 127 && km >= -126 && kx <= 127 && kx >= -126 && (longest_sum + 2) <= (1ull << 27) / (uint64_t)amax;
    }
    const int k_match = gap0 ? match - 2 * gap : match, k_mismatch = gap0 ? mismatch - 2 * gap : mismatch, k_gap = gap0 ? 0 : gap;
    // the mini-stripe engine exists for keyed cells with table scoring; PWA_FORCE_RL / PWA_FORCE_W address the stripe engine
    const bool mini_ok = coded && keyed && ctx->knobs.tb_engine != 0 && !ctx->knobs.force_rl && !ctx->knobs.force_w &&
                         (!local || tb_range_ok(longest_sum, match, mismatch, gap, 26));
    // patterns of 257 .. 1024 rows: ONE wave per pair (mini-stripe kernels with 64 lanes per pair, RL = 8 | 16) instead of 4 - 8 pipelined
    // stripes -- when the call has enough of them to occupy the chip that way (a few such pairs are faster spread over more waves)
    uint64_t n_mid = 0;
    for (uint64_t k = 0; k < n_pairs; ++k) {
        const uint64_t n = slen(pair_a[k]);
        n_mid += n > 256 && n <= 1024 && slen(pair_b[k]) > 0;
    }
    const bool wide_ok = mini_ok && (n_mid >= 256 || ctx->knobs.tb_engine == 2);
    // the stripe engine's pairs: RL = 2 gives ONE pair more waves in flight (10 % at 10k x 10k), but a list whose stripes fill the chip anyway
    // runs faster on fewer, taller ones -- [gpu, r03] fills at RL = 2 / RL = 4, NW: 16 pairs 10k x 10k 1.90 / 1.22 ms, 64 pairs 4.12 / 3.33,
    // 256 pairs 12.8 / 8.7, 512 pairs 2000 x 2000 1.07 / 0.75, 32 pairs 30k x 30k 18.0 / 12.4 (profiles/r03_align_shapes.txt)
    uint64_t stripes2 = 0;
    for (uint64_t k = 0; k < n_pairs; ++k) {
        const uint64_t n = slen(pair_a[k]);
        if (!slen(pair_b[k]) || n > 0x7fffffc0ull || (mini_ok && n <= 256) || (wide_ok && n <= 1024)) continue;
        stripes2 += (n + 127) / 128;
    }
    const bool tall_stripes = stripes2 >= 1024 && !ctx->knobs.force_rl;
    auto class_of = [&](uint64_t n) -> TbClass {   // (w of a mini class = its lanes per pair)
        if (mini_ok && n <= 256)
            for (const int rl : kMiniRL)
                if (n <= (uint64_t)(16 * rl)) return TbClass{true, rl, 16};
        if (wide_ok && n <= 1024) return TbClass{true, n <= 384 ? 6 : n <= 512 ? 8 : n <= 768 ? 12 : 16, 64};
        PairGeom g = choose_geom(ctx->knobs, n, keyed, true);
        if (tall_stripes && g.rl == 2 && g.w == 4) g.rl = 4;
        return TbClass{false, g.rl, g.w};
    };
    // band bytes of a pair: the stripe engine's own (also the one-pair-per-wave form's: a single stripe of 64 RL rows); the four-pair
    // mini-stripe form's for a text of m_task columns (its task's longest)
    auto band_of = [&](const TbClass& c, uint64_t n, uint64_t m_task) -> uint64_t {
        if (c.mini && c.w == 16) return (uint64_t)mini_band_steps(m_task) * 16 * (uint64_t)c.rl;
        if (c.mini) return (uint64_t)band_steps(m_task) * 64 * (uint64_t)c.rl;
        return ::tb_band_bytes(n, m_task, c.rl);
    };

    // ---- ranges of consecutive pairs whose traceback bands fit the free HBM
    size_t free_b = 0, total_b = 0;
    HIPC(ctx, hipMemGetInfo(&free_b, &total_b));
    const uint64_t budget = std::max<uint64_t>((uint64_t)(free_b * 0.8), 64ull << 20);
    // A range's band is written once and read along one path per pair, so nothing is gained by a huge one, while
    // hipMalloc gets slow for very large requests ([gpu] profiles/r01_malloc_probe.txt: 0.3 ms up to 8 GiB,
    // 0.24 s for 10.5 GB, >1 s for 16 GiB): ranges of <= 6 GiB of band (enough pairs to fill every CU; or one
    // pair, whatever it needs), all using ONE allocation sized for the largest range.
    // (with the int32 score band on, that one is the large allocation: 2 GiB of codes + 8 GiB of scores)
    // [gpu, r03] But a range is a launch of its own, and a launch takes at least the time ONE wave needs for its longest pair (1.3 ms
    // for a 10k-column text on the mini-stripe engine, whatever the number of pairs): the 4096 x (150 x 10k) batch -- 6.56 GB of
    // band, just over 6 GiB -- ran as 4020 + 76 pairs in 2 x 1.3 ms, and with the score band (32.8 GB) as four launches of 256
    // waves on a chip of 1024 SIMDs.  So: a range should hold ~8192 pairs where the list has them (2048+ waves), it may use up to
    // 48 GiB for that (the workspace is kept in the context: the slow hipMalloc is paid once), and a list that needs several ranges is
    // cut into EQUAL ones, not into full ones and a remainder.
    const uint64_t band_mult = ctx->score_band ? 5 : 1;
    uint64_t chunk_target = std::min<uint64_t>(budget, 8ull << 30);
    uint64_t pairs_target = ~0ull;   // live pairs per range, when the list is cut into several
    {
        uint64_t total = 0, live = 0;
        for (uint64_t k = 0; k < n_pairs; ++k) {
            const uint64_t n = slen(pair_a[k]), m = slen(pair_b[k]);
            if (!(n && m) || n > 0x7fffffc0ull || m > 0x7fffffc0ull) continue;
            total += align_up(band_of(class_of(n), n, m), 256) * band_mult + align_up(n + m + 1, 16);
            ++live;
        }
        if (live) {
            const uint64_t cap = ctx->knobs.range_bytes ? ctx->knobs.range_bytes : std::min<uint64_t>(budget, 48ull << 30);
            const uint64_t for_8192 = (uint64_t)((long double)total / (long double)live * 8192.0L);
            chunk_target = std::min<uint64_t>(cap, std::max<uint64_t>(chunk_target, for_8192));
            const uint64_t n_ranges = (total + chunk_target - 1) / chunk_target;
            if (n_ranges > 1) {
                chunk_target = std::min<uint64_t>(cap, total / n_ranges + total / live + (1ull << 20));   // equal shares (+ one average pair)
                // ... counted in PAIRS, in whole rounds of the chip: a launch lasts as long as its busiest wave, and 8193 pairs are 2049
                // tasks for 2048 wave slots -- [gpu, r03] 16 384 pairs 150 x 10k cut 8193 + 8191: the first fill took 4.1 ms, the second 3.0
                pairs_target = (live + n_ranges - 1) / n_ranges;
                if (pairs_target > 4096) pairs_target = (pairs_target + 4095) / 4096 * 4096;   // 1024 waves of four pairs (or 4 x 1024 of one)
                const uint64_t fit = cap / std::max<uint64_t>(total / live, 1);
                if (pairs_target > fit) pairs_target = std::max<uint64_t>(fit / 4096 * 4096, std::min<uint64_t>(fit, 4096));
                chunk_target = std::min<uint64_t>(cap, std::max<uint64_t>(chunk_target, (uint64_t)((long double)total / (long double)live * (long double)pairs_target * 1.02L)));
            }
        }
    }
    mark("plan: memory + range size");
    struct Launch {                    // the pairs of one class inside one range
        TbClass cls;
        std::vector<uint32_t> q;       // pair index inside the range, in launch order (mini: longest text first)
        std::vector<uint64_t> bo;      // band offset of each (bytes; the int32 score band uses the same offsets in elements)
        std::vector<uint64_t> mt;      // mini: the text length the pair's band is sized for (its task's longest)
        uint64_t dummy_bo[3] = {0, 0, 0};
        uint32_t n_dummy = 0;
    };
    struct Range {
        uint64_t k0, k1, band, opsb;
        bool tiled;      // the caller's op regions ops_off[k] .. + n_k + m_k of the range's pairs follow one another without a gap:
        uint64_t span;   // the device op buffer then mirrors that range and comes back with ONE copy, straight into `ops`
        std::vector<Launch> launches;
    };
    std::vector<Range> ranges;
    uint64_t band_cap = 0, ops_cap_b = 0, nc_cap = 0;
    for (uint64_t k0 = 0; k0 < n_pairs;) {
        uint64_t k1 = k0, est = 0, opsb = 0, live_in = 0;
        while (k1 < n_pairs) {
            const uint64_t n = slen(pair_a[k1]), m = slen(pair_b[k1]);
            if (n > 0x7fffffc0ull || m > 0x7fffffc0ull) return fail(ctx, PWA_E_CAPACITY, "sequence longer than 2^31");
            const uint64_t need = (n && m) ? align_up(band_of(class_of(n), n, m), 256) : 0;
            if (k1 > k0 && ((est + need) * band_mult + opsb + n + m > chunk_target || (need && live_in >= pairs_target))) break;
            live_in += need != 0;
            est += need;
            opsb += align_up(n + m + 1, 16);
            ++k1;
        }
        Range rg{k0, k1, 0, opsb, want_ops, 0, {}};
        // the range's launches: one per class present, pairs in caller order (mini: by text length, so that the four pairs of a
        // wave run about the same number of steps; the band of each is sized for its task's longest text)
        for (uint64_t k = k0; k < k1; ++k) {
            const uint64_t n = slen(pair_a[k]), m = slen(pair_b[k]);
            if (!(n && m)) continue;
            const TbClass c = class_of(n);
            size_t li = 0;
            while (li < rg.launches.size() && !(rg.launches[li].cls == c)) ++li;
            if (li == rg.launches.size()) {
                rg.launches.emplace_back();
                rg.launches.back().cls = c;
            }
            rg.launches[li].q.push_back((uint32_t)(k - k0));
        }
        uint64_t bo = 0;
        for (Launch& L : rg.launches) {
            const size_t np = L.q.size();
            L.bo.resize(np);
            if (L.cls.mini) {
                const size_t ppw = (size_t)(64 / L.cls.w);
                sort_by_length_desc(L.q, [&](uint32_t x) { return slen(pair_b[k0 + x]); });
                L.mt.resize(np);
                for (size_t p = 0; p < np; ++p) L.mt[p] = slen(pair_b[k0 + L.q[p / ppw * ppw]]);   // the task's first pair has its longest text
                L.n_dummy = (uint32_t)((ppw - np % ppw) % ppw);
            }
            for (size_t p = 0; p < np; ++p) {
                L.bo[p] = bo;
                bo += align_up(band_of(L.cls, slen(pair_a[k0 + L.q[p]]), L.cls.mini ? L.mt[p] : slen(pair_b[k0 + L.q[p]])), 256);
            }
            for (uint32_t d = 0; d < L.n_dummy; ++d) {   // the last task's empty patterns write their padding here
                L.dummy_bo[d] = bo;
                bo += align_up(band_of(L.cls, 0, L.mt[np - 1]), 256);
            }
        }
        rg.band = bo;
        if (rg.band * band_mult + opsb > budget && rg.band + opsb > (uint64_t)(free_b * 0.97))
            return fail(ctx, PWA_E_NOMEM, "traceback band of a single pair exceeds free HBM");
        if (want_ops) {
            for (uint64_t k = k0; k < k1; ++k) {
                const uint64_t cap = slen(pair_a[k]) + slen(pair_b[k]);
                if (k + 1 < k1 && ops_off[k + 1] != ops_off[k] + cap) rg.tiled = false;
                rg.span += cap;
            }
            if (ctx->knobs.no_tiled_ops) rg.tiled = false;
        }
        band_cap = std::max(band_cap, rg.band);
        ops_cap_b = std::max(ops_cap_b, std::max(opsb, rg.tiled ? rg.span + 16 : 0));
        nc_cap = std::max(nc_cap, k1 - k0);
        ranges.push_back(std::move(rg));
        k0 = k1;
    }
    mark("plan: ranges + launches");
    DevBuf d_band, d_sband, d_ops_own, d_res_own;
    void *p_band = nullptr, *p_sband = nullptr, *p_ops = nullptr, *p_res = nullptr;
    if (!ranges.empty()) {
        // + one traceback window: the walk stages whole windows
        HIPC(ctx, cached_workspace(ctx->band_cache, ctx->band_cache_bytes, band_cap + 32768, d_band, &p_band));
        if (ctx->score_band)
            HIPC(ctx, cached_workspace(ctx->sband_cache, ctx->sband_cache_bytes, band_cap * sizeof(int32_t), d_sband, &p_sband));
        HIPC(ctx, cached_workspace(ctx->pool[pwa_ctx::POOL_OPS], ctx->pool_bytes[pwa_ctx::POOL_OPS], want_ops ? ops_cap_b : 16, d_ops_own, &p_ops));
        HIPC(ctx, cached_workspace(ctx->pool[pwa_ctx::POOL_RES], ctx->pool_bytes[pwa_ctx::POOL_RES], nc_cap * sizeof(PairResult), d_res_own, &p_res));
    }
    mark("band / ops allocation");
    if (dbg) std::fprintf(stderr, "[pwa] bands at %p (codes, %.2f GB) %p (scores)\n", p_band, (double)band_cap / 1e9, p_sband);
    uint8_t* const d_ops = static_cast<uint8_t*>(p_ops);
    PairResult* const d_res = static_cast<PairResult*>(p_res);
    std::vector<uint8_t> host_ops;   // staging, only for ranges whose op regions do not tile

    for (const Range& rg : ranges) {
        const uint64_t k0 = rg.k0, k1 = rg.k1, opsb = rg.opsb;
        const uint64_t nc = k1 - k0;
        HIPC(ctx, ctx->pin[pwa_ctx::PIN_RES].reserve(nc * sizeof(PairResult)));
        PairResult* const res = ctx->pin[pwa_ctx::PIN_RES].as<PairResult>();   // page-locked: uploaded, and read back after the walk
        std::vector<uint64_t> ooff(nc);
        uint64_t oo = 0;
        const uint64_t ops_lo = (want_ops && nc) ? ops_off[k0] : 0;
        if (want_ops && !rg.tiled) host_ops.resize(opsb);
        for (uint64_t q = 0; q < nc; ++q) {
            const uint64_t k = k0 + q, n = slen(pair_a[k]), m = slen(pair_b[k]);
            std::memset(&res[q], 0, sizeof(PairResult));
            ooff[q] = rg.tiled ? ops_off[k] - ops_lo : oo;
            if (!(n && m) && !local) {
                res[q].score = wrap_mul((int64_t)(n + m), gap);
                res[q].end_i = (uint32_t)n;
                res[q].end_j = (uint32_t)m;
            }
            oo += align_up(n + m + 1, 16);
        }
        HIPC(ctx, hipMemcpy(d_res, res, nc * sizeof(PairResult), hipMemcpyHostToDevice));
        mark("range results init");
        for (const Launch& L : rg.launches) {
            const size_t np = L.q.size();
            std::vector<PairDesc> pd;
            pd.reserve(np + L.n_dummy);
            for (size_t p = 0; p < np; ++p) {
                const uint64_t q = L.q[p], k = k0 + q, n = slen(pair_a[k]), m = slen(pair_b[k]);
                PairDesc d;
                std::memset(&d, 0, sizeof d);
                d.pat = arena_base + aoff[pair_a[k]];
                d.txt = arena_base + aoff[pair_b[k]];
                d.n = (int32_t)n;
                d.m = (int32_t)m;
                d.tb = static_cast<uint8_t*>(p_band) + L.bo[p];
                if (ctx->score_band) d.sband = static_cast<int32_t*>(p_sband) + L.bo[p];
                d.res = d_res + q;
                d.ops = want_ops ? d_ops + ooff[q] : d_ops;   // WALK_OVERLAP never writes ops
                d.ops_cap = (uint32_t)std::min<uint64_t>(n + m, 0xffffffffu);
                d.score_bias = gap0 ? wrap_mul((int64_t)(n + m), gap) : 0;
                pd.push_back(d);
                ctx->band_bytes += band_of(L.cls, n, L.cls.mini ? L.mt[p] : m) * band_mult;
            }
            for (uint32_t dmy = 0; dmy < L.n_dummy; ++dmy) {   // empty patterns that fill the last task: every cell of theirs is padding
                PairDesc d = pd[np - 1];
                d.n = 0;
                d.tb = static_cast<uint8_t*>(p_band) + L.dummy_bo[dmy];
                if (ctx->score_band) d.sband = static_cast<int32_t*>(p_sband) + L.dummy_bo[dmy];
                pd.push_back(d);
            }
            PairLaunch pl;
            pl.from_pool = true;
            pl.perm = coded && keyed;
            pl.keyed = keyed;
            pl.gap0 = gap0;
            int rc = L.cls.mini ? pl.build_mini(ctx, pd, (uint32_t)np, k_match, k_mismatch, k_gap, L.cls.rl, L.cls.w)
                                : pl.build(ctx, pd, k_match, k_mismatch, k_gap, PairGeom{L.cls.rl, L.cls.w});
            if (rc != PWA_OK) return rc;
            pl.G.dash = dash_sym;
            mark("task list build + upload");
            if (dbg) std::fprintf(stderr, "[pwa] fill launch %s RL=%d W|LN=%d grid=%u pairs=%u tasks=%u rows=%llu\n", L.cls.mini ? "mini" : "stripes", L.cls.rl,
                                  L.cls.w, pl.grid, pl.G.n_pairs, pl.G.n_tasks, (unsigned long long)pl.row_bytes);
            HIPC(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
            rc = pl.launch(ctx, ctx->stream, local, true, want_ops ? WALK_OPS : WALK_OVERLAP, ctx->ev[1], ctx->score_band);
            if (rc != PWA_OK) return rc;
            HIPC(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
            HIPC(ctx, hipStreamSynchronize(ctx->stream));
            mark("fill + walk (device)");
            rc = pl.check(ctx);
            if (rc != PWA_OK) return rc;
            float a = 0, c = 0;
            HIPC(ctx, hipEventElapsedTime(&a, ctx->ev[0], ctx->ev[1]));
            HIPC(ctx, hipEventElapsedTime(&c, ctx->ev[1], ctx->ev[2]));
            ctx->fill_ms += a;
            ctx->tb_ms += c;
        }
        HIPC(ctx, hipMemcpy(res, d_res, nc * sizeof(PairResult), hipMemcpyDeviceToHost));
        if (want_ops && rg.tiled && rg.span) HIPC(ctx, hipMemcpy(ops + ops_lo, d_ops, rg.span, hipMemcpyDeviceToHost));   // straight into the caller's list
        if (want_ops && !rg.tiled) HIPC(ctx, hipMemcpy(host_ops.data(), d_ops, opsb, hipMemcpyDeviceToHost));
        mark("results (+ ops) to host");
        if (dbg && want_ops) {   // the op-list walk leaves its LDS round trips in `overlap` (unused by that walk)
            uint64_t trips = 0, nops = 0;
            for (uint64_t q = 0; q < nc; ++q) trips += res[q].overlap, nops += res[q].n_ops;
            std::fprintf(stderr, "[pwa] walk: %llu ops in %llu trips\n", (unsigned long long)nops, (unsigned long long)trips);
        }
        for (uint64_t q = 0; q < nc; ++q) {
            const uint64_t k = k0 + q, n = slen(pair_a[k]), m = slen(pair_b[k]);
            uint64_t cnt = res[q].n_ops;
            if (!(n && m)) {
                // one side empty: NW walks the boundary (hw2.cpp:170-179), SW emits nothing (239); no column
                // without a gap, so the overlap is 0 (hw2.cpp:267-278)
                cnt = local ? 0 : n + m;
                if (want_ops)
                    for (uint64_t o = 0; o < cnt; ++o) ops[ops_off[k] + o] = n ? 'D' : 'I';
                if (start_cells) start_cells[2 * k] = start_cells[2 * k + 1] = 0;
                if (overlap_out) overlap_out[k] = 0;
            } else {
                if (res[q].overflow) return fail(ctx, PWA_E_CAPACITY, "internal: traceback longer than n+m");
                if (want_ops && !rg.tiled) std::memcpy(ops + ops_off[k], host_ops.data() + ooff[q], cnt);
                if (start_cells) {
                    start_cells[2 * k] = res[q].start_i;
                    start_cells[2 * k + 1] = res[q].start_j;
                }
                if (overlap_out) overlap_out[k] = res[q].overlap;
            }
            score_out[k] = res[q].score;
            if (n_ops) n_ops[k] = cnt;
            if (end_cells) {
                end_cells[2 * k] = res[q].end_i;
                end_cells[2 * k + 1] = res[q].end_j;
            }
        }
        mark("scatter to caller buffers");
    }
    if (dbg) std::fprintf(stderr, "[pwa] %s: %llu pairs in %zu range(s): fills %.3f ms, walks %.3f ms (device), %.2f GB of band written\n", want_ops ? "align_batch" : "overlaps",
                          (unsigned long long)n_pairs, ranges.size(), ctx->fill_ms, ctx->tb_ms, (double)ctx->band_bytes / 1e9);
    return PWA_OK;
} catch (const std::bad_alloc&) {
    return fail(ctx, PWA_E_NOMEM, "host allocation failed");
} catch (...) {
    return fail(ctx, PWA_E_HIP, "unexpected C++ exception");   // nothing may propagate across the C ABI
}

extern "C" {

int pwa_align_batch(pwa_ctx* ctx, int mode, int match, int mismatch, int gap, const uint8_t* seq_bytes,
                    const uint64_t* seq_off, uint32_t n_seq, const uint32_t* pair_a, const uint32_t* pair_b,
                    uint64_t n_pairs, int32_t* score_out, uint8_t* ops, const uint64_t* ops_off, uint64_t* n_ops,
                    uint64_t* end_cells, uint64_t* start_cells) {
    if (ctx && !ops) return fail(ctx, PWA_E_INVALID, "null input");
    return align_batch_impl(ctx, mode, match, mismatch, gap, seq_bytes, seq_off, n_seq, pair_a, pair_b, n_pairs, score_out, ops,
                            ops_off, n_ops, end_cells, start_cells, nullptr);
}

int pwa_overlaps(pwa_ctx* ctx, int mode, int match, int mismatch, int gap, const uint8_t* seq_bytes, const uint64_t* seq_off,
                 uint32_t n_seq, const uint32_t* pair_a, const uint32_t* pair_b, uint64_t n_pairs, int32_t* score_out,
                 int32_t* overlap_out) {
    if (ctx && !overlap_out) return fail(ctx, PWA_E_INVALID, "null input");
    return align_batch_impl(ctx, mode, match, mismatch, gap, seq_bytes, seq_off, n_seq, pair_a, pair_b, n_pairs, score_out, nullptr,
                            nullptr, nullptr, nullptr, nullptr, overlap_out);
}

int pwa_align(pwa_ctx* ctx, int mode, int match, int mismatch, int gap, const uint8_t* pattern, uint64_t n,
              const uint8_t* text, uint64_t m, int32_t* score, uint8_t* ops, uint64_t ops_cap, uint64_t* n_ops,
              uint64_t end_cell[2], uint64_t start_cell[2]) try {
    if (!ctx) return PWA_E_INVALID;
    if (!score || !ops || !n_ops || (n && !pattern) || (m && !text)) return fail(ctx, PWA_E_INVALID, "null input");
    if (ops_cap < n + m) return fail(ctx, PWA_E_CAPACITY, "ops_cap must be at least n + m");
    std::vector<uint8_t> bytes(n + m);
    if (n) std::memcpy(bytes.data(), pattern, n);
    if (m) std::memcpy(bytes.data() + n, text, m);
    const uint64_t off[3] = {0, n, n + m};
    const uint32_t a = 0, b = 1;
    const uint64_t ooff = 0;
    return pwa_align_batch(ctx, mode, match, mismatch, gap, bytes.data(), off, 2, &a, &b, 1, score, ops, &ooff, n_ops,
                           end_cell, start_cell);
} catch (const std::bad_alloc&) {
    return fail(ctx, PWA_E_NOMEM, "host allocation failed");
} catch (...) {
    return fail(ctx, PWA_E_HIP, "unexpected C++ exception");   // nothing may propagate across the C ABI
}

int pwa_align_matrices(pwa_ctx* ctx, int mode, int match, int mismatch, int gap, const uint8_t* pattern, uint64_t n,
                       const uint8_t* text, uint64_t m, int32_t* dp_out, char* tb_out) try {
    if (!ctx) return PWA_E_INVALID;
    if (mode != PWA_MODE_NW && mode != PWA_MODE_SW) return fail(ctx, PWA_E_INVALID, "unknown mode");
    if ((n && !pattern) || (m && !text) || (!dp_out && !tb_out)) return fail(ctx, PWA_E_INVALID, "null input");
    if (n > 0x7fffffc0ull || m > 0x7fffffc0ull) return fail(ctx, PWA_E_CAPACITY, "sequence longer than 2^31");
    const bool keyed = tb_range_ok(n + m, match, mismatch, gap, mode == PWA_MODE_SW ? 26 : 28) && !ctx->knobs.no_keyed_tb;
    const bool local = mode == PWA_MODE_SW;
    const uint64_t W = m + 1;
    // row 0 and column 0 exactly as the reference initialises them (hw2.cpp:119-136 / 193-194)
    for (uint64_t i = 0; i <= n; ++i) {
        if (dp_out) dp_out[i * W] = local ? 0 : wrap_mul((int64_t)i, gap);
        if (tb_out) tb_out[i * W] = (!local && i > 0) ? 'u' : ' ';
    }
    for (uint64_t j = 0; j <= m; ++j) {
        if (dp_out) dp_out[j] = local ? 0 : wrap_mul((int64_t)j, gap);
        if (tb_out) tb_out[j] = (!local && j > 0) ? 'l' : ' ';
    }
    if (n == 0 || m == 0) return PWA_OK;
    HIPC(ctx, hipSetDevice(ctx->device));
    // patterns of up to 256 rows over an alphabet of <= 7 symbols: the mini-stripe engine, as pwa_align_batch would pick it (so that
    // the whole-matrix comparison covers that engine's cells too); everything else: the stripe engine on raw bytes
    uint8_t code_of[256];
    int n_alpha = 0;
    {
        bool seen[256] = {false};
        for (uint64_t o = 0; o < n; ++o) seen[pattern[o]] = true;
        for (uint64_t o = 0; o < m; ++o) seen[text[o]] = true;
        for (int v = 0; v < 256; ++v) {
            code_of[v] = (uint8_t)std::min(n_alpha, 7);
            if (seen[v]) ++n_alpha;
        }
    }
    const int64_t kd_match = ((int64_t)match - gap) * 4 + 2, kd_mismatch = ((int64_t)mismatch - gap) * 4 + 2;
    const bool coded = n_alpha <= 7 && kd_match <= 127 && kd_match >= -126 && kd_mismatch <= 127 && kd_mismatch >= -126 && !ctx->knobs.no_pair_table;
    int mini_rl = 0, wide_rl = 0;   // wide: one pair per wave (PWA_TB_ENGINE=2 here: a single pair would normally take pipelined stripes)
    if (coded && keyed && ctx->knobs.tb_engine != 0 && !ctx->knobs.force_rl && !ctx->knobs.force_w &&
        (mode != PWA_MODE_SW || tb_range_ok(n + m, match, mismatch, gap, 26))) {
        if (n <= 256) {
            for (const int rl : kMiniRL)
                if (!mini_rl && n <= (uint64_t)(16 * rl)) mini_rl = rl;
        } else if (n <= 1024 && ctx->knobs.tb_engine == 2) {
            wide_rl = n <= 384 ? 6 : n <= 512 ? 8 : n <= 768 ? 12 : 16;
        }
    }
    const PairGeom geom = mini_rl ? PairGeom{mini_rl, 1} : wide_rl ? PairGeom{wide_rl, 1} : choose_geom(ctx->knobs, n, keyed, true);
    const uint64_t kRL = (uint64_t)geom.rl;
    const uint64_t band = mini_rl ? (uint64_t)mini_band_steps(m) * 16 * kRL : tb_band_bytes(n, m, geom.rl);   // (wide: one 64 RL-row stripe)
    DevBuf d_pat, d_txt, d_band, d_sband, d_res;
    HIPC(ctx, d_pat.alloc(n + 64));
    HIPC(ctx, d_txt.alloc(m + 64));
    HIPC(ctx, d_band.alloc((mini_rl ? 4 : 1) * band + 32768));
    HIPC(ctx, d_sband.alloc((mini_rl ? 4 : 1) * band * sizeof(int32_t)));
    HIPC(ctx, d_res.alloc(sizeof(PairResult)));
    if (mini_rl || wide_rl) {
        std::vector<uint8_t> cp(n), ct(m);
        for (uint64_t o = 0; o < n; ++o) cp[o] = code_of[pattern[o]];
        for (uint64_t o = 0; o < m; ++o) ct[o] = code_of[text[o]];
        HIPC(ctx, upload_via_bounce(ctx, d_pat.p, cp.data(), n));
        HIPC(ctx, upload_via_bounce(ctx, d_txt.p, ct.data(), m));
    } else {
        HIPC(ctx, hipMemcpy(d_pat.p, pattern, n, hipMemcpyHostToDevice));
        HIPC(ctx, hipMemcpy(d_txt.p, text, m, hipMemcpyHostToDevice));
    }
    HIPC(ctx, hipMemset(d_res.p, 0, sizeof(PairResult)));
    std::vector<PairDesc> pd(1);
    std::memset(&pd[0], 0, sizeof(PairDesc));
    pd[0].pat = d_pat.as<uint8_t>();
    pd[0].txt = d_txt.as<uint8_t>();
    pd[0].n = (int32_t)n;
    pd[0].m = (int32_t)m;
    pd[0].tb = d_band.as<uint8_t>();
    pd[0].sband = d_sband.as<int32_t>();
    pd[0].res = d_res.as<PairResult>();
    PairLaunch pl;
    pl.keyed = keyed;
    int rc;
    if (mini_rl) {
        for (int d = 1; d < 4; ++d) {   // three empty patterns fill the wave; their padding goes behind the pair's bands
            PairDesc e = pd[0];
            e.n = 0;
            e.tb = d_band.as<uint8_t>() + (uint64_t)d * band;
            e.sband = d_sband.as<int32_t>() + (uint64_t)d * band;
            pd.push_back(e);
        }
        pl.perm = true;
        rc = pl.build_mini(ctx, pd, 1, match, mismatch, gap, mini_rl);
    } else if (wide_rl) {
        pl.perm = true;
        rc = pl.build_mini(ctx, pd, 1, match, mismatch, gap, wide_rl, 64);
    } else {
        rc = pl.build(ctx, pd, match, mismatch, gap, geom);
    }
    if (rc != PWA_OK) return rc;
    rc = pl.launch(ctx, ctx->stream, local, true, false, nullptr, true);
    if (rc != PWA_OK) return rc;
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    rc = pl.check(ctx);
    if (rc != PWA_OK) return rc;
    std::vector<uint8_t> hb(tb_out ? band : 0);
    std::vector<int32_t> hs(dp_out ? band : 0);
    if (tb_out) HIPC(ctx, hipMemcpy(hb.data(), d_band.p, band, hipMemcpyDeviceToHost));
    if (dp_out) HIPC(ctx, hipMemcpy(hs.data(), d_sband.p, band * sizeof(int32_t), hipMemcpyDeviceToHost));
    // band codes are tie-break priorities (pair_fill.hip.h): global up 0, left 1, diag 2; local left 0, up 1, diag 2, floor 3
    static const char kCodeNW[4] = {'u', 'l', 'd', 'd'}, kCodeSW[4] = {'l', 'u', 'd', '0'};   // hw2.cpp:145-153 / 214-222
    const char* const kCode = local ? kCodeSW : kCodeNW;
    const uint64_t T = band_steps(m);
    const uint64_t PA = kRL >= 16 ? 16 : (kRL >= 8 ? 8 : 4), PB = kRL - PA;   // BandGeo<LN, RL> of the mini-stripe kernels
    const uint64_t LN = mini_rl ? 16 : 64, Q4 = kRL & ~(uint64_t)3, W4 = kRL & 3;
    for (uint64_t i = 1; i <= n; ++i) {
        const uint64_t q = i - 1;
        for (uint64_t j = 1; j <= m; ++j) {
            uint64_t idx, sidx;   // skewed bands -> row-major matrix
            if (mini_rl || wide_rl) {
                const uint64_t k = q / kRL, r = q % kRL, t = j - 1 + k;
                idx = t * LN * kRL + (r < PA ? k * PA + r : LN * PA + k * PB + (r - PA));
                sidx = t * LN * kRL + (r < Q4 ? (r >> 2) * (LN * 4) + k * 4 + (r & 3) : Q4 * LN + k * W4 + (r & 3));   // BandGeo::sband_off
            } else {
                const uint64_t st = q / (64 * kRL), k = (q % (64 * kRL)) / kRL, r = q % kRL;
                idx = sidx = ((st * T + (j - 1 + k)) * 64 + k) * kRL + r;
            }
            if (tb_out) tb_out[i * W + j] = kCode[hb[idx] & 3];
            if (dp_out) dp_out[i * W + j] = hs[sidx];
        }
    }
    return PWA_OK;
} catch (const std::bad_alloc&) {
    return fail(ctx, PWA_E_NOMEM, "host allocation failed");
} catch (...) {
    return fail(ctx, PWA_E_HIP, "unexpected C++ exception");   // nothing may propagate across the C ABI
}

int pwa_align_last_stats(const pwa_ctx* ctx, float* fill_ms, float* traceback_ms, uint64_t* band_bytes) {
    if (!ctx) return PWA_E_INVALID;
    if (fill_ms) *fill_ms = ctx->fill_ms;
    if (traceback_ms) *traceback_ms = ctx->tb_ms;
    if (band_bytes) *band_bytes = ctx->band_bytes;
    return PWA_OK;
}

}  // extern "C"
