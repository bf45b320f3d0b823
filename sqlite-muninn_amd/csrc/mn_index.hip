// mn_index.hip — host side of libmuninn_hip.so: the C-ABI of include/muninn_hip.h over a
// device-resident HNSW index.  No CPU compute path exists here: distances, searches and link
// updates are HIP kernels (mn_kernels.hip, mn_build.hip); the host only keeps the id→slot table,
// per-node metadata, the level RNG (src/hnsw_algo.c:19-30,240-248) and the cold delete path.
#include "../../include/muninn_hip.h"
#include "mn_guard.hpp"
#include "mn_device.hpp"
#include "mn_comm.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <mutex>
#include <vector>

#define MN_MAX_M 512        // longest supported M (lists of 2M + 1 entries are pruned in LDS)
#define MN_MAX_ROW 2048     // longest neighbour list a row may grow to through deletes / loads (same LDS budget)

static thread_local std::string g_err;

static void set_err(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

// ── host allocations of this library: countable, and the n-th one can be made to fail (test hook of the exception barrier,
//    mn_guard.hpp).  The linker's export map (exports.map: mn_* only) keeps the replacement private to this shared object: it
//    serves the new-expressions compiled into it (std::vector, std::string bodies instantiated here) and nothing else in the
//    process. ──
#include <atomic>
static std::atomic<long long> g_alloc_count{0}, g_alloc_fail_at{0};
static inline void *mn_counted_alloc(size_t n) {
    const long long c = g_alloc_count.fetch_add(1, std::memory_order_relaxed) + 1;
    if (c == g_alloc_fail_at.load(std::memory_order_relaxed))
        throw std::bad_alloc();
    void *p = malloc(n ? n : 1);
    if (!p)
        throw std::bad_alloc();
    return p;
}
void *operator new(size_t n) { return mn_counted_alloc(n); }
void *operator new[](size_t n) { return mn_counted_alloc(n); }
void operator delete(void *p) noexcept { free(p); }
void operator delete[](void *p) noexcept { free(p); }
void operator delete(void *p, size_t) noexcept { free(p); }
void operator delete[](void *p, size_t) noexcept { free(p); }
extern "C" long long mn_debug_fault_alloc(long long nth) {
    g_alloc_fail_at.store(nth > 0 ? nth : 0, std::memory_order_relaxed);
    return g_alloc_count.exchange(0, std::memory_order_relaxed);
}

#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) {                                                          \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return -1;                                                                    \
        }                                                                                 \
    } while (0)

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0; // elements
    // `want` (> n, optional): a size the caller knows the buffer will reach (a bulk build) — taken exactly, in ONE allocation,
    // instead of the geometric steps and their 1.5x slack
    int reserve(size_t n, bool keep, hipStream_t st, int fill_byte = -1, size_t want = 0) {
        if (n <= cap)
            return 0;
        size_t nc = cap ? cap : 1024;
        while (nc < n)
            nc = nc + nc / 2 + 1024;
        if (want > n)
            nc = want;
        T *np = nullptr;
        HIPCHK(hipMalloc(&np, nc * sizeof(T)));
        if (fill_byte >= 0)
            HIPCHK(hipMemsetAsync(np, fill_byte, nc * sizeof(T), st));
        if (keep && p && cap)
            HIPCHK(hipMemcpyAsync(np, p, cap * sizeof(T), hipMemcpyDeviceToDevice, st));
        if (p) {
            HIPCHK(hipStreamSynchronize(st));
            HIPCHK(hipFree(p));
        }
        p = np;
        cap = nc;
        return 0;
    }
    void release() {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct mn_index {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    mn_build_stats bstats = {0, 0, 0, 0, 0, 0};
    int dim = 0, ld = 0, metric = 0, order = MN_ORDER_SSE, M = 0, M_max0 = 0, efc = 0;
    int W0 = 0, WU = 0; // row strides of links0 / links_up (>= M_max0 / M; grown when a list must exceed M_max)
    double level_mult = 0;
    int64_t entry_id = -1;
    int max_level = -1;
    unsigned rng_state = 42;
    int node_count = 0;
    int n_deleted = 0; // soft-deleted nodes among the slots
    bool last_on_host = false; // `last` already holds the counters of the last search (the few-queries path)
    int64_t slot_hint = 0; // slots a running bulk build will reach (sync_meta sizes the device tables for it at once)
    // host metadata, slot-indexed
    std::vector<int64_t> ids;
    std::vector<signed char> levels;
    std::vector<unsigned char> deleted;
    std::vector<int> up_off;
    int n_slots = 0, n_pool_rows = 0;
    int meta_uploaded = 0; // slots whose metadata is on the device
    // reference-compatible open-addressing table id → slot (src/hnsw_algo.c:38-91)
    std::vector<int> ht;
    int ht_cap = 256;
    // device state
    DevBuf<float> d_vectors, d_norms;
    DevBuf<int> d_links0, d_links_up, d_up_off;
    DevBuf<signed char> d_levels;
    DevBuf<unsigned char> d_deleted, d_dirty;
    DevBuf<long long> d_ids;
    // host mirror of the link rows (pulled on demand for delete / inspection / load)
    std::vector<int> h_links0, h_links_up;
    bool host_links_valid = true;   // host mirror == device
    bool dev_links_stale = false;   // host mirror modified, device not yet updated
    bool dev_links_all = false;     // ... and the whole table has to go (load, re-stride); otherwise only h_dirty_rows
    std::vector<std::pair<int, int>> h_dirty_rows; // (slot, level) rows of the mirror edited since the last push
    // nodes appended by mn_hnsw_load_node whose vectors wait on the host for one bulk upload
    std::vector<float> load_vecs;
    int load_first = -1;
    // workspaces
    DevBuf<unsigned> ws_bm0, ws_bmu;
    DevBuf<uint2> ws_cand, ws_res;
    DevBuf<float> ws_q, ws_outd;
    DevBuf<long long> ws_outi;
    DevBuf<int> ws_outc, ws_qslots, ws_sel, ws_nsel, ws_upidx;
    DevBuf<int> lk_target, lk_src, lk_counters, lk_count, lk_fill, lk_binoff, lk_touched, lk_bins, lk_newrows;
    DevBuf<int> lk_rec, lk_rec_all, lk_cls; // divided link step of the jointly built graph: this rank's records, everybody's, class counts
    DevBuf<unsigned long long> ws_counters;
    DevBuf<int> ws_state;
    DevBuf<int> er_slot, er_level, er_nbr;
    DevBuf<float> er_dist;
    // speculative exact build: read logs of a window's searches, per-row rewrite epochs
    DevBuf<int> ws_readlog, ws_nread, ws_ncommit, d_stamp0, d_stampU, d_sidx0, d_sidxU, d_saved_rows, d_pre_act, d_pre_cnt, d_pre_row, d_spec_why;
    int spec_epoch = 0;
    std::vector<int> staged; // mn_hnsw_batch_stage: slots added but not yet searched / linked
    DevBuf<int> d_staged;
    // multi-GPU exchange buffers (mn_hnsw_build_shared / mn_hnsw_search_sharded)
    DevBuf<int> sh_sel, sh_nsel, sh_gcnt, sh_lcnt;
    DevBuf<int> ws_chlog;            // change log of a single exact insert (mn_hnsw_insert_logged)
    bool want_chlog = false;         // the next run_sequential of one node fills it
    std::vector<int> h_chlog;        // its host copy: [0] entries or -1, then MN_CHLOG_INTS ints each
    // slots whose lists a delete edited: the reference leaves those edits un-persisted (src/hnsw_vtab.c:702-706) until a later
    // insert re-persists the node WHOLE, so an insert that touches one cannot be described by a change log
    std::vector<unsigned char> h_stale;
    int n_stale = 0;
    // pinned host block the kernels of a small search read the queries from and write the answers to (one query per call is
    // the SQL surface's shape: five pageable copies and two synchronisations cost more than the search)
    unsigned char *pin = nullptr;
    size_t pin_cap = 0;
    DevBuf<unsigned long long> sh_ovf; // [world] heap-workspace overflow counts of the last sharded search, all-gathered
    int sh_ovf_pending = 0;             // entries of sh_ovf not yet looked at by the host (0: none)
    DevBuf<long long> sh_gids, sh_lids;
    DevBuf<float> sh_gd, sh_ld;
    long long last_spec_searched = 0; // searches the last speculative build ran (≥ nodes inserted)
    bool broken = false; // an insert failed after its kernels had begun to rewrite link rows: nothing can be trusted
    mn_launch_stats last = {0, 0, 0, 0};
};

// ───────────────────────── small host helpers ─────────────────────────

static unsigned xorshift32(unsigned *state) { // src/hnsw_algo.c:19-26
    unsigned x = *state;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    *state = x;
    return x;
}

static int random_level(mn_index *x) { // src/hnsw_algo.c:240-248
    double r = (double)xorshift32(&x->rng_state) / (double)0xFFFFFFFFu;
    if (r == 0.0)
        r = 1e-10;
    int level = (int)(-log(r) * x->level_mult);
    if (level >= 32)
        level = 31;
    return level;
}

static int ht_hash(int64_t id, int cap) { // src/hnsw_algo.c:38-47
    uint64_t h = (uint64_t)id;
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdULL;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ULL;
    h ^= h >> 33;
    return (int)(h & (uint64_t)(cap - 1));
}

static int ht_find(const mn_index *x, int64_t id) {
    int s = ht_hash(id, x->ht_cap);
    for (int i = 0; i < x->ht_cap; i++) {
        int p = (s + i) & (x->ht_cap - 1);
        if (x->ht[p] < 0)
            return -1;
        if (x->ids[x->ht[p]] == id)
            return x->ht[p];
    }
    return -1;
}

static int ht_put(std::vector<int> &t, int cap, const std::vector<int64_t> &ids, int slot) {
    int s = ht_hash(ids[slot], cap);
    for (int i = 0; i < cap; i++) {
        int p = (s + i) & (cap - 1);
        if (t[p] < 0) {
            t[p] = slot;
            return 0;
        }
        if (ids[t[p]] == ids[slot])
            return -1;
    }
    return -1;
}

static void ht_grow(mn_index *x) { // src/hnsw_algo.c:76-91: rehash in old-table order
    int nc = x->ht_cap * 2;
    std::vector<int> nt((size_t)nc, -1);
    for (int i = 0; i < x->ht_cap; i++)
        if (x->ht[i] >= 0)
            ht_put(nt, nc, x->ids, x->ht[i]);
    x->ht.swap(nt);
    x->ht_cap = nc;
}

// The reference grows its table on the LIVE count (src/hnsw_algo.c:527) while soft-deleted nodes keep their entries, so
// delete + insert churn can fill it; ht_insert then fails ("table full", :61-74) and hnsw_insert returns -1.  Index of the
// first of n inserts that would hit that, or -1.
static int64_t first_table_full(const mn_index *x, int64_t n) {
    long long cap = x->ht_cap, live = x->node_count, occ = x->n_slots;
    for (int64_t i = 0; i < n; i++) {
        if (live * 10 > cap * 7)
            cap *= 2;
        if (occ >= cap)
            return i;
        live++;
        occ++;
    }
    return -1;
}

// what insert_impl needs to take back the nodes it appended when it fails before any link row was rewritten
struct InsertUndo {
    int first, node_count, max_level, n_pool_rows, old_cap = 0, grew_at = -1;
    unsigned rng;
    int64_t entry_id;
    std::vector<int> old_ht; // the table as it stood just before the call's first growth
};

static void ht_erase_newest(std::vector<int> &t, int cap, const std::vector<int64_t> &ids, int slot) {
    // entries are erased newest first, so each one is the last link of its probe chain: clearing it restores the table
    int s = ht_hash(ids[slot], cap);
    for (int i = 0; i < cap; i++) {
        int p = (s + i) & (cap - 1);
        if (t[p] == slot) {
            t[p] = -1;
            return;
        }
    }
}

static void undo_insert(mn_index *x, InsertUndo &u) {
    int hi = x->n_slots;
    if (u.grew_at >= 0) {
        x->ht.swap(u.old_ht);
        x->ht_cap = u.old_cap;
        hi = u.grew_at;
    }
    for (int s = hi - 1; s >= u.first; s--)
        ht_erase_newest(x->ht, x->ht_cap, x->ids, s);
    x->ids.resize((size_t)u.first);
    x->levels.resize((size_t)u.first);
    x->deleted.resize((size_t)u.first);
    x->up_off.resize((size_t)u.first);
    x->n_slots = u.first;
    x->n_pool_rows = u.n_pool_rows;
    x->node_count = u.node_count;
    x->rng_state = u.rng;
    x->entry_id = u.entry_id;
    x->max_level = u.max_level;
    x->meta_uploaded = std::min(x->meta_uploaded, u.first);
    x->staged.clear();
    if (x->host_links_valid) {
        x->h_links0.resize((size_t)x->n_slots * x->W0);
        x->h_links_up.resize((size_t)x->n_pool_rows * x->WU);
    }
}

static int use_device(mn_index *x) {
    if (x->broken) {
        set_err("index unusable: an earlier insert failed after it had begun to rewrite neighbour rows");
        return -1;
    }
    HIPCHK(hipSetDevice(x->device));
    return 0;
}

static MnDevIndex dev_view(mn_index *x) {
    MnDevIndex v;
    v.vectors = x->d_vectors.p;
    v.norms = x->d_norms.p;
    v.links0 = x->d_links0.p;
    v.links_up = x->d_links_up.p;
    v.up_off = x->d_up_off.p;
    v.levels = x->d_levels.p;
    v.deleted = x->d_deleted.p;
    v.dirty = x->d_dirty.p;
    v.ids = x->d_ids.p;
    v.dim = x->dim;
    v.ld = x->ld;
    v.metric = x->metric;
    v.order = x->order;
    v.W0 = x->W0;
    v.WU = x->WU;
    v.M0 = x->M_max0;
    v.MU = x->M;
    v.WX = std::max(x->W0, x->WU);
    v.n_slots = x->n_slots;
    v.n_pool_rows = x->n_pool_rows;
    v.has_deleted = x->n_deleted > 0;
    return v;
}

// append a node to the host tables (node_create + ht_insert); returns slot
static int host_add_node(mn_index *x, int64_t id, int level, int deleted) {
    int s = x->n_slots;
    x->ids.push_back(id);
    x->levels.push_back((signed char)level);
    x->deleted.push_back((unsigned char)deleted);
    if (level > 0) {
        x->up_off.push_back(x->n_pool_rows);
        x->n_pool_rows += level;
    } else {
        x->up_off.push_back(-1);
    }
    x->n_slots++;
    if (ht_put(x->ht, x->ht_cap, x->ids, s) != 0) {
        x->ids.pop_back();
        x->levels.pop_back();
        x->deleted.pop_back();
        if (level > 0)
            x->n_pool_rows -= level;
        x->up_off.pop_back();
        x->n_slots--;
        return -1;
    }
    if (x->host_links_valid) {
        x->h_links0.resize((size_t)x->n_slots * x->W0, -1);
        x->h_links_up.resize((size_t)x->n_pool_rows * x->WU, -1);
    }
    return s;
}

// upload n vectors (host [n][dim]) into slots [first, first+n), zero padded to ld, and their norms
static int upload_vectors(mn_index *x, int first, const float *vecs, int n, bool src_on_device = false);

// the index's pinned host block (staging of one insert, mailbox of a small search), at least `need` bytes; nullptr on failure
static unsigned char *pin_reserve(mn_index *x, size_t need) {
    if (need <= x->pin_cap)
        return x->pin;
    if (hipStreamSynchronize(x->stream) != hipSuccess) // (copies out of the old block may still be queued)
        return nullptr;
    if (x->pin)
        (void)hipHostFree(x->pin);
    x->pin = nullptr;
    x->pin_cap = 0;
    void *p = nullptr;
    if (hipHostMalloc(&p, need * 2, hipHostMallocDefault) != hipSuccess)
        return nullptr;
    x->pin = (unsigned char *)p;
    x->pin_cap = need * 2;
    return x->pin;
}
// layout of the block during ONE insert (every piece is consumed before the insert's single synchronisation returns)
#define MN_PIN_META 0      /* up_off, level, deleted, id of the new slot */
#define MN_PIN_IN 64       /* slot, entry slot, max level */
#define MN_PIN_OUT 96      /* state back (2 ints), counters (4 u64) */
#define MN_PIN_VEC 256     /* the vector, zero padded to ld */
static size_t pin_log_off(const mn_index *x) { return (MN_PIN_VEC + (size_t)x->ld * sizeof(float) + 63) & ~(size_t)63; }
static size_t pin_insert_bytes(const mn_index *x) {
    return pin_log_off(x) + ((size_t)1 + (size_t)MN_CHLOG_CAP * MN_CHLOG_INTS) * sizeof(int);
}

// make device buffers large enough for the host tables and upload metadata of new slots
static int sync_meta(mn_index *x) {
    hipStream_t st = x->stream;
    size_t ns = (size_t)x->n_slots;
    size_t hs = std::max(ns, (size_t)x->slot_hint); // a bulk build has announced its final slot count
    if (x->d_vectors.reserve(ns * x->ld, true, st, -1, hs * x->ld)) return -1;
    if (x->d_norms.reserve(ns, true, st, -1, hs)) return -1;
    if (x->d_links0.reserve(ns * x->W0, true, st, 0xFF, hs * x->W0)) return -1;
    if (x->d_links_up.reserve((size_t)std::max(1, x->n_pool_rows) * x->WU, true, st, 0xFF)) return -1;
    if (x->d_up_off.reserve(ns, true, st, -1, hs)) return -1;
    if (x->d_levels.reserve(ns, true, st, -1, hs)) return -1;
    if (x->d_deleted.reserve(ns, true, st, -1, hs)) return -1;
    if (x->d_dirty.reserve(ns, true, st, 0, hs)) return -1;
    if (x->d_ids.reserve(ns, true, st, -1, hs)) return -1;
    int a = x->meta_uploaded, n = x->n_slots - a;
    unsigned char *pin = n == 1 ? pin_reserve(x, pin_insert_bytes(x)) : nullptr;
    if (n == 1 && pin) { // one insert: staged in the pinned block — four truly asynchronous copies, nothing to wait for
        HIPCHK(hipStreamSynchronize(st)); // (whoever starts writing into the block waits for its previous user; idle as a rule)
        int *pi = reinterpret_cast<int *>(pin + MN_PIN_META);
        pi[0] = x->up_off[a];
        reinterpret_cast<unsigned char *>(pi + 1)[0] = (unsigned char)x->levels[a];
        reinterpret_cast<unsigned char *>(pi + 2)[0] = x->deleted[a];
        memcpy(pi + 4, &x->ids[a], sizeof(long long));
        HIPCHK(hipMemcpyAsync(x->d_up_off.p + a, pi, sizeof(int), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(x->d_levels.p + a, pi + 1, 1, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(x->d_deleted.p + a, pi + 2, 1, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(x->d_ids.p + a, pi + 4, sizeof(long long), hipMemcpyHostToDevice, st));
        x->meta_uploaded = x->n_slots;
    } else if (n > 0) {
        HIPCHK(hipMemcpyAsync(x->d_up_off.p + a, x->up_off.data() + a, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(x->d_levels.p + a, x->levels.data() + a, (size_t)n, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(x->d_deleted.p + a, x->deleted.data() + a, (size_t)n, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(x->d_ids.p + a, x->ids.data() + a, (size_t)n * sizeof(long long), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st)); // host vectors may be reallocated by later appends
        x->meta_uploaded = x->n_slots;
    }
    if (!x->load_vecs.empty()) { // vectors of loaded nodes (mn_hnsw_load_node): one upload for the lot
        std::vector<float> v;
        v.swap(x->load_vecs);
        const int first = x->load_first;
        x->load_first = -1;
        if (upload_vectors(x, first, v.data(), (int)(v.size() / (size_t)x->dim)))
            return -1;
    }
    return 0;
}

static int upload_vectors(mn_index *x, int first, const float *vecs, int n, bool src_on_device) {
    hipStream_t st = x->stream;
    // (src_on_device: the rows are already in HBM — mn_hnsw_build_dev — and never visit the host)
    const hipMemcpyKind kind = src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (n == 1 && !src_on_device && x->pin && x->pin_cap >= pin_insert_bytes(x)) { // (sync_meta has just sized the block)
        float *pv = reinterpret_cast<float *>(x->pin + MN_PIN_VEC);
        memcpy(pv, vecs, (size_t)x->dim * sizeof(float));
        for (int i = x->dim; i < x->ld; i++)
            pv[i] = 0.0f;
        HIPCHK(hipMemcpyAsync(x->d_vectors.p + (size_t)first * x->ld, pv, (size_t)x->ld * sizeof(float), hipMemcpyHostToDevice, st));
        if (x->metric == MN_METRIC_COSINE)
            mn_launch_norms(dev_view(x), first, 1, x->d_norms.p, st);
        return 0; // the insert's own synchronisation covers the copy
    }
    if (x->ld == x->dim) {
        HIPCHK(hipMemcpyAsync(x->d_vectors.p + (size_t)first * x->ld, vecs, (size_t)n * x->dim * sizeof(float), kind, st));
    } else {
        HIPCHK(hipMemsetAsync(x->d_vectors.p + (size_t)first * x->ld, 0, (size_t)n * x->ld * sizeof(float), st));
        HIPCHK(hipMemcpy2DAsync(x->d_vectors.p + (size_t)first * x->ld, (size_t)x->ld * sizeof(float), vecs,
                                (size_t)x->dim * sizeof(float), (size_t)x->dim * sizeof(float), (size_t)n, kind, st));
    }
    if (x->metric == MN_METRIC_COSINE)
        mn_launch_norms(dev_view(x), first, n, x->d_norms.p, st);
    HIPCHK(hipStreamSynchronize(st));
    return 0;
}

static size_t d_row_off(const mn_index *x, int slot, int level) { // element offset of a row inside d_links0 / d_links_up
    return level == 0 ? (size_t)slot * x->W0 : ((size_t)x->up_off[slot] + (level - 1)) * x->WU;
}

static int pull_links(mn_index *x) {
    if (x->host_links_valid)
        return 0;
    x->h_links0.assign((size_t)x->n_slots * x->W0, -1);
    x->h_links_up.assign((size_t)x->n_pool_rows * x->WU, -1);
    const size_t n0 = std::min(x->h_links0.size(), x->d_links0.cap), nu = std::min(x->h_links_up.size(), x->d_links_up.cap);
    if (n0)
        HIPCHK(hipMemcpyAsync(x->h_links0.data(), x->d_links0.p, n0 * sizeof(int), hipMemcpyDeviceToHost, x->stream));
    if (nu)
        HIPCHK(hipMemcpyAsync(x->h_links_up.data(), x->d_links_up.p, nu * sizeof(int), hipMemcpyDeviceToHost, x->stream));
    HIPCHK(hipStreamSynchronize(x->stream));
    x->host_links_valid = true;
    x->dev_links_stale = false;
    x->dev_links_all = false;
    x->h_dirty_rows.clear();
    return 0;
}

static int *h_row(mn_index *x, int slot, int level, int *W);

static int push_links(mn_index *x) {
    if (!x->dev_links_stale)
        return 0;
    if (sync_meta(x))
        return -1;
    if (x->dev_links_all) {
        if (x->n_slots)
            HIPCHK(hipMemcpyAsync(x->d_links0.p, x->h_links0.data(), x->h_links0.size() * sizeof(int), hipMemcpyHostToDevice,
                                  x->stream));
        if (x->n_pool_rows)
            HIPCHK(hipMemcpyAsync(x->d_links_up.p, x->h_links_up.data(), x->h_links_up.size() * sizeof(int),
                                  hipMemcpyHostToDevice, x->stream));
    } else {
        for (const auto &r : x->h_dirty_rows) { // a delete edits a few dozen rows: send those, not the table
            int W;
            const int *row = h_row(x, r.first, r.second, &W);
            int *dst = (r.second == 0 ? x->d_links0.p : x->d_links_up.p) + d_row_off(x, r.first, r.second);
            HIPCHK(hipMemcpyAsync(dst, row, (size_t)W * sizeof(int), hipMemcpyHostToDevice, x->stream));
        }
    }
    HIPCHK(hipStreamSynchronize(x->stream));
    x->dev_links_stale = false;
    x->dev_links_all = false;
    x->h_dirty_rows.clear();
    return 0;
}

static int *h_row(mn_index *x, int slot, int level, int *W) {
    if (level == 0) {
        *W = x->W0;
        return x->h_links0.data() + (size_t)slot * x->W0;
    }
    *W = x->WU;
    return x->h_links_up.data() + ((size_t)x->up_off[slot] + (level - 1)) * x->WU;
}

// Give the rows of one kind (layer 0, or the layers above) a longer stride.  Needs the full mirror; the device copy is
// dropped and rebuilt from it at the next push (fresh buffers, so that rows of future slots read as empty).
static int restride(mn_index *x, bool layer0, int newW) {
    if (pull_links(x))
        return -1;
    std::vector<int> &h = layer0 ? x->h_links0 : x->h_links_up;
    const int oldW = layer0 ? x->W0 : x->WU;
    const size_t rows = layer0 ? (size_t)x->n_slots : (size_t)x->n_pool_rows;
    std::vector<int> nh(rows * (size_t)newW, -1);
    for (size_t r = 0; r < rows; r++)
        memcpy(nh.data() + r * newW, h.data() + r * oldW, (size_t)oldW * sizeof(int));
    h.swap(nh);
    (layer0 ? x->W0 : x->WU) = newW;
    (void)hipStreamSynchronize(x->stream);
    (layer0 ? x->d_links0 : x->d_links_up).release();
    x->dev_links_stale = true;
    x->dev_links_all = true;
    return 0;
}

static int h_row_count(const int *row, int W) {
    int n = 0;
    while (n < W && row[n] >= 0)
        n++;
    return n;
}

// ───────────────────────── library ─────────────────────────

extern "C" int mn_abi_version(void) { return MN_ABI_VERSION; }
extern "C" const char *mn_last_error(void) { return g_err.c_str(); }

extern "C" int mn_device_count(void) try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0)
            ok++;
    }
    return ok;
} MN_GUARD_END(set_err, MN_NOTHING, -1)

extern "C" int mn_vec_parse_metric(const char *name, int *out) { // src/vec_math.c:192-204
    if (!name || !out)
        return -1;
    if (strcmp(name, "l2") == 0) {
        *out = MN_METRIC_L2;
        return 0;
    }
    if (strcmp(name, "cosine") == 0) {
        *out = MN_METRIC_COSINE;
        return 0;
    }
    if (strcmp(name, "inner_product") == 0) {
        *out = MN_METRIC_INNER_PRODUCT;
        return 0;
    }
    return -1;
}

extern "C" int mn_vec_dist_batch(int metric, int order, const float *query, const float *rows, int64_t n, int dim,
                                 float *out) try {
    if (mn_device_count() <= 0) {
        set_err("mn_vec_dist_batch: no gfx950 device");
        return -1;
    }
    if (n <= 0)
        return 0;
    int ld = (dim + 3) & ~3;
    DevBuf<float> dq, dr, dout; // released on every path
    struct Rel { DevBuf<float> &a, &b, &c; ~Rel() { a.release(); b.release(); c.release(); } } rel{dq, dr, dout};
    if (dq.reserve((size_t)dim, false, nullptr) || dr.reserve((size_t)n * ld, false, nullptr) || dout.reserve((size_t)n, false, nullptr))
        return -1;
    HIPCHK(hipMemcpy(dq.p, query, (size_t)dim * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dr.p, 0, (size_t)n * ld * sizeof(float)));
    HIPCHK(hipMemcpy2D(dr.p, (size_t)ld * sizeof(float), rows, (size_t)dim * sizeof(float), (size_t)dim * sizeof(float),
                       (size_t)n, hipMemcpyHostToDevice));
    mn_launch_dist_batch(metric, order, dq.p, dr.p, n, dim, ld, dout.p, nullptr);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, dout.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
} MN_GUARD_END(set_err, MN_NOTHING, -1)

// ───────────────────────── create / destroy ─────────────────────────

extern "C" mn_index *mn_hnsw_create_on(int dim, int metric, int M, int ef_construction, int device) try {
    if (dim <= 0 || M < 2 || ef_construction < 1 || metric < 0 || metric > 2) {
        set_err("mn_hnsw_create: bad parameters");
        return nullptr;
    }
    if (M > MN_MAX_M) { // the reference has no upper bound; here a list (2M + 1 entries, three LDS arrays) must fit a workgroup's LDS
        set_err("mn_hnsw_create: M=%d exceeds this build's limit of %d", M, MN_MAX_M);
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_err("mn_hnsw_create: HIP device %d not available (no CPU fallback)", device);
        return nullptr;
    }
    mn_index *x = new mn_index();
    x->device = device;
    x->dim = dim;
    x->ld = (dim + 3) & ~3;
    x->metric = metric;
    x->M = M;
    x->M_max0 = 2 * M; // src/hnsw_algo.c:188
    x->W0 = x->M_max0;
    x->WU = M;
    x->efc = ef_construction;
    x->level_mult = 1.0 / log((double)M); // :192
    x->ht.assign(256, -1);                // :197
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&x->ev0) != hipSuccess || hipEventCreate(&x->ev1) != hipSuccess || hipEventCreate(&x->ev2) != hipSuccess) {
        set_err("mn_hnsw_create: cannot create HIP stream/events on device %d", device);
        delete x;
        return nullptr;
    }
    { // once per process and device: the kernels' code objects are loaded now, not under the first query (MN_LAZY_MODULES=1: as before)
        static std::mutex mu;
        static unsigned long long touched = 0;
        std::lock_guard<std::mutex> lk(mu);
        if (device < 64 && !(touched >> device & 1ull) && !getenv("MN_LAZY_MODULES")) {
            touched |= 1ull << device;
            mn_module_touch_kernels();
            mn_module_touch_seq();
            mn_module_touch_spec();
            mn_module_touch_build();
        }
    }
    return x;
} MN_GUARD_END(set_err, MN_NOTHING, nullptr)

extern "C" mn_index *mn_hnsw_create(int dim, int metric, int M, int ef_construction) try {
    return mn_hnsw_create_on(dim, metric, M, ef_construction, 0);
} MN_GUARD_END(set_err, MN_NOTHING, nullptr)

extern "C" void mn_hnsw_destroy(mn_index *x) {
    if (!x)
        return;
    (void)hipSetDevice(x->device);
    if (x->stream)
        (void)hipStreamSynchronize(x->stream);
    x->d_vectors.release(); x->d_norms.release(); x->d_links0.release(); x->d_links_up.release();
    x->d_up_off.release(); x->d_levels.release(); x->d_deleted.release(); x->d_dirty.release(); x->d_ids.release();
    x->ws_bm0.release(); x->ws_bmu.release(); x->ws_cand.release(); x->ws_res.release(); x->ws_q.release();
    x->ws_outd.release(); x->ws_outi.release(); x->ws_outc.release(); x->ws_qslots.release(); x->ws_sel.release();
    x->ws_nsel.release(); x->ws_upidx.release(); x->lk_target.release(); x->lk_src.release(); x->lk_counters.release();
    x->lk_count.release(); x->lk_fill.release(); x->lk_binoff.release(); x->lk_touched.release(); x->lk_bins.release();
    x->lk_newrows.release(); x->ws_counters.release(); x->ws_state.release();
    x->lk_rec.release(); x->lk_rec_all.release(); x->lk_cls.release();
    x->er_slot.release(); x->er_level.release(); x->er_nbr.release(); x->er_dist.release();
    x->d_staged.release();
    x->sh_sel.release(); x->sh_nsel.release(); x->sh_gcnt.release(); x->sh_lcnt.release(); x->sh_ovf.release(); x->ws_chlog.release(); x->sh_gids.release();
    if (x->pin) { (void)hipHostFree(x->pin); x->pin = nullptr; x->pin_cap = 0; }
    x->sh_lids.release(); x->sh_gd.release(); x->sh_ld.release();
    x->ws_readlog.release(); x->ws_nread.release(); x->ws_ncommit.release(); x->d_stamp0.release(); x->d_stampU.release();
    x->d_sidx0.release(); x->d_sidxU.release(); x->d_saved_rows.release();
    x->d_pre_act.release(); x->d_pre_cnt.release(); x->d_pre_row.release(); x->d_spec_why.release();
    if (x->ev0) (void)hipEventDestroy(x->ev0);
    if (x->ev1) (void)hipEventDestroy(x->ev1);
    if (x->ev2) (void)hipEventDestroy(x->ev2);
    if (x->stream) (void)hipStreamDestroy(x->stream);
    delete x;
}

extern "C" void mn_hnsw_seed_rng(mn_index *x, unsigned seed) { x->rng_state = seed ? seed : 1; }

extern "C" int mn_hnsw_set_order(mn_index *x, int order) try {
    if (x->n_slots > 0) {
        set_err("mn_hnsw_set_order: index not empty");
        return -1;
    }
    if (order != MN_ORDER_SSE && order != MN_ORDER_WAVE)
        return -1;
    x->order = order;
    return 0;
} MN_GUARD_END(set_err, if (x) x->broken = true, -1)

// ───────────────────────── search ─────────────────────────

// allocations of a search launch over nq queries (the only step of it that can fail for lack of memory)
static int reserve_search_ws(mn_index *x, int64_t nq, int ef) {
    hipStream_t st = x->stream;
    const int64_t bm0_words = ((int64_t)x->n_slots + 31) / 32;
    if (x->ws_bm0.reserve((size_t)nq * bm0_words, false, st)) return -1;
    if (x->ws_cand.reserve((size_t)nq * (16 * ef + 1024), false, st)) return -1;
    if (x->ws_res.reserve((size_t)nq * (ef > MN_RES_LDS ? ef : 8), false, st)) return -1;
    if (x->ws_counters.reserve(4, false, st)) return -1;
    return 0;
}

static int prepare_search_ws(mn_index *x, int64_t nq, int ef, MnSearchArgs &a, bool zero_counters = true,
                             bool lds_bitmap_ok = false) {
    hipStream_t st = x->stream;
    a.bm0_words = ((int64_t)x->n_slots + 31) / 32;
    a.cand_gcap = 16 * ef + 1024;
    a.res_gcap = ef > MN_RES_LDS ? ef : 8;
    if (reserve_search_ws(x, nq, ef))
        return -1;
    // few queries on a small index (the SQL surface: one query per xFilter): k_beam_coop keeps the visited bitmap in LDS
    a.lds_bitmap = lds_bitmap_ok && nq <= 128 &&
                   (long long)mn_search_lds_bytes(x->ld, true) + 1024 + a.bm0_words * (long long)sizeof(unsigned) <= 64 * 1024 &&
                   !(getenv("MN_COOP") && atoi(getenv("MN_COOP")) == 0) && !(getenv("MN_LDS_BITMAP") && atoi(getenv("MN_LDS_BITMAP")) == 0);
    if (!a.lds_bitmap)
        HIPCHK(hipMemsetAsync(x->ws_bm0.p, 0, (size_t)nq * a.bm0_words * sizeof(unsigned), st));
    if (zero_counters)
        HIPCHK(hipMemsetAsync(x->ws_counters.p, 0, 4 * sizeof(unsigned long long), st));
    a.bitmap0 = x->ws_bm0.p;
    a.cand_ovf = x->ws_cand.p;
    a.res_ovf = x->ws_res.p;
    a.counters = x->ws_counters.p;
    a.bitmap_up = nullptr;
    a.bmu_words = 0;
    {
        // the latency kernels request the rows of ALL listed neighbours next to the visited probe only when the vectors do not
        // fit the 256 MB Infinity Cache (a row is then an HBM round trip worth hiding: 1M x 768, one query 1.575 -> 1.52 ms);
        // on a cache-resident index the extra rows cost more than the probe (10k x 768: 0.58 -> 0.61 ms).  MN_SPEC_ROWS=0 / 1 forces.
        const char *sr = getenv("MN_SPEC_ROWS");
        const bool big = (size_t)x->n_slots * x->ld * sizeof(float) > ((size_t)256 << 20);
        a.no_spec_rows = sr ? (atoi(sr) == 0 ? 1 : 0) : (big ? 0 : 1);
#ifdef MN_SSE_TILE_PATH
        const char *e = getenv("MN_SSE_TILE"); // tuning knob of the opt-in tiled SSE path
        a.use_tile = e ? atoi(e) : 1;
#else
        a.use_tile = 0;
#endif
    }
    return 0;
}

static int fetch_counters(mn_index *x) {
    unsigned long long c[4];
    if (!x->ws_counters.p) // nothing launched yet
        return 0;
    HIPCHK(hipMemcpyAsync(c, x->ws_counters.p, sizeof(c), hipMemcpyDeviceToHost, x->stream));
    HIPCHK(hipStreamSynchronize(x->stream));
    float ms = 0;
    if (hipEventElapsedTime(&ms, x->ev0, x->ev1) == hipSuccess)
        x->last.last_kernel_ms = ms;
    x->last.last_n_dist = (int64_t)c[0];
    x->last.last_n_expanded = (int64_t)c[1];
    x->last.last_n_overflow = (int64_t)c[2];
    return 0;
}

// q_counters: see MnSearchArgs (the few-queries path: the stream then carries the search kernel and nothing else — no counter
// memset, no event records, no copy back)
static int search_batch_dev_impl(mn_index *x, const float *d_queries, int64_t nq, int k, int ef, int64_t *d_ids, float *d_dists,
                                 int *d_counts, unsigned long long *q_counters) {
    if (use_device(x))
        return -1;
    if (nq <= 0)
        return 0;
    if (k <= 0) {
        set_err("mn_hnsw_search: k must be > 0");
        return -1;
    }
    if (ef < k) // src/hnsw_algo.c:673
        ef = k;
    hipStream_t st = x->stream;
    if (x->entry_id == -1 || x->node_count == 0) { // :671
        HIPCHK(hipMemsetAsync(d_counts, 0, (size_t)nq * sizeof(int), st));
        HIPCHK(hipMemsetAsync(d_ids, 0xFF, (size_t)nq * k * sizeof(int64_t), st));
        HIPCHK(hipMemsetAsync(d_dists, 0, (size_t)nq * k * sizeof(float), st));
        return 0;
    }
    if (push_links(x) || sync_meta(x))
        return -1;
    // per-query workspace (visited bitmap + heap spill) is bounded to ~8 GiB: larger batches run as chunks
    const size_t per_q = (size_t)(((int64_t)x->n_slots + 31) / 32) * 4 + (size_t)(16 * ef + 1024) * 8 +
                         (size_t)(ef > MN_RES_LDS ? ef : 8) * 8;
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(nq, (int64_t)((8ull << 30) / per_q)));
    const int entry_slot = ht_find(x, x->entry_id);
    MnDevIndex v = dev_view(x);
    // every allocation happens here, for the largest chunk: nothing between the two event records can fail half-way
    if (reserve_search_ws(x, chunk, ef))
        return -1;
    if (!q_counters) {
        HIPCHK(hipEventRecord(x->ev0, st));
        x->last_on_host = false;
    }
    for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
        const int64_t m = std::min<int64_t>(chunk, nq - q0);
        MnSearchArgs a;
        memset(&a, 0, sizeof(a));
        if (prepare_search_ws(x, m, ef, a, q0 == 0 && !q_counters, true)) { // (sized above: memsets only)
            if (!q_counters)
                (void)hipEventRecord(x->ev1, st);
            return -1;
        }
        a.q_counters = q_counters ? q_counters + (size_t)q0 * 4 : nullptr;
        a.queries = d_queries + (size_t)q0 * x->dim;
        a.nq = m;
        a.k = k;
        a.ef = ef;
        a.entry_slot = entry_slot;
        a.max_level = x->max_level;
        a.out_ids = (long long *)d_ids + (size_t)q0 * k;
        a.out_dists = d_dists + (size_t)q0 * k;
        a.out_counts = d_counts + q0;
        mn_launch_search(v, a, false, st);
    }
    if (!q_counters)
        HIPCHK(hipEventRecord(x->ev1, st));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int mn_hnsw_search_batch_dev(mn_index *x, const float *d_queries, int64_t nq, int k, int ef, int64_t *d_ids,
                                        float *d_dists, int *d_counts) try {
    return search_batch_dev_impl(x, d_queries, nq, k, ef, d_ids, d_dists, d_counts, nullptr);
} MN_GUARD_END(set_err, MN_NOTHING, -1)

// A sharded search all-gathers every shard's overflow count next to its top-k (mn_hnsw_search_sharded_dev): every rank sees
// the same counts, so every rank fails alike — and names the shard — the first time its host looks (any synchronising call).
static int check_sharded_overflow(mn_index *x) {
    if (!x->sh_ovf_pending)
        return 0;
    const int world = x->sh_ovf_pending;
    x->sh_ovf_pending = 0;
    std::vector<unsigned long long> h((size_t)world);
    HIPCHK(hipMemcpyAsync(h.data(), x->sh_ovf.p, (size_t)world * sizeof(unsigned long long), hipMemcpyDeviceToHost, x->stream));
    HIPCHK(hipStreamSynchronize(x->stream));
    for (int r = 0; r < world; r++)
        if (h[(size_t)r]) {
            set_err("mn_hnsw_search_sharded: shard %d: %llu queries exceeded heap workspace (its top-k lists are truncated)", r,
                    h[(size_t)r]);
            return -1;
        }
    return 0;
}

// for mn_shards.hip (one process, several GPUs): synchronises the index's stream and reports an overflow of its last search
int mn_index_search_overflow(mn_index *x, long long *n_overflow) {
    *n_overflow = 0;
    if (x->entry_id == -1 || x->node_count == 0)
        return 0;
    if (fetch_counters(x))
        return -1;
    *n_overflow = (long long)x->last.last_n_overflow;
    return 0;
}

extern "C" int mn_hnsw_sync(mn_index *x) try {
    if (use_device(x))
        return -1;
    HIPCHK(hipStreamSynchronize(x->stream));
    return check_sharded_overflow(x);
} MN_GUARD_END(set_err, MN_NOTHING, -1)

static int counters_from(mn_index *x, const unsigned long long *c) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, x->ev0, x->ev1) == hipSuccess)
        x->last.last_kernel_ms = ms;
    x->last.last_n_dist = (int64_t)c[0];
    x->last.last_n_expanded = (int64_t)c[1];
    x->last.last_n_overflow = (int64_t)c[2];
    return 0;
}

// A handful of queries from host memory: the kernel reads them from, and writes ids / distances / counts to, one pinned host
// block; the counters follow in the same block; ONE synchronisation.
static int search_small(mn_index *x, const float *queries, int64_t nq, int k, int ef, int64_t *out_ids, float *out_dists,
                        int *out_counts) {
    hipStream_t st = x->stream;
    const size_t qb = (size_t)nq * x->dim * sizeof(float), ib = (size_t)nq * k * sizeof(int64_t), db = (size_t)nq * k * sizeof(float);
    const size_t cb = ((size_t)nq * sizeof(int) + 7) & ~(size_t)7;
    const size_t o_ids = (qb + 15) & ~(size_t)15, o_d = o_ids + ib, o_c = (o_d + db + 7) & ~(size_t)7, o_cnt = o_c + cb;
    const size_t need = o_cnt + (size_t)nq * 4 * sizeof(unsigned long long);
    if (push_links(x) || sync_meta(x)) // (before the block is written: a pending single-slot upload stages through it too)
        return -1;
    if (!pin_reserve(x, need)) {
        set_err("mn_hnsw_search: cannot allocate %zu bytes of pinned host memory", need);
        return -1;
    }
    HIPCHK(hipStreamSynchronize(st));
    memcpy(x->pin, queries, qb);
    unsigned long long *qc = reinterpret_cast<unsigned long long *>(x->pin + o_cnt);
    if (search_batch_dev_impl(x, (const float *)x->pin, nq, k, ef, (int64_t *)(x->pin + o_ids), (float *)(x->pin + o_d),
                              (int *)(x->pin + o_c), qc))
        return -1;
    const bool launched = x->entry_id != -1 && x->node_count > 0;
    HIPCHK(hipStreamSynchronize(st)); // the one wait of a lone query: the kernel has answered into the block, counters included
    memcpy(out_ids, x->pin + o_ids, ib);
    memcpy(out_dists, x->pin + o_d, db);
    memcpy(out_counts, x->pin + o_c, (size_t)nq * sizeof(int));
    if (launched) {
        unsigned long long tot[4] = {0, 0, 0, 0};
        for (int64_t q = 0; q < nq; q++)
            for (int i = 0; i < 3; i++)
                tot[i] += qc[q * 4 + i];
        x->last.last_n_dist = (int64_t)tot[0];
        x->last.last_n_expanded = (int64_t)tot[1];
        x->last.last_n_overflow = (int64_t)tot[2];
        x->last.last_kernel_ms = 0.0f; // (not timed: no event records on this path)
        x->last_on_host = true;
        if (x->last.last_n_overflow) {
            set_err("mn_hnsw_search: %lld queries exceeded heap workspace", (long long)x->last.last_n_overflow);
            return -1;
        }
    }
    return 0;
}

extern "C" int mn_hnsw_search_batch(mn_index *x, const float *queries, int64_t nq, int k, int ef, int64_t *out_ids,
                                    float *out_dists, int *out_counts) try {
    if (use_device(x))
        return -1;
    if (nq <= 0)
        return 0;
    if (k > 0 && (size_t)nq * ((size_t)x->dim * 4 + (size_t)k * 12 + 4) <= ((size_t)1 << 20))
        return search_small(x, queries, nq, k, ef, out_ids, out_dists, out_counts);
    hipStream_t st = x->stream;
    if (x->ws_q.reserve((size_t)nq * x->dim, false, st)) return -1;
    if (x->ws_outi.reserve((size_t)nq * k, false, st)) return -1;
    if (x->ws_outd.reserve((size_t)nq * k, false, st)) return -1;
    if (x->ws_outc.reserve((size_t)nq, false, st)) return -1;
    HIPCHK(hipMemcpyAsync(x->ws_q.p, queries, (size_t)nq * x->dim * sizeof(float), hipMemcpyHostToDevice, st));
    if (mn_hnsw_search_batch_dev(x, x->ws_q.p, nq, k, ef, (int64_t *)x->ws_outi.p, x->ws_outd.p, x->ws_outc.p))
        return -1;
    HIPCHK(hipMemcpyAsync(out_ids, x->ws_outi.p, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out_dists, x->ws_outd.p, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out_counts, x->ws_outc.p, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (x->entry_id != -1 && x->node_count > 0) {
        if (fetch_counters(x))
            return -1;
        if (x->last.last_n_overflow) {
            set_err("mn_hnsw_search: %lld queries exceeded heap workspace", (long long)x->last.last_n_overflow);
            return -1;
        }
    }
    return 0;
} MN_GUARD_END(set_err, MN_NOTHING, -1)

extern "C" int mn_hnsw_search(mn_index *x, const float *query, int k, int ef, mn_search_result *results) try {
    if (k <= 0)
        return 0;
    std::vector<int64_t> ids((size_t)k);
    std::vector<float> ds((size_t)k);
    int cnt = 0;
    if (mn_hnsw_search_batch(x, query, 1, k, ef, ids.data(), ds.data(), &cnt))
        return 0; // src/hnsw_algo.h:70: count, 0 on failure
    for (int i = 0; i < cnt; i++) {
        results[i].id = ids[i];
        results[i].distance = ds[i];
    }
    return cnt;
} MN_GUARD_END(set_err, MN_NOTHING, 0)

// ───────────────────────── insert ─────────────────────────

// searched + linked against the graph frozen at call start; slots[] are already appended & uploaded
// search half of a group of inserts against the graph as it stands (k_beam<BUILD>): per node and layer the first
// min(found, M_max) results → ws_sel / ws_nsel.  log_cap > 0 also records the link rows every search read.
static int build_search(mn_index *x, const int *slots, int nq, MnSearchArgs &a, int log_cap) {
    hipStream_t st = x->stream;
    const int fz_max = x->max_level;
    const int nlev = fz_max + 1;
    memset(&a, 0, sizeof(a));
    if (prepare_search_ws(x, nq, x->efc, a))
        return -1;
    // upper-layer bitmaps for batch nodes with level >= 1
    std::vector<int> upidx((size_t)nq, -1);
    int n_upper = 0;
    for (int j = 0; j < nq; j++)
        if (x->levels[slots[j]] >= 1 && fz_max >= 1)
            upidx[j] = n_upper++;
    a.bmu_words = ((int64_t)std::max(1, x->n_pool_rows) + 31) / 32;
    if (n_upper) {
        size_t words = (size_t)n_upper * fz_max * a.bmu_words;
        if (x->ws_bmu.reserve(words, false, st)) return -1;
        HIPCHK(hipMemsetAsync(x->ws_bmu.p, 0, words * sizeof(unsigned), st));
    }
    a.bitmap_up = x->ws_bmu.p;
    if (x->ws_qslots.reserve((size_t)nq, false, st)) return -1;
    if (x->ws_upidx.reserve((size_t)nq, false, st)) return -1;
    if (x->ws_sel.reserve((size_t)nq * nlev * x->M_max0, false, st)) return -1;
    if (x->ws_nsel.reserve((size_t)nq * nlev, false, st)) return -1;
    HIPCHK(hipMemcpyAsync(x->ws_qslots.p, slots, (size_t)nq * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(x->ws_upidx.p, upidx.data(), (size_t)nq * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(x->ws_nsel.p, 0, (size_t)nq * nlev * sizeof(int), st));
    a.query_slots = x->ws_qslots.p;
    a.up_bm_index = x->ws_upidx.p;
    a.nq = nq;
    a.ef = x->efc;
    a.entry_slot = ht_find(x, x->entry_id);
    a.max_level = fz_max;
    a.sel = x->ws_sel.p;
    a.nsel = x->ws_nsel.p;
    a.nlev = nlev;
    if (log_cap > 0) {
        if (x->ws_readlog.reserve((size_t)nq * log_cap * MN_RLOG_INTS, false, st)) return -1;
        if (x->ws_nread.reserve((size_t)nq, false, st)) return -1;
        a.readlog = x->ws_readlog.p;
        a.readcap = log_cap;
        a.nread = x->ws_nread.p;
    }
    MnDevIndex v = dev_view(x);
    HIPCHK(hipEventRecord(x->ev0, st));
    x->last_on_host = false;
    mn_launch_search(v, a, true, st);
    HIPCHK(hipEventRecord(x->ev1, st));
    return 0;
}

// link half of a group of inserts: nq batch nodes (device array of their slots) with their selected lists; then the
// entry point / top layer update in batch order (src/hnsw_algo.c:660-663)
static int link_batch(mn_index *x, const std::vector<int> &slots, const int *d_slots, int nlev, const int *d_sel,
                      const int *d_nsel, mn_comm *c = nullptr) {
    hipStream_t st = x->stream;
    const int nq = (int)slots.size();
    MnDevIndex v = dev_view(x);
    const int max_tuples = nq * x->M_max0;
    if (x->lk_target.reserve((size_t)max_tuples, false, st)) return -1;
    if (x->lk_src.reserve((size_t)max_tuples, false, st)) return -1;
    if (x->lk_counters.reserve(4, false, st)) return -1;
    if (x->lk_count.reserve((size_t)x->d_ids.cap, false, st, 0)) return -1;
    if (x->lk_fill.reserve((size_t)x->d_ids.cap, false, st)) return -1;
    if (x->lk_binoff.reserve((size_t)x->d_ids.cap, false, st)) return -1;
    if (x->lk_touched.reserve((size_t)max_tuples, false, st)) return -1;
    if (x->lk_bins.reserve((size_t)max_tuples, false, st)) return -1;
    if (x->lk_newrows.reserve((size_t)max_tuples * std::max(x->W0, x->WU), false, st)) return -1;
    MnLinkArgs la;
    memset(&la, 0, sizeof(la));
    la.nq = nq;
    la.nlev = nlev;
    la.query_slots = d_slots;
    la.sel = d_sel;
    la.nsel = d_nsel;
    la.t_target = x->lk_target.p;
    la.t_src = x->lk_src.p;
    la.counters = x->lk_counters.p;
    la.count = x->lk_count.p;
    la.fill = x->lk_fill.p;
    la.binoff = x->lk_binoff.p;
    la.touched = x->lk_touched.p;
    la.bins = x->lk_bins.p;
    la.newrows = x->lk_newrows.p;
    const int world = c ? c->world : 1, rank = c ? c->rank : 0;
    if (world > 1) {
        // Jointly built graph: the replay of the reverse edges — the bulk of the link half — is divided over the ranks by
        // target (slot mod world).  Every rank runs the cheap forward half, so it knows every class's target count; it replays
        // its own class, the finished rows travel as {target, row} records in one all-gather per layer, and every replica
        // commits all of them: the same rows as a one-GPU build, whoever computed them.
        const int recsz = 1 + std::max(x->W0, x->WU);
        if (x->lk_rec.reserve((size_t)max_tuples * recsz, false, st) || x->lk_cls.reserve((size_t)world + 1, false, st))
            return -1;
        la.world = world;
        la.rank = rank;
        la.records = x->lk_rec.p;
        la.cls_count = x->lk_cls.p;
        la.rec_count = x->lk_cls.p + world;
        std::vector<int> cls((size_t)world);
        for (int l = 0; l < nlev; l++) {
            la.level = l;
            la.M_max = l == 0 ? x->M_max0 : x->M;
            const int mt = l == 0 ? max_tuples : nq * x->M;
            mn_launch_link_first(v, la, mt, st);
            int status = hipGetLastError() != hipSuccess ||
                         hipMemcpyAsync(cls.data(), x->lk_cls.p, (size_t)world * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
                         hipStreamSynchronize(st) != hipSuccess;
            int seg = 0;
            for (int r = 0; r < world && !status; r++)
                seg = std::max(seg, cls[(size_t)r]);
            if (!status && seg > 0 && x->lk_rec_all.reserve((size_t)world * seg * recsz, false, st))
                status = 1;
            int failed = -1;
            const int ag = mn_comm_agree(c, status, st, &failed); // (a rank that cannot go on says so before the collective)
            if (ag) {
                if (ag < 0)
                    set_err("mn_hnsw_build_shared: %s", mn_comm_last_error_str());
                else
                    set_err("mn_hnsw_build_shared: rank %d failed in the link step; all ranks stop", failed);
                return -1;
            }
            if (seg > 0) {
                int *mine = x->lk_rec_all.p + (size_t)rank * seg * recsz;
                if (cls[(size_t)rank] > 0)
                    HIPCHK(hipMemcpyAsync(mine, x->lk_rec.p, (size_t)cls[(size_t)rank] * recsz * sizeof(int), hipMemcpyDeviceToDevice, st));
                if (mn_comm_allgather_dev(c, mine, x->lk_rec_all.p, (size_t)seg * recsz * sizeof(int), st)) {
                    set_err("mn_hnsw_build_shared: %s", mn_comm_last_error_str());
                    return -1;
                }
            }
            mn_launch_link_commit_records(v, la, mt, x->lk_rec_all.p, seg, st);
        }
    } else {
        for (int l = 0; l < nlev; l++) {
            la.level = l;
            la.M_max = l == 0 ? x->M_max0 : x->M;
            int mt = l == 0 ? max_tuples : nq * x->M;
            mn_launch_link(v, la, mt, st);
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(x->ev2, st));
    HIPCHK(hipStreamSynchronize(st)); // (the searches' heap workspace was checked before this step: run_batch / batch_search)
    x->host_links_valid = false;
    for (int j = 0; j < nq; j++) {
        int lv = x->levels[slots[j]];
        if (lv > x->max_level) {
            x->entry_id = x->ids[slots[j]];
            x->max_level = lv;
        }
    }
    return 0;
}

static int run_batch(mn_index *x, const std::vector<int> &slots, bool *links_touched) {
    const int nq = (int)slots.size();
    *links_touched = false;
    if (nq == 0)
        return 0;
    const int nlev = x->max_level + 1;
    MnSearchArgs a;
    if (build_search(x, slots.data(), nq, a, 0))
        return -1;
    // the search half only reads the graph: a failure up to here leaves it untouched (the caller takes the nodes back)
    if (fetch_counters(x)) // synchronises
        return -1;
    if (x->last.last_n_overflow) {
        set_err("mn_hnsw_insert: %lld searches exceeded heap workspace", (long long)x->last.last_n_overflow);
        return -1;
    }
    *links_touched = true;
    x->bstats.search_ms += x->last.last_kernel_ms;
    x->bstats.n_dist += x->last.last_n_dist;
    x->bstats.n_expanded += x->last.last_n_expanded;
    x->bstats.batches++;
    x->bstats.nodes += nq;
    if (link_batch(x, slots, x->ws_qslots.p, nlev, x->ws_sel.p, x->ws_nsel.p))
        return -1;
    float ms = 0;
    if (hipEventElapsedTime(&ms, x->ev1, x->ev2) == hipSuccess)
        x->bstats.link_ms += ms;
    return 0;
}

static int run_sequential(mn_index *x, const std::vector<int> &slots) {
    hipStream_t st = x->stream;
    const int n = (int)slots.size();
    if (n == 0)
        return 0;
    if (mn_insert_seq_lds_bytes(dev_view(x)) > mn_lds_optin_limit()) { // (beyond 64 KB the launcher asks the device for the grant)
        set_err("mn_hnsw_insert: dimension %d with neighbour lists of up to %d entries does not fit the insert kernel's LDS", x->dim,
                std::max(x->W0, x->WU));
        return -1;
    }
    MnSearchArgs a;
    memset(&a, 0, sizeof(a));
    x->last_on_host = false;
    if (prepare_search_ws(x, 1, x->efc, a))
        return -1;
    long long bmu_words = ((int64_t)std::max(1, x->n_pool_rows) + 31) / 32;
    if (x->ws_bmu.reserve((size_t)bmu_words, false, st)) return -1;
    if (x->ws_qslots.reserve((size_t)n, false, st)) return -1;
    if (x->ws_state.reserve(2, false, st)) return -1;
    int state[2] = {ht_find(x, x->entry_id), x->max_level};
    // one insert: inputs and outputs go through the pinned block (sized by sync_meta for this insert) and there is ONE wait
    unsigned char *pin = n == 1 && x->pin && x->pin_cap >= pin_insert_bytes(x) ? x->pin : nullptr;
    if (pin) {
        int *in = reinterpret_cast<int *>(pin + MN_PIN_IN);
        in[0] = slots[0];
        in[1] = state[0];
        in[2] = state[1];
        HIPCHK(hipMemcpyAsync(x->ws_qslots.p, in, sizeof(int), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(x->ws_state.p, in + 1, 2 * sizeof(int), hipMemcpyHostToDevice, st));
    } else {
        HIPCHK(hipMemcpyAsync(x->ws_qslots.p, slots.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(x->ws_state.p, state, sizeof(state), hipMemcpyHostToDevice, st));
    }
    MnDevIndex v = dev_view(x);
    const bool logged = x->want_chlog && n == 1;
    x->want_chlog = false;
    if (logged && x->ws_chlog.reserve((size_t)1 + (size_t)MN_CHLOG_CAP * MN_CHLOG_INTS, false, st))
        return -1;
    HIPCHK(hipEventRecord(x->ev0, st));
    const int chunk = 1024; // keeps any one launch to about a second
    for (int pos = 0; pos < n; pos += chunk) {
        int m = std::min(chunk, n - pos);
        mn_launch_insert_seq(v, x->ws_qslots.p + pos, m, x->efc, x->ws_state.p, a.bitmap0, a.bm0_words, x->ws_bmu.p,
                             bmu_words, a.cand_ovf, a.cand_gcap, a.res_ovf, a.res_gcap, a.counters, st,
                             logged ? x->ws_chlog.p : nullptr, MN_CHLOG_CAP);
    }
    HIPCHK(hipEventRecord(x->ev1, st));
    HIPCHK(hipGetLastError());
    if (pin) {
        int *out = reinterpret_cast<int *>(pin + MN_PIN_OUT);
        const size_t log_ints = (size_t)1 + (size_t)MN_CHLOG_CAP * MN_CHLOG_INTS;
        HIPCHK(hipMemcpyAsync(out, x->ws_state.p, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(out + 2, x->ws_counters.p, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        if (logged)
            HIPCHK(hipMemcpyAsync(pin + pin_log_off(x), x->ws_chlog.p, log_ints * sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        state[0] = out[0];
        state[1] = out[1];
        unsigned long long c[4];
        memcpy(c, out + 2, sizeof(c));
        counters_from(x, c);
        if (logged) {
            const int *lg = reinterpret_cast<const int *>(pin + pin_log_off(x));
            const int cnt = lg[0];
            x->h_chlog.assign(lg, lg + 1 + (size_t)(cnt > 0 && cnt <= MN_CHLOG_CAP ? cnt : 0) * MN_CHLOG_INTS);
        }
    } else {
        HIPCHK(hipMemcpyAsync(state, x->ws_state.p, sizeof(state), hipMemcpyDeviceToHost, st));
        if (logged) {
            x->h_chlog.assign((size_t)1 + (size_t)MN_CHLOG_CAP * MN_CHLOG_INTS, 0);
            HIPCHK(hipMemcpyAsync(x->h_chlog.data(), x->ws_chlog.p, x->h_chlog.size() * sizeof(int), hipMemcpyDeviceToHost, st));
        }
        if (fetch_counters(x))
            return -1;
    }
    if (x->last.last_n_overflow) {
        set_err("mn_hnsw_insert: heap workspace exceeded");
        return -1;
    }
    x->host_links_valid = false;
    x->entry_id = x->ids[state[0]];
    x->max_level = state[1];
    return 0;
}

// The same inserts, same order, same graph as run_sequential — by speculation (mn_spec.hip): windows of consecutive
// inserts are searched at once, then committed in order up to the first one whose search read a row that an earlier
// insert of the window rewrote; the next window starts there.
static int run_speculative(mn_index *x, const std::vector<int> &slots) {
    hipStream_t st = x->stream;
    const int n = (int)slots.size();
    const int LOG_CAP = 2048;
    if (x->d_stamp0.reserve((size_t)x->d_ids.cap, true, st, 0)) return -1;
    if (x->d_stampU.reserve((size_t)std::max(1, x->n_pool_rows) + 1, true, st, 0)) return -1;
    if (x->d_sidx0.reserve((size_t)x->d_ids.cap, false, st)) return -1; // (read only where the stamp is this window's)
    if (x->d_sidxU.reserve((size_t)std::max(1, x->n_pool_rows) + 1, false, st)) return -1;
    if (x->d_saved_rows.reserve((size_t)MN_SPEC_SAVE_CAP * 64, false, st)) return -1;
    if (x->ws_ncommit.reserve(1, false, st)) return -1;
    const bool trace = getenv("MN_SPEC_TRACE") != nullptr;
    if (trace) {
        if (x->d_spec_why.reserve(8, false, st)) return -1;
        HIPCHK(hipMemsetAsync(x->d_spec_why.p, 0, 8 * sizeof(int), st));
    }
    int window = 16, poor = 0;
    long long searched = 0, rounds = 0, plain = 0;
    double t_search = 0, t_commit = 0; // device ms (HIP events), reported under MN_SPEC_TRACE
    for (int pos = 0; pos < n;) {
        if (poor >= 8) {
            // a small index: nearly every search crosses the previous insert's rows, so windows commit one insert at a
            // time.  Run a stretch with the single-wavefront kernel, then try again (the index has grown meanwhile).
            const int m = std::min(512, n - pos);
            if (run_sequential(x, std::vector<int>(slots.begin() + pos, slots.begin() + pos + m)))
                return -1;
            pos += m;
            plain += m;
            poor = 4;
            continue;
        }
        int W = std::min(window, n - pos);
        // a node that raises the top layer becomes the entry point of everything after it (:660-663): it closes its window
        for (int k = 0; k < W; k++)
            if (x->levels[slots[pos + k]] > x->max_level) {
                W = k + 1;
                break;
            }
        MnSearchArgs a;
        if (build_search(x, slots.data() + pos, W, a, LOG_CAP))
            return -1;
        if (++x->spec_epoch == 0x7fffffff) { // epochs are compared for equality only: restart them before they wrap
            HIPCHK(hipMemsetAsync(x->d_stamp0.p, 0, x->d_stamp0.cap * sizeof(int), st));
            HIPCHK(hipMemsetAsync(x->d_stampU.p, 0, x->d_stampU.cap * sizeof(int), st));
            x->spec_epoch = 1;
        }
        MnDevIndex v = dev_view(x);
        // the link decisions of the whole window ahead of its commit (k_spec_prepare; MN_SPEC_AHEAD=0: decided inside the commit)
        const bool ahead = !(getenv("MN_SPEC_AHEAD") && atoi(getenv("MN_SPEC_AHEAD")) == 0) && v.W0 <= 64;
        if (ahead) {
            const size_t cells = (size_t)W * a.nlev * v.W0;
            if (x->d_pre_act.reserve(cells, false, st) || x->d_pre_cnt.reserve(cells, false, st) ||
                x->d_pre_row.reserve(cells * 64, false, st))
                return -1;
        }
        mn_launch_spec_commit(v, x->ws_qslots.p, W, a.nlev, x->ws_sel.p, x->ws_nsel.p, x->ws_readlog.p, LOG_CAP, x->ws_nread.p,
                              x->d_stamp0.p, x->d_stampU.p, x->d_sidx0.p, x->d_sidxU.p, x->d_saved_rows.p,
                              ahead ? x->d_pre_act.p : nullptr, x->d_pre_cnt.p, x->d_pre_row.p, trace ? x->d_spec_why.p : nullptr,
                              x->spec_epoch, x->ws_ncommit.p, st);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(x->ev2, st));
        int done = 0;
        HIPCHK(hipMemcpyAsync(&done, x->ws_ncommit.p, sizeof(int), hipMemcpyDeviceToHost, st));
        if (fetch_counters(x)) // synchronises
            return -1;
        {
            float ms = 0;
            t_search += x->last.last_kernel_ms;
            if (hipEventElapsedTime(&ms, x->ev1, x->ev2) == hipSuccess)
                t_commit += ms;
        }
        if (x->last.last_n_overflow) {
            set_err("mn_hnsw_insert: heap workspace exceeded");
            return -1;
        }
        if (done < 1 || done > W) {
            set_err("mn_hnsw_insert: speculative commit returned %d of %d", done, W);
            return -1;
        }
        for (int k = 0; k < done; k++) { // entry point / top layer in insertion order (:660-663)
            int lv = x->levels[slots[pos + k]];
            if (lv > x->max_level) {
                x->entry_id = x->ids[slots[pos + k]];
                x->max_level = lv;
            }
        }
        searched += W;
        rounds++;
        pos += done;
        poor = done <= 1 && W > 1 ? poor + 1 : 0;
        // A window's searches run side by side and a search is one latency-bound wavefront, so a wider window costs
        // bandwidth, not time; searches beyond the committed prefix are simply repeated.  Keep it a few times the prefix.
        window = done == W ? std::min(128, window * 2) : std::max(16, std::min(128, 3 * done)); // (≤ 128: one workgroup each, k_beam_coop)
    }
    x->host_links_valid = false;
    x->last_spec_searched = searched;
    if (trace) {
        int why[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIPCHK(hipMemcpy(why, x->d_spec_why.p, sizeof(why), hipMemcpyDeviceToHost));
        fprintf(stderr, "[mn_spec] windows ended by: log overflow %d, > 64 rewritten rows %d, a row with the log's defaults %d, an old list "
                        "not kept %d, a removed neighbour the search could have pushed %d, an added node it would have pushed %d, a "
                        "greedy step that would have gone elsewhere %d; committed whole %d\n", why[1], why[2], why[3], why[4], why[5],
                why[6], why[0], why[7]);
    }
    if (trace)
        fprintf(stderr, "[mn_spec] %d inserts: %lld rounds (%.1f committed per round), %lld searches, %lld by k_insert_seq; device ms "
                        "per round: search %.2f commit %.2f\n", n, rounds, rounds ? (double)(n - plain) / rounds : 0.0, searched, plain,
                rounds ? t_search / rounds : 0.0, rounds ? t_commit / rounds : 0.0);
    return 0;
}

#define MN_BUILD_STAGE_ONLY 100 // internal: add the nodes, leave search + link to mn_hnsw_batch_search / _link

static int insert_impl(mn_index *x, const int64_t *ids, const float *vectors, int64_t n, int mode, bool src_on_device = false) {
    if (use_device(x))
        return -1;
    if (n <= 0)
        return 0;
    // duplicates against the index and inside the call (src/hnsw_algo.c:522)
    {
        std::vector<int64_t> sorted(ids, ids + n);
        std::sort(sorted.begin(), sorted.end());
        for (int64_t i = 1; i < n; i++)
            if (sorted[i] == sorted[i - 1]) {
                set_err("mn_hnsw_insert: duplicate id %lld in batch", (long long)sorted[i]);
                return -1;
            }
        for (int64_t i = 0; i < n; i++)
            if (ht_find(x, ids[i]) >= 0) {
                set_err("mn_hnsw_insert: duplicate id %lld", (long long)ids[i]);
                return -1;
            }
    }
    {
        const int64_t f = first_table_full(x, n);
        if (f >= 0) {
            // the reference: random_level() is drawn, ht_insert fails, hnsw_insert returns -1 (src/hnsw_algo.c:532-540)
            if (n == 1)
                (void)random_level(x);
            set_err("mn_hnsw_insert: node table full (%d entries incl. soft-deleted nodes; the table grows on the live count only)",
                    x->ht_cap);
            return -1;
        }
    }
    if (push_links(x))
        return -1;
    const int first = x->n_slots;
    InsertUndo undo;
    undo.first = first;
    undo.node_count = x->node_count;
    undo.max_level = x->max_level;
    undo.n_pool_rows = x->n_pool_rows;
    undo.rng = x->rng_state;
    undo.entry_id = x->entry_id;
    bool links_touched = false, any_rest = false;
    try {
    std::vector<int> slots((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        if (x->node_count * 10 > x->ht_cap * 7) { // :527
            if (undo.grew_at < 0) {
                undo.grew_at = x->n_slots;
                undo.old_ht = x->ht;
                undo.old_cap = x->ht_cap;
            }
            ht_grow(x);
        }
        int level = random_level(x); // :532
        slots[i] = host_add_node(x, ids[i], level, 0);
        if (slots[i] < 0) { // (first_table_full rules this out)
            undo_insert(x, undo);
            set_err("mn_hnsw_insert: node table insert failed for id %lld", (long long)ids[i]);
            return -1;
        }
        x->node_count++;
    }
    if (sync_meta(x) || upload_vectors(x, first, vectors, (int)n, src_on_device)) {
        undo_insert(x, undo);
        return -1;
    }
    HIPCHK(hipMemsetAsync(x->d_dirty.p + first, 1, (size_t)n, x->stream)); // new nodes are always persisted
    size_t pos = 0;
    if (x->entry_id == -1) { // :544-548 first node just becomes the entry point
        x->entry_id = ids[0];
        x->max_level = x->levels[slots[0]];
        pos = 1;
        // -1 is the reference's "no entry point" marker AND a legal rowid: after inserting rowid -1 into an empty index
        // the reference still believes it is empty and the next insert becomes the entry point too, unlinked
        // (one-at-a-time semantics only; found by scripts/fuzz_parity.py)
        while (mode == MN_BUILD_SEQUENTIAL && x->entry_id == -1 && pos < (size_t)n) {
            x->entry_id = ids[pos];
            x->max_level = x->levels[slots[pos]];
            pos++;
        }
    }
    std::vector<int> rest(slots.begin() + pos, slots.end());
    if (mode == MN_BUILD_STAGE_ONLY) { // mn_hnsw_batch_stage: the caller drives search and link itself
        x->staged.swap(rest);
        return 0;
    }
    int rc;
    any_rest = !rest.empty();
    links_touched = true; // the exact modes edit rows in place from the first insert on
    if (mode == MN_BUILD_SEQUENTIAL) {
        // same result either way; speculation pays once several inserts are queued (MN_SPECULATE=0 turns it off)
        const char *sp = getenv("MN_SPECULATE");
        const bool spec_on = !(sp && atoi(sp) == 0);
        // (k_spec_commit stages ≤ 64 targets of ≤ 64 links, on rows that never outgrew M_max)
        if (spec_on && rest.size() >= 4 && x->W0 <= 64 && x->W0 == x->M_max0 && x->WU == x->M)
            rc = run_speculative(x, rest);
        else
            rc = run_sequential(x, rest);
    } else {
        rc = run_batch(x, rest, &links_touched);
    }
    if (rc != 0) {
        // "nothing inserted on -1": take the nodes back while the graph is still as it was; past that point the rows of
        // existing nodes may already name the new slots, and the index refuses further use instead of serving them
        if (!links_touched)
            undo_insert(x, undo);
        else if (!rest.empty())
            x->broken = true;
    }
    return rc;
    } catch (...) {
        // the same promise when host memory ran out half way (std::bad_alloc out of a table or a staging vector): nothing
        // inserted while the graph is as it was, an unusable index past that point; the C-ABI's barrier reports it
        if (!links_touched)
            undo_insert(x, undo);
        else if (any_rest)
            x->broken = true;
        throw;
    }
}

// The persist set of src/hnsw_vtab.c:755-768, accumulated over any number of inserts: every new node and
// every node that was a neighbour of a new node when it was linked.  Reading it clears it.
extern "C" int64_t mn_hnsw_take_dirty(mn_index *x, int64_t *ids, int64_t cap) try {
    if (use_device(x))
        return -1;
    if (x->n_slots == 0 || !x->d_dirty.p)
        return 0;
    const size_t n = std::min((size_t)x->n_slots, x->d_dirty.cap);
    std::vector<unsigned char> h(n);
    HIPCHK(hipStreamSynchronize(x->stream));
    HIPCHK(hipMemcpy(h.data(), x->d_dirty.p, n, hipMemcpyDeviceToHost));
    int64_t cnt = 0;
    for (size_t s = 0; s < n; s++)
        cnt += h[s] != 0;
    if (cnt > cap)
        return cnt; // nothing cleared: call again with room for cnt ids
    int64_t o = 0;
    for (size_t s = 0; s < n; s++)
        if (h[s]) {
            ids[o++] = x->ids[s];
            if (s < x->h_stale.size() && x->h_stale[s]) { // the caller rewrites this node whole
                x->h_stale[s] = 0;
                x->n_stale--;
            }
        }
    HIPCHK(hipMemset(x->d_dirty.p, 0, n));
    return cnt;
} MN_GUARD_END(set_err, MN_NOTHING, -1)

extern "C" int mn_hnsw_insert(mn_index *x, int64_t id, const float *vector) try {
    return insert_impl(x, &id, vector, 1, MN_BUILD_SEQUENTIAL);
} MN_GUARD_END(set_err, MN_NOTHING, -1)

extern "C" int mn_hnsw_insert_batch(mn_index *x, const int64_t *ids, const float *vectors, int64_t n, int mode) try {
    return insert_impl(x, ids, vectors, n, mode);
} MN_GUARD_END(set_err, MN_NOTHING, -1)

// hnsw_insert (src/hnsw_algo.c:520-666) that also says WHICH edges it added and removed: the caller's persistence
// (src/hnsw_vtab.c:755-776 rewrites the "{t}_edges" rows of the new node and of every neighbour — ≈ 1 100 rows per insert) can
// then touch only the ≈ 100 rows that changed.  *n_log = entries written (op 1: edge src -> dst added at `level` with
// `distance`; op 2: edge src -> dst removed), or -1 when the log cannot describe this insert (more than `cap` changes, a
// neighbour list wider than 64 links, or a touched node whose lists an earlier mn_hnsw_delete edited — the reference never
// persisted those edits, so only a whole rewrite of that node matches it): the persist set (mn_hnsw_take_dirty) is then
// still complete.  With a valid log the persist set is emptied — the caller has everything.  Same graph, same return
// convention as mn_hnsw_insert.
extern "C" int mn_hnsw_insert_logged(mn_index *x, int64_t id, const float *vector, mn_edge_change *log, int cap, int *n_log) try {
    *n_log = -1;
    if (use_device(x))
        return -1;
    const int was_slots = x->n_slots;
    x->want_chlog = true;
    x->h_chlog.clear();
    const int rc = insert_impl(x, &id, vector, 1, MN_BUILD_SEQUENTIAL);
    x->want_chlog = false;
    if (rc != 0)
        return rc;
    if (x->n_slots == was_slots + 1 && x->h_chlog.empty()) { // first node of an empty index: no search, no edges
        *n_log = 0;
    } else if (!x->h_chlog.empty() && x->h_chlog[0] >= 0 && x->h_chlog[0] <= cap) {
        const int n = x->h_chlog[0];
        bool stale = false; // touched = the new node's selected neighbours = dst of its own "added" entries
        for (int i = 0; i < n && x->n_stale > 0 && !stale; i++) {
            const int *e = x->h_chlog.data() + 1 + (size_t)i * MN_CHLOG_INTS;
            stale = e[0] == 1 && (size_t)e[3] < x->h_stale.size() && x->h_stale[e[3]];
        }
        for (int i = 0; i < n && !stale; i++) {
            const int *e = x->h_chlog.data() + 1 + (size_t)i * MN_CHLOG_INTS;
            log[i].op = e[0];
            log[i].src = x->ids[e[1]];
            log[i].level = e[2];
            log[i].dst = x->ids[e[3]];
            memcpy(&log[i].distance, &e[4], sizeof(float));
        }
        if (!stale)
            *n_log = n;
    }
    if (*n_log >= 0 && x->d_dirty.p && x->n_slots > 0) // (the log replaces the persist set for this insert)
        HIPCHK(hipMemsetAsync(x->d_dirty.p, 0, std::min((size_t)x->n_slots, x->d_dirty.cap), x->stream));
    return 0;
} MN_GUARD_END(set_err, MN_NOTHING, -1)

// The caller's edge-by-edge copy of the graph no longer matches the index — for the n given nodes, or (ids == NULL) for every
// node present now (its transaction rolled back): logged inserts report -1 for every insert that touches such a node, until
// the node has gone through the persist set (a whole rewrite).  Unknown ids are ignored.
static void mark_stale(mn_index *x, int slot) {
    if (x->h_stale.size() < (size_t)x->n_slots)
        x->h_stale.resize((size_t)x->n_slots, 0);
    if (!x->h_stale[slot]) {
        x->h_stale[slot] = 1;
        x->n_stale++;
    }
}
extern "C" int mn_hnsw_log_invalidate(mn_index *x, const int64_t *ids, int64_t n) try {
    if (!x)
        return -1;
    if (!ids) {
        x->h_stale.assign((size_t)x->n_slots, 1);
        x->n_stale = x->n_slots;
        return 0;
    }
    for (int64_t i = 0; i < n; i++) {
        const int s = ht_find(x, ids[i]);
        if (s >= 0)
            mark_stale(x, s);
    }
    return 0;
} MN_GUARD_END(set_err, MN_NOTHING, -1)

static int build_impl(mn_index *x, const int64_t *ids, const float *vectors, int64_t n, int grow_div, int max_batch,
                      bool src_on_device) {
    if (grow_div <= 0)
        grow_div = 16;
    if (max_batch <= 0)
        max_batch = 8192;
    if (!getenv("MN_BUILD_NO_RESERVE"))
        x->slot_hint = (int64_t)x->n_slots + n; // the slot-indexed device tables are allocated once, at their final size
    int64_t pos = 0;
    while (pos < n) {
        int64_t b = std::max<int64_t>(1, x->node_count / grow_div);
        b = std::min<int64_t>(b, max_batch);
        b = std::min<int64_t>(b, n - pos);
        if (insert_impl(x, ids + pos, vectors + (size_t)pos * x->dim, b, MN_BUILD_BATCHED, src_on_device))
            return -1;
        pos += b;
    }
    return 0;
}

extern "C" int mn_hnsw_build(mn_index *x, const int64_t *ids, const float *vectors, int64_t n, int grow_div,
                             int max_batch) try {
    return build_impl(x, ids, vectors, n, grow_div, max_batch, false);
} MN_GUARD_END(set_err, if (x) x->broken = true, -1)

// the same build from rows that are already in HBM on the index's device ([n][dim] f32, e.g. the embeddings a Node2Vec run
// has just normalised: mn_node2vec_train_into): same batches, same graph as mn_hnsw_build on a host copy of those rows
extern "C" int mn_hnsw_build_dev(mn_index *x, const int64_t *ids, const float *d_vectors, int64_t n, int grow_div,
                                 int max_batch) try {
    return build_impl(x, ids, d_vectors, n, grow_div, max_batch, true);
} MN_GUARD_END(set_err, if (x) x->broken = true, -1)

extern "C" int mn_hnsw_device(mn_index *x) { return x->device; }

// ── one batch of the batch-synchronous build in three steps, so that the search half can be split over several GPUs
//    that each hold a replica of the index (sqlite-muninn_amd/parallel.py build_distributed) ──
#define MN_STAGE_CHECK(x)                                                       \
    if (use_device(x))                                                          \
        return -1;

extern "C" int mn_hnsw_batch_stage(mn_index *x, const int64_t *ids, const float *vectors, int64_t n) try {
    MN_STAGE_CHECK(x)
    if (!x->staged.empty()) {
        set_err("mn_hnsw_batch_stage: the previous batch was not linked");
        return -1;
    }
    if (insert_impl(x, ids, vectors, n, MN_BUILD_STAGE_ONLY))
        return -1;
    const int m = (int)x->staged.size();
    if (m) {
        if (x->d_staged.reserve((size_t)m, false, x->stream)) return -1;
        HIPCHK(hipMemcpy(x->d_staged.p, x->staged.data(), (size_t)m * sizeof(int), hipMemcpyHostToDevice));
    }
    return m;
} MN_GUARD_END(set_err, if (x) x->broken = true, -1)

extern "C" int mn_hnsw_batch_dims(mn_index *x, int *nlev, int *row_width) try {
    *nlev = x->max_level + 1;
    *row_width = x->M_max0;
    return 0;
} MN_GUARD_END(set_err, MN_NOTHING, -1)

extern "C" int mn_hnsw_batch_search(mn_index *x, int lo, int hi, int *d_sel, int *d_nsel) try {
    MN_STAGE_CHECK(x)
    const int m = (int)x->staged.size();
    if (lo < 0 || hi > m || lo > hi) {
        set_err("mn_hnsw_batch_search: bad range [%d, %d) of %d staged nodes", lo, hi, m);
        return -1;
    }
    if (hi == lo)
        return 0;
    const int nlev = x->max_level + 1;
    MnSearchArgs a;
    if (build_search(x, x->staged.data() + lo, hi - lo, a, 0))
        return -1;
    const size_t row = (size_t)nlev * x->M_max0;
    HIPCHK(hipMemcpyAsync(d_sel + (size_t)lo * row, x->ws_sel.p, (size_t)(hi - lo) * row * sizeof(int), hipMemcpyDeviceToDevice,
                          x->stream));
    HIPCHK(hipMemcpyAsync(d_nsel + (size_t)lo * nlev, x->ws_nsel.p, (size_t)(hi - lo) * nlev * sizeof(int), hipMemcpyDeviceToDevice,
                          x->stream));
    if (fetch_counters(x)) // synchronises
        return -1;
    if (x->last.last_n_overflow) {
        set_err("mn_hnsw_batch_search: %lld searches exceeded heap workspace", (long long)x->last.last_n_overflow);
        return -1;
    }
    return 0;
} MN_GUARD_END(set_err, if (x) x->broken = true, -1)

static int batch_link_impl(mn_index *x, const int *d_sel, const int *d_nsel, mn_comm *c) {
    MN_STAGE_CHECK(x)
    std::vector<int> slots;
    slots.swap(x->staged);
    if (slots.empty())
        return 0;
    return link_batch(x, slots, x->d_staged.p, x->max_level + 1, d_sel, d_nsel, c);
}
extern "C" int mn_hnsw_batch_link(mn_index *x, const int *d_sel, const int *d_nsel) try {
    return batch_link_impl(x, d_sel, d_nsel, nullptr);
} MN_GUARD_END(set_err, if (x) x->broken = true, -1)

// ───────────────────────── multi-GPU: shared build, sharded search ─────────────────────────

extern "C" int mn_hnsw_build_shared(mn_index *x, mn_comm *c, const int64_t *ids, const float *vectors, int64_t n, int grow_div,
                                    int max_batch, int min_split) try {
    MN_STAGE_CHECK(x)
    if (grow_div <= 0)
        grow_div = 16;
    if (max_batch <= 0)
        max_batch = 8192;
    if (min_split <= 0)
        min_split = 256;
    const int world = c ? c->world : 1, rank = c ? c->rank : 0;
    hipStream_t st = x->stream;
    if (!getenv("MN_BUILD_NO_RESERVE"))
        x->slot_hint = (int64_t)x->n_slots + n;
    int64_t pos = 0;
    while (pos < n) {
        int64_t b = std::max<int64_t>(1, x->node_count / grow_div); // the batches of mn_hnsw_build
        b = std::min<int64_t>(b, max_batch);
        b = std::min<int64_t>(b, n - pos);
        // A rank whose local step fails (staging, workspace, its slice of the searches: overflow, out of memory) may not simply
        // return: its peers are in — or about to enter — the batch's all-gather and would wait for ever.  Every rank finishes
        // its local step, the ranks exchange one status word, and all of them fail together (mn_comm_agree).
        int status = 0;
        const int m = mn_hnsw_batch_stage(x, ids + pos, vectors + (size_t)pos * x->dim, b);
        if (m < 0)
            status = 1;
        if (const char *fi = getenv("MN_FAULT_INJECT")) { // test hook: "build_shared:<rank>:<first id position of the batch>"
            int fr = -1;
            long long fp = -1;
            if (sscanf(fi, "build_shared:%d:%lld", &fr, &fp) == 2 && fr == rank && fp >= pos && fp < pos + b && !status) {
                set_err("injected failure (MN_FAULT_INJECT)");
                status = 1;
            }
        }
        pos += b;
        const int nlev = x->max_level + 1, w0 = x->M_max0;
        const bool split = world > 1 && m >= min_split;
        const int per = split ? (m + world - 1) / world : std::max(m, 0);
        const int rows = split ? per * world : std::max(m, 0);
        const size_t row_sel = (size_t)nlev * w0;
        if (!status && m > 0) {
            if (x->sh_sel.reserve((size_t)rows * row_sel, false, st) || x->sh_nsel.reserve((size_t)rows * nlev, false, st) ||
                hipMemsetAsync(x->sh_sel.p, 0xFF, (size_t)rows * row_sel * sizeof(int), st) != hipSuccess ||
                hipMemsetAsync(x->sh_nsel.p, 0, (size_t)rows * nlev * sizeof(int), st) != hipSuccess) {
                status = 1;
            } else {
                const int lo = split ? std::min(m, rank * per) : 0, hi = split ? std::min(m, rank * per + per) : m;
                // from here on the batch's nodes are in the host tables but not linked: a failure must not leave a half-built
                // index in use
                if (mn_hnsw_batch_search(x, lo, hi, x->sh_sel.p, x->sh_nsel.p)) { // this rank's slice of the batch's searches
                    x->broken = true;
                    status = 1;
                }
            }
        }
        if (world > 1) {
            int failed = -1;
            const std::string mine = status ? std::string(mn_last_error()) : std::string();
            const int ag = mn_comm_agree(c, status, st, &failed);
            if (ag) {
                if (ag < 0)
                    set_err("mn_hnsw_build_shared: %s", mn_comm_last_error_str());
                else if (failed == rank)
                    set_err("mn_hnsw_build_shared: rank %d failed: %s", rank, mine.c_str());
                else
                    set_err("mn_hnsw_build_shared: rank %d failed in this batch; all ranks stop", failed);
                x->broken = true;
                return -1;
            }
        } else if (status) {
            return -1;
        }
        if (m == 0)
            continue;
        if (split) { // in place: rank r's rows already sit at r * per
            if (mn_comm_allgather_dev(c, x->sh_sel.p + (size_t)rank * per * row_sel, x->sh_sel.p, (size_t)per * row_sel * sizeof(int), st) ||
                mn_comm_allgather_dev(c, x->sh_nsel.p + (size_t)rank * per * nlev, x->sh_nsel.p, (size_t)per * nlev * sizeof(int), st)) {
                set_err("mn_hnsw_build_shared: %s", mn_comm_last_error_str());
                x->broken = true;
                return -1;
            }
        }
        // every replica links the whole batch; with a divided batch the reverse edges' replay is divided too (link_batch)
        if (batch_link_impl(x, x->sh_sel.p, x->sh_nsel.p, split && !getenv("MN_SHARED_LINK_REPLICATED") ? c : nullptr)) {
            x->broken = true;
            return -1;
        }
    }
    return 0;
} MN_GUARD_END(set_err, if (x) x->broken = true, -1)

extern "C" int mn_hnsw_search_sharded_dev(mn_index *x, mn_comm *c, const float *d_queries, int64_t nq, int k, int ef,
                                          int64_t *d_ids, float *d_dists, int *d_counts) try {
    if (use_device(x))
        return -1;
    if (nq <= 0)
        return 0;
    const int world = c ? c->world : 1, rank = c ? c->rank : 0;
    if (world > 64) {
        set_err("mn_hnsw_search_sharded: more than 64 shards");
        return -1;
    }
    hipStream_t st = x->stream;
    const size_t per = (size_t)nq * k;
    if (x->sh_gids.reserve(per * world, false, st) || x->sh_gd.reserve(per * world, false, st) ||
        x->sh_gcnt.reserve((size_t)nq * world, false, st))
        return -1;
    // this shard's top-k straight into its slot of the gather buffers, then the one exchange step, then the merge
    long long *my_ids = x->sh_gids.p + per * rank;
    float *my_d = x->sh_gd.p + per * rank;
    int *my_c = x->sh_gcnt.p + (size_t)nq * rank;
    if (mn_hnsw_search_batch_dev(x, d_queries, nq, k, ef, (int64_t *)my_ids, my_d, my_c))
        return -1;
    // this shard's heap-workspace overflow count travels with its lists: a truncated list must not be merged silently
    if (x->sh_ovf.reserve((size_t)world, false, st))
        return -1;
    if (x->entry_id == -1 || x->node_count == 0 || !x->ws_counters.p)
        HIPCHK(hipMemsetAsync(x->sh_ovf.p + rank, 0, sizeof(unsigned long long), st));
    else
        HIPCHK(hipMemcpyAsync(x->sh_ovf.p + rank, x->ws_counters.p + 2, sizeof(unsigned long long), hipMemcpyDeviceToDevice, st));
    if (const char *fi = getenv("MN_FAULT_INJECT")) { // test hook: "search_overflow:<rank>" = this shard reports 3 truncated lists
        int fr = -1;
        const unsigned long long three = 3;
        if (sscanf(fi, "search_overflow:%d", &fr) == 1 && fr == rank) {
            HIPCHK(hipMemcpyAsync(x->sh_ovf.p + rank, &three, sizeof(three), hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st));
        }
    }
    x->sh_ovf_pending = world;
    if (world > 1 || (c && c->nccl)) {
        if (mn_comm_allgather_dev(c, my_ids, x->sh_gids.p, per * sizeof(long long), st) ||
            mn_comm_allgather_dev(c, my_d, x->sh_gd.p, per * sizeof(float), st) ||
            mn_comm_allgather_dev(c, x->sh_ovf.p + rank, x->sh_ovf.p, sizeof(unsigned long long), st) ||
            mn_comm_allgather_dev(c, my_c, x->sh_gcnt.p, (size_t)nq * sizeof(int), st)) {
            set_err("mn_hnsw_search_sharded: %s", mn_comm_last_error_str());
            return -1;
        }
    }
    mn_launch_merge_topk(x->sh_gids.p, x->sh_gd.p, x->sh_gcnt.p, world, nq, k, (long long *)d_ids, d_dists, d_counts, st);
    HIPCHK(hipGetLastError());
    return 0;
} MN_GUARD_END(set_err, MN_NOTHING, -1)

extern "C" int mn_hnsw_search_sharded(mn_index *x, mn_comm *c, const float *queries, int64_t nq, int k, int ef, int64_t *out_ids,
                                      float *out_dists, int *out_counts) try {
    if (use_device(x))
        return -1;
    if (nq <= 0)
        return 0;
    hipStream_t st = x->stream;
    if (x->ws_q.reserve((size_t)nq * x->dim, false, st) || x->sh_lids.reserve((size_t)nq * k, false, st) ||
        x->sh_ld.reserve((size_t)nq * k, false, st) || x->sh_lcnt.reserve((size_t)nq, false, st))
        return -1;
    HIPCHK(hipMemcpyAsync(x->ws_q.p, queries, (size_t)nq * x->dim * sizeof(float), hipMemcpyHostToDevice, st));
    if (mn_hnsw_search_sharded_dev(x, c, x->ws_q.p, nq, k, ef, (int64_t *)x->sh_lids.p, x->sh_ld.p, x->sh_lcnt.p))
        return -1;
    HIPCHK(hipMemcpyAsync(out_ids, x->sh_lids.p, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out_dists, x->sh_ld.p, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out_counts, x->sh_lcnt.p, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return check_sharded_overflow(x);
} MN_GUARD_END(set_err, MN_NOTHING, -1)

// ───────────────────────── delete (cold path, host-side list surgery) ─────────────────────────
// hnsw_delete edits the lists of the deleted node's neighbours only (a few dozen rows).  Those rows are worked on as
// plain growable lists — the reference's node_add_neighbor grows a list without bound (src/hnsw_algo.c:142-163) — read
// from the host mirror when it is current and otherwise row by row from HBM, and written back the same way.  Only when
// a list ends up longer than the row stride is the table re-strided (rare: reconnection under heavy deletion at tiny M).

struct RowEdit {
    mn_index *x;
    std::unordered_map<unsigned long long, std::vector<int>> rows;
    std::vector<unsigned long long> modified;
    static unsigned long long key(int slot, int level) { return ((unsigned long long)level << 32) | (unsigned)slot; }
    std::vector<int> *get(int slot, int level) {
        const unsigned long long k = key(slot, level);
        auto it = rows.find(k);
        if (it != rows.end())
            return &it->second;
        const int W = level == 0 ? x->W0 : x->WU;
        std::vector<int> buf((size_t)W, -1);
        if (x->host_links_valid) {
            int w2;
            memcpy(buf.data(), h_row(x, slot, level, &w2), (size_t)W * sizeof(int));
        } else {
            const int *src = (level == 0 ? x->d_links0.p : x->d_links_up.p) + d_row_off(x, slot, level);
            if (hipMemcpy(buf.data(), src, (size_t)W * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
                return nullptr;
        }
        buf.resize((size_t)h_row_count(buf.data(), W));
        return &(rows[k] = std::move(buf));
    }
    void touch(int slot, int level) { modified.push_back(key(slot, level)); }
};

static int grown_stride(int W, int need) { // a little headroom, so that a run of deletes does not re-stride every time
    int nw = std::max(need, W + std::max(16, W / 2));
    return (nw + 15) & ~15;
}

extern "C" int mn_hnsw_delete(mn_index *x, int64_t id) try { // src/hnsw_algo.c:717-805
    if (use_device(x))
        return -1;
    const int s = ht_find(x, id);
    if (s < 0 || x->deleted[s])
        return -1;
    if (push_links(x) || sync_meta(x)) // rows are read from whichever copy is current
        return -1;
    HIPCHK(hipStreamSynchronize(x->stream));
    RowEdit ed{x, {}, {}};
    const int min_conn = x->M / 2;
    auto live_at = [&](int n, int l) { return !x->deleted[n] && l <= x->levels[n]; };
    // (the reference flags the node deleted first, :722; nothing below looks at that flag for s itself)
    for (int l = 0; l <= x->levels[s]; l++) {
        std::vector<int> *own = ed.get(s, l);
        if (!own) {
            set_err("mn_hnsw_delete: cannot read neighbour rows");
            return -1;
        }
        const std::vector<int> former(*own); // :731-738
        const int nc = (int)former.size();
        for (int i = 0; i < nc; i++) { // :741-746 node_remove_neighbor: swap with last (:166-177)
            const int nb = former[i];
            if (x->deleted[nb] || l > x->levels[nb])
                continue;
            std::vector<int> *r = ed.get(nb, l);
            if (!r)
                return -1;
            for (size_t k = 0; k < r->size(); k++)
                if ((*r)[k] == s) {
                    (*r)[k] = r->back();
                    r->pop_back();
                    ed.touch(nb, l);
                    break;
                }
        }
        for (int i = 0; i < nc; i++) { // :750-785 reconnect neighbours left with fewer than M/2 links
            const int orphan = former[i];
            if (!live_at(orphan, l))
                continue;
            std::vector<int> *orow = ed.get(orphan, l);
            if (!orow)
                return -1;
            if ((int)orow->size() >= min_conn)
                continue;
            for (int j = 0; j < nc && (int)orow->size() < min_conn; j++) {
                if (i == j)
                    continue;
                const int cand = former[j];
                if (!live_at(cand, l))
                    continue;
                if (std::find(orow->begin(), orow->end(), cand) != orow->end())
                    continue;
                orow->push_back(cand); // node_add_neighbor(orphan, l, cand): not present (checked above)
                ed.touch(orphan, l);
                std::vector<int> *crow = ed.get(cand, l);
                if (!crow)
                    return -1;
                orow = ed.get(orphan, l); // (the map may have rehashed)
                if (std::find(crow->begin(), crow->end(), orphan) == crow->end()) { // node_add_neighbor skips duplicates (:147-150)
                    crow->push_back(orphan);
                    ed.touch(cand, l);
                }
            }
        }
    }
    // longest list per kind of row → does the table need a longer stride?
    int need0 = 0, needU = 0;
    for (unsigned long long k : ed.modified) {
        const int len = (int)ed.rows[k].size();
        ((k >> 32) == 0 ? need0 : needU) = std::max((k >> 32) == 0 ? need0 : needU, len);
    }
    if (std::max(need0, needU) > MN_MAX_ROW) {
        set_err("mn_hnsw_delete: a neighbour list would grow to %d entries (limit %d)", std::max(need0, needU), MN_MAX_ROW);
        return -1;
    }
    if (need0 > x->W0 && restride(x, true, grown_stride(x->W0, need0)))
        return -1;
    if (needU > x->WU && restride(x, false, grown_stride(x->WU, needU)))
        return -1;
    std::sort(ed.modified.begin(), ed.modified.end());
    ed.modified.erase(std::unique(ed.modified.begin(), ed.modified.end()), ed.modified.end());
    for (unsigned long long k : ed.modified) {
        const int slot = (int)(k & 0xffffffffu), level = (int)(k >> 32);
        const std::vector<int> &list = ed.rows[k];
        const int W = level == 0 ? x->W0 : x->WU;
        mark_stale(x, slot);
        std::vector<int> padded((size_t)W, -1);
        std::copy(list.begin(), list.end(), padded.begin());
        if (x->host_links_valid) {
            int w2;
            memcpy(h_row(x, slot, level, &w2), padded.data(), (size_t)W * sizeof(int));
            if (!x->dev_links_all)
                x->h_dirty_rows.push_back({slot, level});
            x->dev_links_stale = true;
        } else {
            int *dst = (level == 0 ? x->d_links0.p : x->d_links_up.p) + d_row_off(x, slot, level);
            HIPCHK(hipMemcpy(dst, padded.data(), (size_t)W * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    x->deleted[s] = 1; // :722-723
    x->node_count--;
    x->n_deleted++;
    if (x->entry_id == id) { // :790-802 scan in hash-table order, strict >
        x->entry_id = -1;
        x->max_level = -1;
        for (int i = 0; i < x->ht_cap; i++) {
            int t = x->ht[i];
            if (t >= 0 && !x->deleted[t] && x->levels[t] > x->max_level) {
                x->max_level = x->levels[t];
                x->entry_id = x->ids[t];
            }
        }
    }
    if (s < x->meta_uploaded)
        HIPCHK(hipMemcpy(x->d_deleted.p + s, &x->deleted[s], 1, hipMemcpyHostToDevice));
    return 0;
} MN_GUARD_END(set_err, if (x) x->broken = true, -1)

// ───────────────────────── inspection / load ─────────────────────────

extern "C" int mn_hnsw_get_vector(mn_index *x, int64_t id, float *out) try {
    if (use_device(x))
        return -1;
    int s = ht_find(x, id);
    if (s < 0 || x->deleted[s])
        return -1;
    if (!x->load_vecs.empty() && sync_meta(x))
        return -1;
    HIPCHK(hipMemcpy(out, x->d_vectors.p + (size_t)s * x->ld, (size_t)x->dim * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
} MN_GUARD_END(set_err, MN_NOTHING, -1)

extern "C" int mn_hnsw_node_count(mn_index *x) { return x->node_count; }
extern "C" int64_t mn_hnsw_entry_point(mn_index *x) { return x->entry_id; }
extern "C" int mn_hnsw_max_level(mn_index *x) { return x->max_level; }
extern "C" int mn_hnsw_node_level(mn_index *x, int64_t id) try {
    int s = ht_find(x, id);
    return s < 0 ? -1 : x->levels[s];
} MN_GUARD_END(set_err, MN_NOTHING, -1)
extern "C" int mn_hnsw_node_deleted(mn_index *x, int64_t id) try {
    int s = ht_find(x, id);
    return s < 0 ? -1 : x->deleted[s];
} MN_GUARD_END(set_err, MN_NOTHING, -1)

extern "C" int mn_hnsw_neighbors(mn_index *x, int64_t id, int level, int64_t *out, int cap) try {
    if (use_device(x))
        return -1;
    int s = ht_find(x, id);
    if (s < 0 || level > x->levels[s])
        return -1;
    int W = level == 0 ? x->W0 : x->WU;
    std::vector<int> tmp;
    const int *row;
    if (x->host_links_valid) {
        row = h_row(x, s, level, &W);
    } else { // one row straight from HBM: do not pull the whole graph for a point read
        tmp.resize((size_t)W);
        HIPCHK(hipStreamSynchronize(x->stream));
        HIPCHK(hipMemcpy(tmp.data(), (level == 0 ? x->d_links0.p : x->d_links_up.p) + d_row_off(x, s, level),
                         (size_t)W * sizeof(int), hipMemcpyDeviceToHost));
        row = tmp.data();
    }
    int n = h_row_count(row, W);
    for (int i = 0; i < n && i < cap; i++)
        out[i] = x->ids[row[i]];
    return n;
} MN_GUARD_END(set_err, MN_NOTHING, -1)

extern "C" int mn_hnsw_load_node(mn_index *x, int64_t id, const float *vector, int level, int deleted) try {
    if (use_device(x))
        return -1;
    if (level < 0 || level >= 32)
        return -1;
    if (pull_links(x))
        return -1;
    if (x->node_count * 10 > x->ht_cap * 7) // src/hnsw_vtab.c:304-306
        ht_grow(x);
    int s = host_add_node(x, id, level, deleted);
    if (s < 0) {
        // Table full (it grows on the LIVE count while soft-deleted rows keep their entries): the reference's load loop
        // ignores ht_insert's failure and carries on without the node (src/hnsw_vtab.c:316); so does this one.
        // A duplicate id takes the same path there.
        return 1;
    }
    if (!deleted)
        x->node_count++;
    else
        x->n_deleted++;
    // host only: the vector waits for one bulk upload (sync_meta) instead of a copy + two synchronisations per row
    if (x->load_vecs.empty())
        x->load_first = s;
    x->load_vecs.insert(x->load_vecs.end(), vector, vector + x->dim);
    x->dev_links_stale = true; // new (empty) rows of the mirror
    x->dev_links_all = true;
    if (x->load_vecs.size() * sizeof(float) >= (256u << 20))
        return sync_meta(x);
    return 0;
} MN_GUARD_END(set_err, if (x) x->broken = true, -1)

extern "C" int mn_hnsw_load_neighbors(mn_index *x, int64_t id, int level, const int64_t *nbrs, int n) try {
    if (use_device(x))
        return -1;
    int s = ht_find(x, id);
    if (s < 0 || level > x->levels[s])
        return -1;
    if (pull_links(x))
        return -1;
    for (int i = 0; i < n; i++) {
        int t = ht_find(x, nbrs[i]);
        if (t < 0) { // (a row the caller holds and the index does not: a whole rewrite of s is what removes it)
            mark_stale(x, s);
            continue;
        }
        int W;
        int *row = h_row(x, s, level, &W);
        int cnt = h_row_count(row, W);
        bool present = false; // node_add_neighbor (src/hnsw_algo.c:142-163): no duplicates, grows as needed
        for (int k = 0; k < cnt; k++)
            present |= row[k] == t;
        if (present)
            continue;
        if (cnt >= W) { // a list longer than M_max (left by deletes in the database that is being loaded)
            if (cnt + 1 > MN_MAX_ROW) {
                set_err("mn_hnsw_load_neighbors: more than %d neighbours at level %d", MN_MAX_ROW, level);
                return -1;
            }
            if (restride(x, level == 0, grown_stride(W, cnt + 1)))
                return -1;
            row = h_row(x, s, level, &W);
        }
        row[cnt] = t;
    }
    x->dev_links_stale = true;
    x->dev_links_all = true;
    return 0;
} MN_GUARD_END(set_err, if (x) x->broken = true, -1)

extern "C" int mn_hnsw_set_entry(mn_index *x, int64_t entry, int max_level) try {
    x->entry_id = entry;
    x->max_level = max_level;
    return 0;
} MN_GUARD_END(set_err, if (x) x->broken = true, -1)

// ───────────────────────── bulk export ─────────────────────────

extern "C" int mn_hnsw_slot_count(mn_index *x) { return x->n_slots; }

extern "C" int mn_hnsw_export_nodes(mn_index *x, int64_t *ids, int *levels, int *deleted) try {
    for (int s = 0; s < x->n_slots; s++) {
        ids[s] = x->ids[s];
        levels[s] = x->levels[s];
        deleted[s] = x->deleted[s];
    }
    return 0;
} MN_GUARD_END(set_err, MN_NOTHING, -1)

extern "C" int mn_hnsw_export_vectors(mn_index *x, float *out) try {
    if (use_device(x))
        return -1;
    if (x->n_slots == 0)
        return 0;
    if (sync_meta(x))
        return -1;
    HIPCHK(hipStreamSynchronize(x->stream));
    HIPCHK(hipMemcpy2D(out, (size_t)x->dim * sizeof(float), x->d_vectors.p, (size_t)x->ld * sizeof(float),
                       (size_t)x->dim * sizeof(float), (size_t)x->n_slots, hipMemcpyDeviceToHost));
    return 0;
} MN_GUARD_END(set_err, MN_NOTHING, -1)

extern "C" int mn_hnsw_row_width(mn_index *x, int level) { return level == 0 ? x->W0 : x->WU; }

extern "C" int mn_hnsw_export_links(mn_index *x, int level, int *out, int *width) try {
    if (use_device(x))
        return -1;
    if (pull_links(x))
        return -1;
    const int W = level == 0 ? x->W0 : x->WU;
    *width = W;
    for (int s = 0; s < x->n_slots; s++) {
        int *dst = out + (size_t)s * W;
        if (level > x->levels[s]) {
            for (int i = 0; i < W; i++)
                dst[i] = -1;
        } else {
            int w2;
            const int *row = h_row(x, s, level, &w2);
            memcpy(dst, row, (size_t)W * sizeof(int));
        }
    }
    return 0;
} MN_GUARD_END(set_err, MN_NOTHING, -1)

extern "C" int64_t mn_hnsw_edges_of(mn_index *x, const int64_t *ids, int n, int64_t *out_src, int64_t *out_dst,
                                    int *out_level, float *out_dist, int64_t cap) try {
    if (use_device(x))
        return -1;
    if (push_links(x) || sync_meta(x))
        return -1;
    std::vector<int> rs, rl;
    for (int i = 0; i < n; i++) {
        int s = ht_find(x, ids[i]);
        if (s < 0) {
            set_err("mn_hnsw_edges_of: unknown id %lld", (long long)ids[i]);
            return -1;
        }
        for (int l = 0; l <= x->levels[s]; l++) {
            rs.push_back(s);
            rl.push_back(l);
        }
    }
    const int R = (int)rs.size();
    if (R == 0)
        return 0;
    hipStream_t st = x->stream;
    const int W0 = std::max(x->W0, x->WU); // rows of either kind are written with the common stride WX
    if (x->er_slot.reserve((size_t)R, false, st)) return -1;
    if (x->er_level.reserve((size_t)R, false, st)) return -1;
    if (x->er_nbr.reserve((size_t)R * W0, false, st)) return -1;
    if (x->er_dist.reserve((size_t)R * W0, false, st)) return -1;
    HIPCHK(hipMemcpyAsync(x->er_slot.p, rs.data(), (size_t)R * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(x->er_level.p, rl.data(), (size_t)R * sizeof(int), hipMemcpyHostToDevice, st));
    mn_launch_edge_rows(dev_view(x), x->er_slot.p, x->er_level.p, R, x->er_nbr.p, x->er_dist.p, st);
    HIPCHK(hipGetLastError());
    std::vector<int> nbr((size_t)R * W0);
    std::vector<float> dist((size_t)R * W0);
    HIPCHK(hipMemcpyAsync(nbr.data(), x->er_nbr.p, nbr.size() * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(dist.data(), x->er_dist.p, dist.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int64_t ne = 0;
    for (int r = 0; r < R; r++)
        for (int i = 0; i < W0; i++) {
            int t = nbr[(size_t)r * W0 + i];
            if (t < 0)
                break;
            if (ne < cap) {
                out_src[ne] = x->ids[rs[r]];
                out_dst[ne] = x->ids[t];
                out_level[ne] = rl[r];
                out_dist[ne] = dist[(size_t)r * W0 + i];
            }
            ne++;
        }
    return ne;
} MN_GUARD_END(set_err, MN_NOTHING, -1)

// ───────────────────────── measurement hooks ─────────────────────────

extern "C" int mn_hnsw_build_stats(mn_index *x, mn_build_stats *out, int reset) try {
    *out = x->bstats;
    if (reset)
        x->bstats = {0, 0, 0, 0, 0, 0};
    return 0;
} MN_GUARD_END(set_err, MN_NOTHING, -1)

extern "C" int mn_hnsw_last_launch(mn_index *x, mn_launch_stats *out) try {
    if (use_device(x))
        return -1;
    if (!x->last_on_host && fetch_counters(x)) // (a lone query's counters came back with its answer)
        return -1;
    *out = x->last;
    return 0;
} MN_GUARD_END(set_err, MN_NOTHING, -1)

extern "C" void *mn_dev_malloc(mn_index *x, size_t bytes) try {
    void *p = nullptr;
    if (hipSetDevice(x->device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) {
        set_err("mn_dev_malloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
} MN_GUARD_END(set_err, MN_NOTHING, nullptr)
extern "C" void mn_dev_free(mn_index *x, void *p) {
    (void)hipSetDevice(x->device);
    (void)hipFree(p);
}
extern "C" int mn_dev_upload(mn_index *x, void *dst, const void *src, size_t bytes) try {
    if (use_device(x))
        return -1;
    HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
} MN_GUARD_END(set_err, MN_NOTHING, -1)
extern "C" int mn_dev_download(mn_index *x, void *dst, const void *src, size_t bytes) try {
    if (use_device(x))
        return -1;
    HIPCHK(hipStreamSynchronize(x->stream));
    HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
} MN_GUARD_END(set_err, MN_NOTHING, -1)

extern "C" int mn_hnsw_bruteforce_topk(mn_index *x, const float *d_queries, int64_t nq, int k, int64_t *out_ids) try {
    if (use_device(x))
        return -1;
    if (push_links(x) || sync_meta(x))
        return -1;
    hipStream_t st = x->stream;
    if (x->ws_outi.reserve((size_t)nq * k, false, st)) return -1;
    const char *force = getenv("MN_BRUTE"); // "valu" = the index's own inner loop (any k <= 128); default: MFMA when k <= 16
    MnDevIndex v = dev_view(x);
    if (k <= 16 && nq > 0 && x->n_slots > 0 && !(force && !strcmp(force, "valu"))) {
        int nc = 0, rpc = 0;
        const size_t bytes = mn_brute_mfma_scratch_bytes(v, nq, k, &nc, &rpc);
        DevBuf<unsigned char> scratch;
        if (scratch.reserve(bytes, false, st))
            return -1;
        HIPCHK(hipEventRecord(x->ev0, st));
        const int rc = mn_launch_bruteforce_mfma(v, d_queries, nq, k, x->ws_outi.p, scratch.p, st);
        HIPCHK(hipEventRecord(x->ev1, st));
        hipError_t e = hipStreamSynchronize(st);
        scratch.release();
        if (rc != 0 || e != hipSuccess) {
            set_err("mn_hnsw_bruteforce_topk: MFMA kernel failed (%s)", hipGetErrorString(e));
            return -1;
        }
        float ms = 0;
        if (hipEventElapsedTime(&ms, x->ev0, x->ev1) == hipSuccess)
            x->last.last_kernel_ms = ms;
    } else {
        mn_launch_bruteforce(v, d_queries, nq, k, x->ws_outi.p, nullptr, st);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(out_ids, x->ws_outi.p, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
} MN_GUARD_END(set_err, MN_NOTHING, -1)
