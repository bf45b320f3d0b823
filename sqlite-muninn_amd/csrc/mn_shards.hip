// mn_shards.hip — BASELINE config 3 inside ONE process: an index sharded over the GPUs of a node (rowid mod n → one HNSW
// graph per GPU, each an ordinary mn_index), driven by one host thread.  Every call is made of the public per-index entry
// points (include/muninn_hip.h) plus peer copies and the merge kernel of mn_hnsw_search_sharded (k_merge_topk): the result
// of a search is the same list, bit for bit, as one rank per GPU exchanging over RCCL would return.
//   search: the queries are uploaded to every shard's GPU and all shards are searched at once (each on its own stream,
//           queued back to back by the one host thread), their top-k lists are copied to the first shard's GPU
//           (hipMemcpyPeerAsync over xGMI) and merged there in the total order (distance, shard, position).
//   build : ids are split by shard and the shards are built side by side, one host thread per GPU.
#include "../../include/muninn_hip.h"
#include "mn_guard.hpp"
#include "mn_device.hpp"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static thread_local std::string s_err;
static void sset_err(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    s_err = buf;
}
#define SCHK(expr)                                                                                   \
    do {                                                                                             \
        hipError_t e__ = (expr);                                                                     \
        if (e__ != hipSuccess) {                                                                     \
            sset_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__);    \
            return -1;                                                                               \
        }                                                                                            \
    } while (0)

struct ShardBuf { // per-shard device buffers, grown on demand (on that shard's GPU)
    float *q = nullptr;
    long long *ids = nullptr;
    float *d = nullptr;
    int *cnt = nullptr;
    size_t q_cap = 0, r_cap = 0, d_cap = 0, c_cap = 0;
};

struct mn_shards {
    int n = 0, dim = 0;
    std::vector<mn_index *> ix;
    std::vector<int> dev;
    std::vector<ShardBuf> buf;
    // gather + merge buffers on dev[0]
    long long *g_ids = nullptr, *o_ids = nullptr;
    float *g_d = nullptr, *o_d = nullptr;
    int *g_cnt = nullptr, *o_cnt = nullptr;
    size_t g_cap = 0, gd_cap = 0, gc_cap = 0, o_cap = 0, od_cap = 0, oc_cap = 0;
    hipStream_t st0 = nullptr;
};

template <typename T> static int grow(T **p, size_t *cap, size_t need) {
    if (need <= *cap)
        return 0;
    if (*p)
        (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    SCHK(hipMalloc(p, need * sizeof(T)));
    *cap = need;
    return 0;
}

extern "C" const char *mn_shards_last_error(void) { return s_err.c_str(); }

extern "C" void mn_shards_destroy(mn_shards *s) {
    if (!s)
        return;
    for (int i = 0; i < s->n; i++) {
        if ((size_t)i < s->buf.size() && hipSetDevice(s->dev[i]) == hipSuccess) {
            (void)hipFree(s->buf[i].q); (void)hipFree(s->buf[i].ids); (void)hipFree(s->buf[i].d); (void)hipFree(s->buf[i].cnt);
        }
        if ((size_t)i < s->ix.size() && s->ix[i])
            mn_hnsw_destroy(s->ix[i]);
    }
    if (s->n && hipSetDevice(s->dev[0]) == hipSuccess) {
        (void)hipFree(s->g_ids); (void)hipFree(s->g_d); (void)hipFree(s->g_cnt);
        (void)hipFree(s->o_ids); (void)hipFree(s->o_d); (void)hipFree(s->o_cnt);
        if (s->st0)
            (void)hipStreamDestroy(s->st0);
    }
    delete s;
}

extern "C" mn_shards *mn_shards_create(int dim, int metric, int M, int ef_construction, const int *devices, int n) try {
    if (n < 1 || n > 64 || !devices) {
        sset_err("mn_shards_create: 1..64 shards");
        return nullptr;
    }
    mn_shards *s = new mn_shards();
    s->n = n;
    s->dim = dim;
    s->dev.assign(devices, devices + n);
    s->buf.resize((size_t)n);
    s->ix.assign((size_t)n, nullptr);
    for (int i = 0; i < n; i++) {
        s->ix[i] = mn_hnsw_create_on(dim, metric, M, ef_construction, devices[i]);
        if (!s->ix[i]) {
            sset_err("mn_shards_create: shard %d on device %d: %s", i, devices[i], mn_last_error());
            mn_shards_destroy(s);
            return nullptr;
        }
    }
    if (hipSetDevice(devices[0]) != hipSuccess || hipStreamCreateWithFlags(&s->st0, hipStreamNonBlocking) != hipSuccess) {
        sset_err("mn_shards_create: no stream on device %d", devices[0]);
        mn_shards_destroy(s);
        return nullptr;
    }
    for (int i = 1; i < n; i++) // peer access where the devices differ (xGMI); copies fall back to staging otherwise
        if (devices[i] != devices[0]) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[0], devices[i]) == hipSuccess && can)
                (void)hipDeviceEnablePeerAccess(devices[i], 0);
        }
    (void)hipGetLastError();
    return s;
} MN_GUARD_END(sset_err, MN_NOTHING, nullptr)

extern "C" int mn_shards_count(const mn_shards *s) { return s ? s->n : 0; }
extern "C" mn_index *mn_shards_index(mn_shards *s, int i) { return s && i >= 0 && i < s->n ? s->ix[i] : nullptr; }
extern "C" int mn_shards_of(const mn_shards *s, int64_t id) { return (int)(((id % s->n) + s->n) % s->n); }

extern "C" int mn_shards_set_order(mn_shards *s, int order) try {
    for (int i = 0; i < s->n; i++)
        if (mn_hnsw_set_order(s->ix[i], order)) {
            sset_err("%s", mn_last_error());
            return -1;
        }
    return 0;
} MN_GUARD_END(sset_err, MN_NOTHING, -1)

extern "C" int mn_shards_insert(mn_shards *s, int64_t id, const float *vector) try {
    const int rc = mn_hnsw_insert(s->ix[mn_shards_of(s, id)], id, vector);
    if (rc)
        sset_err("%s", mn_last_error());
    return rc;
} MN_GUARD_END(sset_err, MN_NOTHING, -1)

extern "C" int mn_shards_delete(mn_shards *s, int64_t id) try {
    const int rc = mn_hnsw_delete(s->ix[mn_shards_of(s, id)], id);
    if (rc)
        sset_err("%s", mn_last_error());
    return rc;
} MN_GUARD_END(sset_err, MN_NOTHING, -1)

extern "C" int mn_shards_build(mn_shards *s, const int64_t *ids, const float *vectors, int64_t n, int grow_div, int max_batch) try {
    const int ns = s->n;
    std::vector<std::vector<int64_t>> sid((size_t)ns);
    std::vector<std::vector<float>> sv((size_t)ns);
    for (int64_t i = 0; i < n; i++) { // input order is kept inside every shard (it decides the graph)
        const int r = mn_shards_of(s, ids[i]);
        sid[r].push_back(ids[i]);
        sv[r].insert(sv[r].end(), vectors + (size_t)i * s->dim, vectors + (size_t)(i + 1) * s->dim);
    }
    std::vector<int> rc((size_t)ns, 0);
    std::vector<std::string> msg((size_t)ns);
    std::vector<std::thread> th;
    for (int r = 0; r < ns; r++)
        th.emplace_back([&, r]() {
            if (sid[r].empty())
                return;
            rc[r] = mn_hnsw_build(s->ix[r], sid[r].data(), sv[r].data(), (int64_t)sid[r].size(), grow_div, max_batch);
            if (rc[r])
                msg[r] = mn_last_error();
        });
    for (auto &t : th)
        t.join();
    for (int r = 0; r < ns; r++)
        if (rc[r]) {
            sset_err("mn_shards_build: shard %d: %s", r, msg[r].c_str());
            return -1;
        }
    return 0;
} MN_GUARD_END(sset_err, MN_NOTHING, -1)

// queries host [nq][dim]; out_* host [nq][k] / [nq]
extern "C" int mn_shards_search_batch(mn_shards *s, const float *queries, int64_t nq, int k, int ef_search, int64_t *out_ids,
                                      float *out_dists, int *out_counts) try {
    if (nq <= 0)
        return 0;
    if (k < 1) {
        sset_err("mn_shards_search_batch: k < 1");
        return -1;
    }
    const size_t per = (size_t)nq * k, qn = (size_t)nq * s->dim;
    const int ns = s->n;
    // every shard: queries in, search queued (asynchronous on the shard's own stream → the GPUs work side by side)
    for (int r = 0; r < ns; r++) {
        ShardBuf &b = s->buf[r];
        SCHK(hipSetDevice(s->dev[r]));
        if (grow(&b.q, &b.q_cap, qn) || grow(&b.ids, &b.r_cap, per) || grow(&b.d, &b.d_cap, per) ||
            grow(&b.cnt, &b.c_cap, (size_t)nq))
            return -1;
        SCHK(hipMemcpy(b.q, queries, qn * sizeof(float), hipMemcpyHostToDevice));
        if (mn_hnsw_search_batch_dev(s->ix[r], b.q, nq, k, ef_search, (int64_t *)b.ids, b.d, b.cnt)) {
            sset_err("mn_shards_search_batch: shard %d: %s", r, mn_last_error());
            return -1;
        }
    }
    SCHK(hipSetDevice(s->dev[0]));
    if (grow(&s->g_ids, &s->g_cap, per * ns) || grow(&s->g_d, &s->gd_cap, per * ns) || grow(&s->g_cnt, &s->gc_cap, (size_t)nq * ns) ||
        grow(&s->o_ids, &s->o_cap, per) || grow(&s->o_d, &s->od_cap, per) || grow(&s->o_cnt, &s->oc_cap, (size_t)nq))
        return -1;
    // the shards' lists → shard 0's GPU, slot r of the gather buffers (the one exchange step of config 3)
    for (int r = 0; r < ns; r++) {
        long long novf = 0;
        if (mn_index_search_overflow(s->ix[r], &novf)) { // (synchronises the shard's stream)
            sset_err("mn_shards_search_batch: shard %d: %s", r, mn_last_error());
            return -1;
        }
        if (novf) { // the unsharded path fails the same way (mn_hnsw_search_batch): a truncated list is not merged
            sset_err("mn_shards_search_batch: shard %d: %lld queries exceeded heap workspace", r, novf);
            return -1;
        }
        const ShardBuf &b = s->buf[r];
        SCHK(hipMemcpyPeerAsync(s->g_ids + per * r, s->dev[0], b.ids, s->dev[r], per * sizeof(long long), s->st0));
        SCHK(hipMemcpyPeerAsync(s->g_d + per * r, s->dev[0], b.d, s->dev[r], per * sizeof(float), s->st0));
        SCHK(hipMemcpyPeerAsync(s->g_cnt + (size_t)nq * r, s->dev[0], b.cnt, s->dev[r], (size_t)nq * sizeof(int), s->st0));
    }
    mn_launch_merge_topk(s->g_ids, s->g_d, s->g_cnt, ns, nq, k, s->o_ids, s->o_d, s->o_cnt, s->st0);
    SCHK(hipGetLastError());
    SCHK(hipMemcpyAsync(out_ids, s->o_ids, per * sizeof(int64_t), hipMemcpyDeviceToHost, s->st0));
    SCHK(hipMemcpyAsync(out_dists, s->o_d, per * sizeof(float), hipMemcpyDeviceToHost, s->st0));
    SCHK(hipMemcpyAsync(out_counts, s->o_cnt, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, s->st0));
    SCHK(hipStreamSynchronize(s->st0));
    return 0;
} MN_GUARD_END(sset_err, MN_NOTHING, -1)

extern "C" int mn_shards_search(mn_shards *s, const float *query, int k, int ef_search, mn_search_result *results) try {
    if (k < 1) // (before anything is sized by k: nothing may throw across the C boundary)
        return 0;
    std::vector<int64_t> ids((size_t)k);
    std::vector<float> ds((size_t)k);
    int cnt = 0;
    if (mn_shards_search_batch(s, query, 1, k, ef_search, ids.data(), ds.data(), &cnt))
        return -1;
    for (int i = 0; i < cnt; i++) {
        results[i].id = ids[(size_t)i];
        results[i].distance = ds[(size_t)i];
    }
    return cnt;
} MN_GUARD_END(sset_err, MN_NOTHING, 0)
