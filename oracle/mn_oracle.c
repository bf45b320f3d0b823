/*
 * mn_oracle.c — CPU oracle: plain-C restatement of sqlite-muninn's HNSW hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see mn_oracle.h).  Not linked into the product library.
 * Build: gcc -O2 -std=c11 -ffp-contract=off (the reference's Linux flags are -O2 -std=c11 with
 * no -march, i.e. the SSE path of src/vec_math.c:75-144 and no FMA contraction).
 *
 * Citations are file:line in the reference repository.
 */
#include "mn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ───────────────────────── a1-a3: distances ───────────────────────── */

/* src/vec_math.c:78-96 — four lane accumulators (lane j sums i ≡ j mod 4 in order), mul then
 * add, horizontal sum tmp[0]+tmp[1]+tmp[2]+tmp[3] (left-assoc), then the scalar tail. */
float orc_vec_l2(const float *a, const float *b, int dim) {
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int i = 0;
    for (; i + 4 <= dim; i += 4) {
        float d0 = a[i] - b[i], d1 = a[i + 1] - b[i + 1], d2 = a[i + 2] - b[i + 2], d3 = a[i + 3] - b[i + 3];
        float p0 = d0 * d0, p1 = d1 * d1, p2 = d2 * d2, p3 = d3 * d3;
        s0 = s0 + p0;
        s1 = s1 + p1;
        s2 = s2 + p2;
        s3 = s3 + p3;
    }
    float sum = ((s0 + s1) + s2) + s3;
    for (; i < dim; i++) {
        float d = a[i] - b[i];
        float p = d * d;
        sum = sum + p;
    }
    return sum;
}

/* dot product in the same SSE order (src/vec_math.c:102-121,:129-141) */
static float dot_sse(const float *a, const float *b, int dim) {
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int i = 0;
    for (; i + 4 <= dim; i += 4) {
        float p0 = a[i] * b[i], p1 = a[i + 1] * b[i + 1], p2 = a[i + 2] * b[i + 2], p3 = a[i + 3] * b[i + 3];
        s0 = s0 + p0;
        s1 = s1 + p1;
        s2 = s2 + p2;
        s3 = s3 + p3;
    }
    float sum = ((s0 + s1) + s2) + s3;
    for (; i < dim; i++) {
        float p = a[i] * b[i];
        sum = sum + p;
    }
    return sum;
}

/* src/vec_math.c:98-126 — dot, |a|², |b|² each in the SSE order; denom < 1e-30 → 1.0 */
float orc_vec_cosine(const float *a, const float *b, int dim) {
    float dot = dot_sse(a, b, dim);
    float na = dot_sse(a, a, dim);
    float nb = dot_sse(b, b, dim);
    float denom = sqrtf(na) * sqrtf(nb);
    if (denom < 1e-30f)
        return 1.0f;
    return 1.0f - (dot / denom);
}

/* src/vec_math.c:128-143 */
float orc_vec_ip(const float *a, const float *b, int dim) {
    return -dot_sse(a, b, dim);
}

/* ORC_ORDER_WAVE: restates sqlite-muninn_amd/csrc wave-order reduction.  Lane L of a 64-lane
 * wavefront owns elements e = 256k + 4L + j (k = chunk, j = 0..3) and folds them with one fmaf
 * chain in (k, j) order; the 64 partials are combined by an xor butterfly 32,16,8,4,2,1. */
static float wave_reduce(float *part) {
    for (int m = 32; m >= 1; m >>= 1) {
        float nxt[64];
        for (int l = 0; l < 64; l++)
            nxt[l] = part[l] + part[l ^ m];
        memcpy(part, nxt, sizeof(nxt));
    }
    return part[0];
}

/* rows are stored zero-padded to ld = round_up(dim, 4); a lane folds whole float4s */
static float dot_wave(const float *a, const float *b, int dim) {
    float part[64];
    const int ld = (dim + 3) & ~3;
    for (int l = 0; l < 64; l++) {
        float acc = 0.0f;
        for (int base = 0; base < ld; base += 256)
            for (int j = 0; j < 4; j++) {
                int e = base + 4 * l + j;
                if (e < ld) {
                    float ae = e < dim ? a[e] : 0.0f, be = e < dim ? b[e] : 0.0f;
                    acc = fmaf(ae, be, acc);
                }
            }
        part[l] = acc;
    }
    return wave_reduce(part);
}

static float l2_wave(const float *a, const float *b, int dim) {
    float part[64];
    const int ld = (dim + 3) & ~3;
    for (int l = 0; l < 64; l++) {
        float acc = 0.0f;
        for (int base = 0; base < ld; base += 256)
            for (int j = 0; j < 4; j++) {
                int e = base + 4 * l + j;
                if (e < ld) {
                    float d = e < dim ? a[e] - b[e] : 0.0f;
                    acc = fmaf(d, d, acc);
                }
            }
        part[l] = acc;
    }
    return wave_reduce(part);
}

static float cosine_finish(float dot, float na, float nb) {
    float denom = sqrtf(na) * sqrtf(nb);
    if (denom < 1e-30f)
        return 1.0f;
    return 1.0f - (dot / denom);
}

float orc_vec_distance(int metric, int order, const float *a, const float *b, int dim) {
    if (order == ORC_ORDER_WAVE) {
        switch (metric) {
        case ORC_METRIC_L2:
            return l2_wave(a, b, dim);
        case ORC_METRIC_COSINE:
            return cosine_finish(dot_wave(a, b, dim), dot_wave(a, a, dim), dot_wave(b, b, dim));
        default:
            return -dot_wave(a, b, dim);
        }
    }
    switch (metric) {
    case ORC_METRIC_L2:
        return orc_vec_l2(a, b, dim);
    case ORC_METRIC_COSINE:
        return orc_vec_cosine(a, b, dim);
    default:
        return orc_vec_ip(a, b, dim);
    }
}

int orc_vec_parse_metric(const char *name, int *out) {
    if (strcmp(name, "l2") == 0) {
        *out = ORC_METRIC_L2;
        return 0;
    }
    if (strcmp(name, "cosine") == 0) {
        *out = ORC_METRIC_COSINE;
        return 0;
    }
    if (strcmp(name, "inner_product") == 0) {
        *out = ORC_METRIC_IP;
        return 0;
    }
    return -1;
}

void orc_dist_batch(int metric, int order, const float *query, const float *rows, int64_t n, int dim, float *out) {
    for (int64_t i = 0; i < n; i++)
        out[i] = orc_vec_distance(metric, order, query, rows + (size_t)i * dim, dim);
}

/* ───────────────────────── a13: binary heap ───────────────────────── */

/* src/priority_queue.c:18-26 — stop when parent <= child */
static void pq_sift_up(orc_pq_item *it, int idx) {
    while (idx > 1) {
        int parent = idx / 2;
        if (it[parent].distance <= it[idx].distance)
            break;
        orc_pq_item t = it[parent];
        it[parent] = it[idx];
        it[idx] = t;
        idx = parent;
    }
}

/* src/priority_queue.c:28-42 — strict <, left child first */
static void pq_sift_down(orc_pq_item *it, int size, int idx) {
    for (;;) {
        int smallest = idx, left = 2 * idx, right = 2 * idx + 1;
        if (left <= size && it[left].distance < it[smallest].distance)
            smallest = left;
        if (right <= size && it[right].distance < it[smallest].distance)
            smallest = right;
        if (smallest == idx)
            break;
        orc_pq_item t = it[idx];
        it[idx] = it[smallest];
        it[smallest] = t;
        idx = smallest;
    }
}

int orc_pq_init(orc_pq *pq, int cap) {
    if (cap < 4)
        cap = 4;
    pq->items = (orc_pq_item *)malloc((size_t)(cap + 1) * sizeof(orc_pq_item));
    if (!pq->items)
        return -1;
    pq->size = 0;
    pq->capacity = cap;
    return 0;
}

int orc_pq_push(orc_pq *pq, int64_t id, float distance) {
    if (pq->size >= pq->capacity) {
        int nc = pq->capacity * 2;
        orc_pq_item *ni = (orc_pq_item *)realloc(pq->items, (size_t)(nc + 1) * sizeof(orc_pq_item));
        if (!ni)
            return -1;
        pq->items = ni;
        pq->capacity = nc;
    }
    pq->size++;
    pq->items[pq->size].id = id;
    pq->items[pq->size].distance = distance;
    pq_sift_up(pq->items, pq->size);
    return 0;
}

orc_pq_item orc_pq_pop(orc_pq *pq) {
    orc_pq_item top = pq->items[1];
    pq->items[1] = pq->items[pq->size];
    pq->size--;
    if (pq->size > 0)
        pq_sift_down(pq->items, pq->size, 1);
    return top;
}

void orc_pq_destroy(orc_pq *pq) {
    free(pq->items);
    pq->items = NULL;
    pq->size = pq->capacity = 0;
}

int orc_pq_trace(const int *ops, const int64_t *ids, const float *dists, int n, int64_t *out_ids, float *out_dists) {
    orc_pq pq;
    int np = 0;
    if (orc_pq_init(&pq, 4) != 0)
        return -1;
    for (int i = 0; i < n; i++) {
        if (ops[i]) {
            orc_pq_push(&pq, ids[i], dists[i]);
        } else if (pq.size > 0) {
            orc_pq_item it = orc_pq_pop(&pq);
            out_ids[np] = it.id;
            out_dists[np] = it.distance;
            np++;
        }
    }
    orc_pq_destroy(&pq);
    return np;
}

/* ───────────────────────── a5: index ───────────────────────── */

#define ORC_MAX_LEVELS 32 /* src/hnsw_algo.h:14 */

typedef struct {
    int *v;
    int n, cap;
} nlist;

struct orc_index {
    int dim, M, M_max0, efc, metric, order, visited_mode;
    double level_mult;
    int64_t entry_id; /* -1 if empty */
    int max_level;
    unsigned rng_state;
    int node_count; /* live nodes, src/hnsw_algo.h:49 */
    /* slot storage (slot = insertion order) */
    int n_slots, cap_slots;
    float *vectors;
    float *norms; /* |v|² in the active order (cosine) */
    int64_t *ids;
    int *levels;
    unsigned char *deleted;
    nlist **nb; /* nb[slot][level] */
    /* reference-compatible open-addressing table id -> slot (src/hnsw_algo.c:38-91) */
    int *ht;
    int ht_cap;
    /* visited (bitmap mode): epoch stamps */
    unsigned *vstamp;
    int vstamp_cap;
    unsigned vepoch;
    /* MN-RU: membership stamps of the list being pruned (same counts as the reference's scan, O(deg) per neighbour) */
    unsigned *mstamp;
    int mstamp_cap;
    unsigned mepoch;
    orc_stats st;
};

/* src/hnsw_algo.c:19-30 */
static unsigned xorshift32(unsigned *state) {
    unsigned x = *state;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    *state = x;
    return x;
}

/* src/hnsw_algo.c:38-47 */
static int ht_slot(int64_t id, int capacity) {
    uint64_t h = (uint64_t)id;
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdULL;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ULL;
    h ^= h >> 33;
    return (int)(h & (uint64_t)(capacity - 1));
}

/* src/hnsw_algo.c:49-59 — returns slot or -1 */
static int ht_find(const orc_index *x, int64_t id) {
    int s = ht_slot(id, x->ht_cap);
    for (int i = 0; i < x->ht_cap; i++) {
        int p = (s + i) & (x->ht_cap - 1);
        if (x->ht[p] < 0)
            return -1;
        if (x->ids[x->ht[p]] == id)
            return x->ht[p];
    }
    return -1;
}

/* src/hnsw_algo.c:61-74 */
static int ht_put(int *table, int cap, const int64_t *ids, int slot) {
    int s = ht_slot(ids[slot], cap);
    for (int i = 0; i < cap; i++) {
        int p = (s + i) & (cap - 1);
        if (table[p] < 0) {
            table[p] = slot;
            return 0;
        }
        if (ids[table[p]] == ids[slot])
            return -1;
    }
    return -1;
}

/* src/hnsw_algo.c:76-91 — rehash in old-table order */
static int ht_grow(orc_index *x) {
    int nc = x->ht_cap * 2;
    int *nt = (int *)malloc((size_t)nc * sizeof(int));
    if (!nt)
        return -1;
    for (int i = 0; i < nc; i++)
        nt[i] = -1;
    for (int i = 0; i < x->ht_cap; i++)
        if (x->ht[i] >= 0)
            ht_put(nt, nc, x->ids, x->ht[i]);
    free(x->ht);
    x->ht = nt;
    x->ht_cap = nc;
    return 0;
}

static float node_dist(orc_index *x, const float *q, float qnorm, int slot) {
    const float *v = x->vectors + (size_t)slot * x->dim;
    x->st.n_dist++;
    if (x->metric == ORC_METRIC_COSINE) {
        /* same bits as recomputing both norms inside the loop (src/vec_math.c:104-121): the three
         * sums are independent accumulations in the same order */
        float dot = (x->order == ORC_ORDER_WAVE) ? dot_wave(q, v, x->dim) : dot_sse(q, v, x->dim);
        return cosine_finish(dot, qnorm, x->norms[slot]);
    }
    return orc_vec_distance(x->metric, x->order, q, v, x->dim);
}

static float vec_norm(const orc_index *x, const float *v) {
    if (x->metric != ORC_METRIC_COSINE)
        return 0.0f;
    return (x->order == ORC_ORDER_WAVE) ? dot_wave(v, v, x->dim) : dot_sse(v, v, x->dim);
}

orc_index *orc_hnsw_create(int dim, int metric, int M, int ef_construction) {
    orc_index *x = (orc_index *)calloc(1, sizeof(orc_index));
    if (!x)
        return NULL;
    x->dim = dim;
    x->M = M;
    x->M_max0 = 2 * M; /* :188 */
    x->efc = ef_construction;
    x->metric = metric;
    x->order = ORC_ORDER_SSE;
    x->visited_mode = ORC_VISITED_BITMAP;
    x->level_mult = 1.0 / log((double)M); /* :192 */
    x->entry_id = -1;
    x->max_level = -1;
    x->ht_cap = 256; /* :197 */
    x->ht = (int *)malloc((size_t)x->ht_cap * sizeof(int));
    if (!x->ht) {
        free(x);
        return NULL;
    }
    for (int i = 0; i < x->ht_cap; i++)
        x->ht[i] = -1;
    x->rng_state = 42; /* :205 */
    return x;
}

void orc_hnsw_destroy(orc_index *x) {
    if (!x)
        return;
    for (int s = 0; s < x->n_slots; s++) {
        for (int l = 0; l <= x->levels[s]; l++)
            free(x->nb[s][l].v);
        free(x->nb[s]);
    }
    free(x->nb);
    free(x->vectors);
    free(x->norms);
    free(x->ids);
    free(x->levels);
    free(x->deleted);
    free(x->ht);
    free(x->vstamp);
    free(x->mstamp);
    free(x);
}

void orc_hnsw_seed_rng(orc_index *x, unsigned seed) {
    x->rng_state = seed ? seed : 1;
}

void orc_hnsw_set_order(orc_index *x, int order) {
    x->order = order;
    for (int s = 0; s < x->n_slots; s++)
        x->norms[s] = vec_norm(x, x->vectors + (size_t)s * x->dim);
}

void orc_hnsw_set_visited(orc_index *x, int mode) {
    x->visited_mode = mode;
}

/* src/hnsw_algo.c:240-248 */
int orc_hnsw_random_level(orc_index *x) {
    double r = (double)xorshift32(&x->rng_state) / (double)0xFFFFFFFFu;
    if (r == 0.0)
        r = 1e-10;
    int level = (int)(-log(r) * x->level_mult);
    if (level >= ORC_MAX_LEVELS)
        level = ORC_MAX_LEVELS - 1;
    return level;
}

static int slots_reserve(orc_index *x, int need) {
    if (need <= x->cap_slots)
        return 0;
    int nc = x->cap_slots ? x->cap_slots : 1024;
    while (nc < need)
        nc *= 2;
    float *nv = (float *)realloc(x->vectors, (size_t)nc * x->dim * sizeof(float));
    if (!nv)
        return -1;
    x->vectors = nv;
    float *nn = (float *)realloc(x->norms, (size_t)nc * sizeof(float));
    if (!nn)
        return -1;
    x->norms = nn;
    int64_t *ni = (int64_t *)realloc(x->ids, (size_t)nc * sizeof(int64_t));
    if (!ni)
        return -1;
    x->ids = ni;
    int *nl = (int *)realloc(x->levels, (size_t)nc * sizeof(int));
    if (!nl)
        return -1;
    x->levels = nl;
    unsigned char *nd = (unsigned char *)realloc(x->deleted, (size_t)nc);
    if (!nd)
        return -1;
    x->deleted = nd;
    nlist **nb = (nlist **)realloc(x->nb, (size_t)nc * sizeof(nlist *));
    if (!nb)
        return -1;
    x->nb = nb;
    x->cap_slots = nc;
    return 0;
}

/* node_create (src/hnsw_algo.c:95-124) + ht_insert; returns slot or -1 */
static int node_new(orc_index *x, int64_t id, const float *vector, int level, int deleted) {
    if (slots_reserve(x, x->n_slots + 1) != 0)
        return -1;
    int s = x->n_slots;
    x->ids[s] = id;
    x->levels[s] = level;
    x->deleted[s] = (unsigned char)deleted;
    memcpy(x->vectors + (size_t)s * x->dim, vector, (size_t)x->dim * sizeof(float));
    x->norms[s] = vec_norm(x, vector);
    x->nb[s] = (nlist *)calloc((size_t)(level + 1), sizeof(nlist));
    if (!x->nb[s])
        return -1;
    x->n_slots++;
    if (ht_put(x->ht, x->ht_cap, x->ids, s) != 0) {
        x->n_slots--;
        free(x->nb[s]);
        return -1;
    }
    return s;
}

/* src/hnsw_algo.c:142-163 */
static int nl_add(orc_index *x, int slot, int level, int nbr) {
    if (level > x->levels[slot])
        return -1;
    nlist *L = &x->nb[slot][level];
    for (int i = 0; i < L->n; i++)
        if (L->v[i] == nbr)
            return 0;
    if (L->n >= L->cap) {
        int nc = L->cap == 0 ? 8 : L->cap * 2;
        int *nv = (int *)realloc(L->v, (size_t)nc * sizeof(int));
        if (!nv)
            return -1;
        L->v = nv;
        L->cap = nc;
    }
    L->v[L->n++] = nbr;
    return 0;
}

/* src/hnsw_algo.c:166-177 — swap with last */
static void nl_remove(orc_index *x, int slot, int level, int nbr) {
    if (level > x->levels[slot])
        return;
    nlist *L = &x->nb[slot][level];
    for (int i = 0; i < L->n; i++)
        if (L->v[i] == nbr) {
            L->v[i] = L->v[L->n - 1];
            L->n--;
            return;
        }
}

/* ───────────────────────── a7: greedy descent ───────────────────────── */

/* src/hnsw_algo.c:257-282.  Note the reference keeps iterating with index i after `current`
 * has been re-pointed, i.e. it continues at position i+1 of the NEW node's list; restated
 * literally. */
static int greedy_layer(orc_index *x, const float *q, float qn, int entry, int level) {
    int cur = entry;
    float cur_d = node_dist(x, q, qn, cur);
    int changed = 1;
    while (changed) {
        changed = 0;
        x->st.n_expanded++;
        for (int i = 0; i < x->nb[cur][level].n; i++) {
            int nb = x->nb[cur][level].v[i];
            if (x->deleted[nb])
                continue;
            float d = node_dist(x, q, qn, nb);
            if (d < cur_d) {
                cur_d = d;
                cur = nb;
                changed = 1;
                x->st.n_expanded++;
            }
        }
    }
    return cur;
}

/* ───────────────────────── a8: beam search ───────────────────────── */

typedef struct {
    orc_index *x;
    int64_t *lin;
    int lin_n, lin_cap;
    int count;
} vset;

static void vs_init(vset *v, orc_index *x, int cap) {
    v->x = x;
    v->lin = NULL;
    v->lin_n = 0;
    v->lin_cap = 0;
    v->count = 0;
    if (x->visited_mode == ORC_VISITED_LINEAR) {
        v->lin_cap = cap;
        v->lin = (int64_t *)malloc((size_t)cap * sizeof(int64_t));
    } else {
        if (x->vstamp_cap < x->n_slots) {
            int nc = x->cap_slots;
            x->vstamp = (unsigned *)realloc(x->vstamp, (size_t)nc * sizeof(unsigned));
            memset(x->vstamp + x->vstamp_cap, 0, (size_t)(nc - x->vstamp_cap) * sizeof(unsigned));
            x->vstamp_cap = nc;
        }
        x->vepoch++;
        if (x->vepoch == 0) {
            memset(x->vstamp, 0, (size_t)x->vstamp_cap * sizeof(unsigned));
            x->vepoch = 1;
        }
    }
}

static int vs_contains(const vset *v, int slot) {
    if (v->lin) {
        for (int i = 0; i < v->lin_n; i++) /* src/hnsw_algo.c:318-325 */
            if (v->lin[i] == slot)
                return 1;
        return 0;
    }
    return v->x->vstamp[slot] == v->x->vepoch;
}

static void vs_add(vset *v, int slot) {
    if (v->lin) {
        if (v->lin_n >= v->lin_cap) {
            v->lin_cap *= 2;
            v->lin = (int64_t *)realloc(v->lin, (size_t)v->lin_cap * sizeof(int64_t));
        }
        v->lin[v->lin_n++] = slot;
    } else {
        v->x->vstamp[slot] = v->x->vepoch;
    }
    v->count++;
}

/* src/hnsw_algo.c:347-448.  One entry point (every call site passes entry_count == 1). */
static int beam_layer(orc_index *x, const float *q, float qn, int entry, int level, int ef, int *res_slots,
                      float *res_dists) {
    orc_pq cand, res;
    vset vis;
    orc_pq_init(&cand, ef * 2);
    orc_pq_init(&res, ef * 2);
    vs_init(&vis, x, ef * 4);

    if (!x->deleted[entry]) { /* :360 */
        float d = node_dist(x, q, qn, entry);
        orc_pq_push(&cand, entry, d);
        orc_pq_push(&res, entry, -d);
        vs_add(&vis, entry);
    }

    int patience_max = ef / 4; /* :372-375 */
    if (patience_max < 10)
        patience_max = 10;
    int stale = 0;

    while (cand.size > 0) {
        orc_pq_item c = orc_pq_pop(&cand);
        if (res.size >= ef) { /* :382-386 */
            float worst = -res.items[1].distance;
            if (c.distance > worst)
                break;
        }
        if (stale >= patience_max && res.size >= ef) /* :391 */
            break;
        int node = (int)c.id;
        int improved = 0;
        x->st.n_expanded++;
        nlist *L = &x->nb[node][level];
        for (int i = 0; i < L->n; i++) { /* :401-426 */
            int nb = L->v[i];
            if (vs_contains(&vis, nb))
                continue;
            vs_add(&vis, nb);
            if (x->deleted[nb])
                continue;
            float d = node_dist(x, q, qn, nb);
            if (res.size < ef) {
                orc_pq_push(&cand, nb, d);
                orc_pq_push(&res, nb, -d);
                improved = 1;
            } else {
                float worst = -res.items[1].distance;
                if (d < worst) {
                    orc_pq_push(&cand, nb, d);
                    orc_pq_pop(&res);
                    orc_pq_push(&res, nb, -d);
                    improved = 1;
                }
            }
            if (cand.size > x->st.max_cand)
                x->st.max_cand = cand.size;
        }
        stale = improved ? 0 : stale + 1; /* :428-432 */
    }
    if (vis.count > x->st.max_visited)
        x->st.max_visited = vis.count;

    int count = res.size; /* :436-441 */
    for (int i = count - 1; i >= 0; i--) {
        orc_pq_item it = orc_pq_pop(&res);
        res_slots[i] = (int)it.id;
        res_dists[i] = -it.distance;
    }
    orc_pq_destroy(&cand);
    orc_pq_destroy(&res);
    free(vis.lin);
    return count;
}

/* ───────────────────────── a10: MN-RU prune ───────────────────────── */

/* src/hnsw_algo.c:460-475, with b's list passed explicitly so the batch schedule can hand in a
 * snapshot */
static int mutual_count(const int *a, int na, const int *b, int nb) {
    int c = 0;
    for (int i = 0; i < na; i++)
        for (int j = 0; j < nb; j++)
            if (b[j] == a[i]) {
                c++;
                break;
            }
    return c;
}

/* The same number — |list ∩ b|, lists hold no duplicates — with the list's members stamped once per prune:
 * O(|b|) instead of O(|list|·|b|).  ORC_FAITHFUL_MN=1 (environment) switches back to the scan above. */
static int mutual_count_stamped(const orc_index *x, const int *b, int nb) {
    int c = 0;
    for (int j = 0; j < nb; j++)
        c += x->mstamp[b[j]] == x->mepoch;
    return c;
}

typedef struct {
    const int *v;
    int n;
    int valid; /* 0 → level > node level → MN = 0 (:461-462) */
} nview;

typedef nview (*nview_fn)(void *ctx, int slot, int level);

static nview live_view(void *ctx, int slot, int level) {
    orc_index *x = (orc_index *)ctx;
    nview r = {NULL, 0, 0};
    if (level <= x->levels[slot]) {
        r.v = x->nb[slot][level].v;
        r.n = x->nb[slot][level].n;
        r.valid = 1;
    }
    return r;
}

/* src/hnsw_algo.c:601-646 — list (nc entries) of node `t` pruned in place to M_max entries.
 * `view` supplies the neighbour lists MN is counted against. */
static void prune_list(orc_index *x, int t, int level, int *list, int nc, int M_max, nview_fn view, void *vctx) {
    float *nd = (float *)malloc((size_t)nc * sizeof(float));
    int *mn = (int *)malloc((size_t)nc * sizeof(int));
    int *cp = (int *)malloc((size_t)nc * sizeof(int));
    const float *tv = x->vectors + (size_t)t * x->dim;
    float tn = x->norms[t];
    memcpy(cp, list, (size_t)nc * sizeof(int));
    x->st.n_prune++;
    static int faithful = -1;
    if (faithful < 0) {
        const char *e = getenv("ORC_FAITHFUL_MN");
        faithful = e && atoi(e) != 0;
    }
    if (!faithful) {
        if (x->mstamp_cap < x->n_slots) {
            int nc2 = x->cap_slots > x->n_slots ? x->cap_slots : x->n_slots;
            x->mstamp = (unsigned *)realloc(x->mstamp, (size_t)nc2 * sizeof(unsigned));
            memset(x->mstamp + x->mstamp_cap, 0, (size_t)(nc2 - x->mstamp_cap) * sizeof(unsigned));
            x->mstamp_cap = nc2;
        }
        if (++x->mepoch == 0) { /* wrapped: start over */
            memset(x->mstamp, 0, (size_t)x->mstamp_cap * sizeof(unsigned));
            x->mepoch = 1;
        }
        for (int j = 0; j < nc; j++)
            x->mstamp[list[j]] = x->mepoch;
    }
    for (int j = 0; j < nc; j++) {
        int nn = cp[j];
        if (x->deleted[nn]) { /* :610-612 */
            nd[j] = 1e30f;
            mn[j] = -1;
        } else {
            nd[j] = node_dist(x, tv, tn, nn);
            nview b = view(vctx, nn, level);
            mn[j] = !b.valid ? 0 : faithful ? mutual_count(list, nc, b.v, b.n) : mutual_count_stamped(x, b.v, b.n);
        }
    }
    for (int a = 0; a < M_max && a < nc; a++) { /* :620-639 */
        int best = a;
        for (int b = a + 1; b < nc; b++)
            if (nd[b] < nd[best] || (nd[b] == nd[best] && mn[b] > mn[best]))
                best = b;
        if (best != a) {
            float td = nd[a];
            nd[a] = nd[best];
            nd[best] = td;
            int tm = mn[a];
            mn[a] = mn[best];
            mn[best] = tm;
            int ti = cp[a];
            cp[a] = cp[best];
            cp[best] = ti;
        }
    }
    memcpy(list, cp, (size_t)M_max * sizeof(int)); /* :640-641 */
    free(nd);
    free(mn);
    free(cp);
}

/* ───────────────────────── a10: insert ───────────────────────── */

static int insert_prologue(orc_index *x, int64_t id, const float *vector, int *out_level) {
    if (ht_find(x, id) >= 0) /* :522 */
        return -1;
    if (x->node_count * 10 > x->ht_cap * 7) /* :527 */
        if (ht_grow(x) != 0)
            return -1;
    int level = orc_hnsw_random_level(x); /* :532 */
    int s = node_new(x, id, vector, level, 0);
    if (s < 0)
        return -1;
    x->node_count++;
    *out_level = level;
    return s;
}

int orc_hnsw_insert(orc_index *x, int64_t id, const float *vector) {
    int level;
    int s = insert_prologue(x, id, vector, &level);
    if (s < 0)
        return -1;
    if (x->entry_id == -1) { /* :544-548 */
        x->entry_id = id;
        x->max_level = level;
        return 0;
    }
    const float *q = x->vectors + (size_t)s * x->dim;
    float qn = x->norms[s];
    int cur = ht_find(x, x->entry_id);
    for (int l = x->max_level; l > level; l--) /* :553-555 */
        cur = greedy_layer(x, q, qn, cur, l);

    int start = level < x->max_level ? level : x->max_level;
    int ef = x->efc;
    int *rs = (int *)malloc((size_t)ef * sizeof(int));
    float *rd = (float *)malloc((size_t)ef * sizeof(float));
    for (int l = start; l >= 0; l--) { /* :572-653 */
        int M_max = (l == 0) ? x->M_max0 : x->M;
        int found = beam_layer(x, q, qn, cur, l, ef, rs, rd);
        int nsel = found < M_max ? found : M_max; /* :511 */
        for (int i = 0; i < nsel; i++) {
            int nb = rs[i];
            nl_add(x, s, l, nb);
            if (l <= x->levels[nb]) { /* :590 */
                nl_add(x, nb, l, s);
                nlist *L = &x->nb[nb][l];
                if (L->n > M_max) { /* :601 */
                    prune_list(x, nb, l, L->v, L->n, M_max, live_view, x);
                    L->n = M_max;
                }
            }
        }
        if (found > 0) /* :651-652 */
            cur = rs[0];
    }
    free(rs);
    free(rd);
    if (level > x->max_level) { /* :660-663 */
        x->entry_id = id;
        x->max_level = level;
    }
    return 0;
}

/* ───────────────────────── batch-synchronous build schedule ─────────────────────────
 * DESIGN.md §"fast build": (1) every node of the batch gets its level from the xorshift stream
 * in batch order and is appended unlinked; (2) each is searched (greedy descent + per-level beam,
 * ef_construction) against the graph as frozen at batch start, entry/max_level frozen too;
 * selected_l(j) = first min(found, M_max) results; (3) per level: forward lists are written,
 * then for every existing target t the sources that selected it are appended in batch order,
 * pruning (src/hnsw_algo.c:601-646) whenever the list exceeds M_max, with MN counted against
 * the lists as they stood after step (forward lists) and before any reverse edge of this batch;
 * (4) entry point / max_level are updated in batch order by the rule of :660-663. */
typedef struct {
    orc_index *x;
    int first_new; /* slots >= first_new are batch nodes */
    nlist *snap;   /* snapshot lists at the level being linked, indexed by slot (only touched targets filled) */
    unsigned char *has_snap;
} snap_ctx;

static nview snap_view(void *ctx, int slot, int level) {
    snap_ctx *c = (snap_ctx *)ctx;
    orc_index *x = c->x;
    nview r = {NULL, 0, 0};
    if (level > x->levels[slot])
        return r;
    r.valid = 1;
    if (slot < c->first_new && c->has_snap[slot]) {
        r.v = c->snap[slot].v;
        r.n = c->snap[slot].n;
    } else {
        r.v = x->nb[slot][level].v;
        r.n = x->nb[slot][level].n;
    }
    return r;
}

int orc_hnsw_insert_batch(orc_index *x, const int64_t *ids, const float *vectors, int n) {
    if (n <= 0)
        return 0;
    int first_new = x->n_slots;
    int bstart = 0;
    /* An empty index cannot be searched: the first node goes in alone (:544-548). */
    if (x->entry_id == -1) {
        if (orc_hnsw_insert(x, ids[0], vectors) != 0)
            return -1;
        bstart = 1;
        first_new = x->n_slots;
        if (n == 1)
            return 0;
    }
    int nb_ = n - bstart;
    int *slots = (int *)malloc((size_t)nb_ * sizeof(int));
    int *lv = (int *)malloc((size_t)nb_ * sizeof(int));
    for (int j = 0; j < nb_; j++) {
        slots[j] = insert_prologue(x, ids[bstart + j], vectors + (size_t)(bstart + j) * x->dim, &lv[j]);
        if (slots[j] < 0) {
            free(slots);
            free(lv);
            return -1;
        }
    }
    int fz_entry = ht_find(x, x->entry_id);
    int fz_max = x->max_level;
    int ef = x->efc;
    /* selected[j][l] lists */
    int nlev = fz_max + 1;
    int W = x->M_max0;
    int *sel = (int *)malloc((size_t)nb_ * nlev * W * sizeof(int));
    int *nsel = (int *)calloc((size_t)nb_ * nlev, sizeof(int));
    int *rs = (int *)malloc((size_t)ef * sizeof(int));
    float *rd = (float *)malloc((size_t)ef * sizeof(float));
    for (int j = 0; j < nb_; j++) {
        int s = slots[j];
        const float *q = x->vectors + (size_t)s * x->dim;
        float qn = x->norms[s];
        int cur = fz_entry;
        for (int l = fz_max; l > lv[j]; l--)
            cur = greedy_layer(x, q, qn, cur, l);
        int start = lv[j] < fz_max ? lv[j] : fz_max;
        for (int l = start; l >= 0; l--) {
            int M_max = (l == 0) ? x->M_max0 : x->M;
            int found = beam_layer(x, q, qn, cur, l, ef, rs, rd);
            int k = found < M_max ? found : M_max;
            nsel[j * nlev + l] = k;
            memcpy(sel + ((size_t)j * nlev + l) * W, rs, (size_t)k * sizeof(int));
            if (found > 0)
                cur = rs[0];
        }
    }
    free(rs);
    free(rd);
    /* link, level by level */
    snap_ctx sc;
    sc.x = x;
    sc.first_new = first_new;
    sc.snap = (nlist *)calloc((size_t)first_new, sizeof(nlist));
    sc.has_snap = (unsigned char *)calloc((size_t)first_new, 1);
    for (int l = 0; l < nlev; l++) {
        int M_max = (l == 0) ? x->M_max0 : x->M;
        /* forward lists */
        for (int j = 0; j < nb_; j++)
            for (int i = 0; i < nsel[j * nlev + l]; i++)
                nl_add(x, slots[j], l, sel[((size_t)j * nlev + l) * W + i]);
        /* snapshot every touched target before any reverse edge lands */
        for (int j = 0; j < nb_; j++)
            for (int i = 0; i < nsel[j * nlev + l]; i++) {
                int t = sel[((size_t)j * nlev + l) * W + i];
                if (!sc.has_snap[t]) {
                    nlist *L = &x->nb[t][l];
                    sc.snap[t].n = L->n;
                    sc.snap[t].v = (int *)malloc((size_t)(L->n + 1) * sizeof(int));
                    memcpy(sc.snap[t].v, L->v, (size_t)L->n * sizeof(int));
                    sc.has_snap[t] = 1;
                }
            }
        /* reverse edges: iterating j in batch order visits each target's sources in batch order,
         * and targets are independent of one another because MN reads only snapshots */
        for (int j = 0; j < nb_; j++)
            for (int i = 0; i < nsel[j * nlev + l]; i++) {
                int t = sel[((size_t)j * nlev + l) * W + i];
                if (l > x->levels[t])
                    continue;
                nl_add(x, t, l, slots[j]);
                nlist *L = &x->nb[t][l];
                if (L->n > M_max) {
                    prune_list(x, t, l, L->v, L->n, M_max, snap_view, &sc);
                    L->n = M_max;
                }
            }
        for (int j = 0; j < nb_; j++)
            for (int i = 0; i < nsel[j * nlev + l]; i++) {
                int t = sel[((size_t)j * nlev + l) * W + i];
                if (sc.has_snap[t]) {
                    free(sc.snap[t].v);
                    sc.snap[t].v = NULL;
                    sc.has_snap[t] = 0;
                }
            }
    }
    free(sc.snap);
    free(sc.has_snap);
    for (int j = 0; j < nb_; j++)
        if (lv[j] > x->max_level) {
            x->entry_id = x->ids[slots[j]];
            x->max_level = lv[j];
        }
    free(sel);
    free(nsel);
    free(slots);
    free(lv);
    return 0;
}

/* ───────────────────────── a11: search ───────────────────────── */

int orc_hnsw_search(orc_index *x, const float *query, int k, int ef, orc_result *out) {
    if (x->entry_id == -1 || x->node_count == 0) /* :671 */
        return 0;
    if (ef < k)
        ef = k;
    float qn = vec_norm(x, query);
    int cur = ht_find(x, x->entry_id);
    for (int l = x->max_level; l > 0; l--)
        cur = greedy_layer(x, query, qn, cur, l);
    int *rs = (int *)malloc((size_t)ef * sizeof(int));
    float *rd = (float *)malloc((size_t)ef * sizeof(float));
    int found = beam_layer(x, query, qn, cur, 0, ef, rs, rd);
    int count = found < k ? found : k;
    for (int i = 0; i < count; i++) {
        out[i].id = x->ids[rs[i]];
        out[i].distance = rd[i];
    }
    free(rs);
    free(rd);
    return count;
}

/* ───────────────────────── a12: delete ───────────────────────── */

/* src/hnsw_algo.c:717-805 */
int orc_hnsw_delete(orc_index *x, int64_t id) {
    int s = ht_find(x, id);
    if (s < 0 || x->deleted[s])
        return -1;
    x->deleted[s] = 1;
    x->node_count--;
    int min_conn = x->M / 2;
    for (int l = 0; l <= x->levels[s]; l++) {
        int nc = x->nb[s][l].n;
        int *former = NULL;
        if (nc > 0) {
            former = (int *)malloc((size_t)nc * sizeof(int));
            memcpy(former, x->nb[s][l].v, (size_t)nc * sizeof(int));
        }
        for (int i = 0; i < nc; i++) { /* :741-746 */
            int nb = x->nb[s][l].v[i];
            if (!x->deleted[nb])
                nl_remove(x, nb, l, s);
        }
        if (former) { /* :750-786 */
            for (int i = 0; i < nc; i++) {
                int orphan = former[i];
                if (x->deleted[orphan] || l > x->levels[orphan])
                    continue;
                if (x->nb[orphan][l].n >= min_conn)
                    continue;
                for (int j = 0; j < nc && x->nb[orphan][l].n < min_conn; j++) {
                    if (i == j)
                        continue;
                    int cand = former[j];
                    if (x->deleted[cand] || l > x->levels[cand])
                        continue;
                    int already = 0;
                    for (int k = 0; k < x->nb[orphan][l].n; k++)
                        if (x->nb[orphan][l].v[k] == cand) {
                            already = 1;
                            break;
                        }
                    if (!already) {
                        nl_add(x, orphan, l, cand);
                        nl_add(x, cand, l, orphan);
                    }
                }
            }
            free(former);
        }
    }
    if (x->entry_id == id) { /* :790-802 — scan in hash-table order, strict > */
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
    return 0;
}

/* ───────────────────────── inspection / loading ───────────────────────── */

int orc_hnsw_node_count(const orc_index *x) {
    return x->node_count;
}
int64_t orc_hnsw_entry_point(const orc_index *x) {
    return x->entry_id;
}
int orc_hnsw_max_level(const orc_index *x) {
    return x->max_level;
}
int orc_hnsw_node_level(const orc_index *x, int64_t id) {
    int s = ht_find(x, id);
    return s < 0 ? -1 : x->levels[s];
}
int orc_hnsw_node_deleted(const orc_index *x, int64_t id) {
    int s = ht_find(x, id);
    return s < 0 ? -1 : x->deleted[s];
}
int orc_hnsw_neighbors(const orc_index *x, int64_t id, int level, int64_t *out, int cap) {
    int s = ht_find(x, id);
    if (s < 0 || level > x->levels[s])
        return -1;
    const nlist *L = &x->nb[s][level];
    for (int i = 0; i < L->n && i < cap; i++)
        out[i] = x->ids[L->v[i]];
    return L->n;
}
const float *orc_hnsw_vector(const orc_index *x, int64_t id) {
    int s = ht_find(x, id);
    if (s < 0 || x->deleted[s])
        return NULL;
    return x->vectors + (size_t)s * x->dim;
}
void orc_hnsw_get_stats(const orc_index *x, orc_stats *out) {
    *out = x->st;
}
void orc_hnsw_reset_stats(orc_index *x) {
    memset(&x->st, 0, sizeof(x->st));
}

/* mirrors load_index_from_shadow's node loop (src/hnsw_vtab.c:297-320) */
int orc_hnsw_load_node(orc_index *x, int64_t id, const float *vector, int level, int deleted) {
    if (x->node_count * 10 > x->ht_cap * 7)
        ht_grow(x);
    int s = node_new(x, id, vector, level, deleted);
    if (s < 0)
        return -1;
    if (!deleted)
        x->node_count++;
    return 0;
}

int orc_hnsw_load_neighbors(orc_index *x, int64_t id, int level, const int64_t *nbrs, int n) {
    int s = ht_find(x, id);
    if (s < 0 || level > x->levels[s])
        return -1;
    for (int i = 0; i < n; i++) {
        int t = ht_find(x, nbrs[i]);
        if (t >= 0)
            nl_add(x, s, level, t);
    }
    return 0;
}

void orc_hnsw_set_entry(orc_index *x, int64_t entry, int max_level) {
    x->entry_id = entry;
    x->max_level = max_level;
}

int orc_hnsw_load_bulk(orc_index *x, int n, const int64_t *ids, const float *vectors, const int *levels,
                       const int *deleted) {
    for (int i = 0; i < n; i++)
        if (orc_hnsw_load_node(x, ids[i], vectors + (size_t)i * x->dim, levels[i], deleted[i]) != 0)
            return -1;
    return 0;
}

int orc_hnsw_load_links_bulk(orc_index *x, int level, const int *rows, int width) {
    for (int s = 0; s < x->n_slots; s++) {
        if (level > x->levels[s])
            continue;
        for (int i = 0; i < width; i++) {
            int t = rows[(size_t)s * width + i];
            if (t < 0)
                break;
            nl_add(x, s, level, t);
        }
    }
    return 0;
}
