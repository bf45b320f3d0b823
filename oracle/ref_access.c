/*
 * ref_access.c — graph accessors compiled INTO oracle/_ref/libmuninn_ref.so next to the
 * reference's own sources (see oracle/Makefile).  TEST INFRASTRUCTURE ONLY.
 *
 * The reference's HnswIndex/HnswNode are plain structs (src/hnsw_algo.h:17-53); ctypes callers
 * would have to mirror their layout, so these few functions read them on the C side instead.
 * Nothing here restates reference logic: it only walks the reference's own data structures
 * through the reference's own header and exported functions (ht_find, src/hnsw_algo.h:98).
 */
#include <stddef.h>
#include "hnsw_algo.h" /* resolved with -I/root/reference/src */

int ref_node_count(HnswIndex *idx) {
    return idx->node_count;
}
int64_t ref_entry_point(HnswIndex *idx) {
    return idx->entry_point;
}
int ref_max_level(HnswIndex *idx) {
    return idx->max_level;
}
int ref_node_level(HnswIndex *idx, int64_t id) {
    HnswNode *n = ht_find(idx->nodes, idx->node_capacity, id);
    return n ? n->level : -1;
}
int ref_node_deleted(HnswIndex *idx, int64_t id) {
    HnswNode *n = ht_find(idx->nodes, idx->node_capacity, id);
    return n ? n->deleted : -1;
}
int ref_neighbors(HnswIndex *idx, int64_t id, int level, int64_t *out, int cap) {
    HnswNode *n = ht_find(idx->nodes, idx->node_capacity, id);
    if (!n || level > n->level)
        return -1;
    for (int i = 0; i < n->neighbor_count[level] && i < cap; i++)
        out[i] = n->neighbors[level][i];
    return n->neighbor_count[level];
}
unsigned ref_rng_state(HnswIndex *idx) {
    return idx->rng_state;
}
/* distance through the reference's own dispatch (src/vec_math.c:180-190) */
float ref_distance(int metric, const float *a, const float *b, int dim) {
    return vec_get_distance_func((VecMetric)metric)(a, b, dim);
}
void ref_dist_batch(int metric, const float *q, const float *rows, int64_t n, int dim, float *out) {
    VecDistanceFunc f = vec_get_distance_func((VecMetric)metric);
    for (int64_t i = 0; i < n; i++)
        out[i] = f(q, rows + (size_t)i * dim, dim);
}
/* batch helpers so Python does not pay a ctypes call per vector */
int ref_insert_many(HnswIndex *idx, const int64_t *ids, const float *vecs, int n) {
    for (int i = 0; i < n; i++)
        if (hnsw_insert(idx, ids[i], vecs + (size_t)i * idx->dim) != 0)
            return i;
    return n;
}
int ref_search_many(HnswIndex *idx, const float *queries, int nq, int k, int ef, int64_t *out_ids, float *out_dists,
                    int *out_counts) {
    HnswSearchResult r[1024];
    if (k > 1024)
        return -1;
    for (int q = 0; q < nq; q++) {
        int c = hnsw_search(idx, queries + (size_t)q * idx->dim, k, ef, r);
        out_counts[q] = c;
        for (int i = 0; i < k; i++) {
            out_ids[(size_t)q * k + i] = i < c ? r[i].id : -1;
            out_dists[(size_t)q * k + i] = i < c ? r[i].distance : 0.0f;
        }
    }
    return 0;
}

/* Bulk load of an existing graph into a reference index, through the reference's own load API
 * (the functions src/hnsw_algo.h:98-104 exports for its shadow-table loader; call sequence as in
 * src/hnsw_vtab.c:297-338).  Lets bench.py time the reference's hnsw_search on the very graph the
 * GPU built, without paying the reference's multi-hour 1M-vector build. */
int ref_load_nodes(HnswIndex *idx, int64_t n, const int64_t *ids, const float *vecs, const int *levels,
                   const unsigned char *deleted, int64_t entry_point, int max_level) {
    for (int64_t i = 0; i < n; i++) {
        if ((int64_t)idx->node_count * 10 > (int64_t)idx->node_capacity * 7)
            ht_resize(idx);
        HnswNode *nd = node_create(ids[i], vecs + (size_t)i * idx->dim, idx->dim, levels[i]);
        if (!nd)
            return -1;
        nd->deleted = deleted ? deleted[i] : 0;
        ht_insert(idx->nodes, idx->node_capacity, nd);
        if (!nd->deleted)
            idx->node_count++;
    }
    idx->entry_point = entry_point;
    idx->max_level = max_level;
    return 0;
}
int ref_load_edges(HnswIndex *idx, int64_t n_edges, const int64_t *src, const int64_t *dst, const int *level) {
    for (int64_t e = 0; e < n_edges; e++) {
        HnswNode *nd = ht_find(idx->nodes, idx->node_capacity, src[e]);
        if (nd && level[e] <= nd->level)
            node_add_neighbor(nd, level[e], dst[e]);
    }
    return 0;
}
