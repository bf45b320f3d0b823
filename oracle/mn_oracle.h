/*
 * mn_oracle.h — CPU oracle for the sqlite-muninn hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This directory is a plain-C restatement of the reference's algorithm for the
 * path BASELINE.json:north_star names.  It is the checker, never the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link,
 * load or call anything in oracle/.  The product path (sqlite-muninn_amd/csrc)
 * never includes this header.
 *
 * Parity pinning: oracle/Makefile compiles the reference's own C sources
 * (where they lie under /root/reference/src) into oracle/_ref/, and
 * tests/test_oracle_vs_ref.py + oracle/gen_golden.py check this restatement
 * bit-for-bit against it; the resulting vectors are committed in tests/golden/.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference repository root).
 */
#ifndef MN_ORACLE_H
#define MN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/vec_math.h:13 — the integer values are persisted in "{t}_config" */
enum { ORC_METRIC_L2 = 0, ORC_METRIC_COSINE = 1, ORC_METRIC_IP = 2 };

/* Summation order of the distance inner loop.
 *   ORC_ORDER_SSE  — the reference's x86 order (src/vec_math.c:78-143): four lane
 *                    accumulators over i mod 4, separate mul and add, ((t0+t1)+t2)+t3,
 *                    then the scalar tail.  Bit-exact to the compiled reference.
 *   ORC_ORDER_WAVE — the order of the HIP "fast" kernel (one wavefront per row,
 *                    float4 per lane, fmaf chain per lane, xor-butterfly across the
 *                    64 lanes).  Used to check the fast kernel bit-for-bit; differs
 *                    from the reference by ~1e-7 relative.
 */
enum { ORC_ORDER_SSE = 0, ORC_ORDER_WAVE = 1 };

/* Visited-set implementation of the beam search (results identical, cost differs):
 *   ORC_VISITED_BITMAP — epoch-stamped array, O(1)
 *   ORC_VISITED_LINEAR — the reference's linear scan (src/hnsw_algo.c:318-325), kept so the
 *                        "faithful" CPU baseline has the reference's real cost profile. */
enum { ORC_VISITED_BITMAP = 0, ORC_VISITED_LINEAR = 1 };

/* ---- a1-a3: distances (src/vec_math.c) ---- */
float orc_vec_l2(const float *a, const float *b, int dim);
float orc_vec_cosine(const float *a, const float *b, int dim);
float orc_vec_ip(const float *a, const float *b, int dim);
float orc_vec_distance(int metric, int order, const float *a, const float *b, int dim);
int orc_vec_parse_metric(const char *name, int *out); /* src/vec_math.c:192-204 */
/* out[i] = distance(query, rows[i]) */
void orc_dist_batch(int metric, int order, const float *query, const float *rows, int64_t n, int dim, float *out);

/* ---- a13: 1-based binary min-heap (src/priority_queue.c) ---- */
typedef struct {
    int64_t id;
    float distance;
} orc_pq_item;
typedef struct {
    orc_pq_item *items;
    int size, capacity;
} orc_pq;
int orc_pq_init(orc_pq *pq, int cap);
int orc_pq_push(orc_pq *pq, int64_t id, float distance);
orc_pq_item orc_pq_pop(orc_pq *pq);
void orc_pq_destroy(orc_pq *pq);
/* Replays a push/pop trace: ops[i] = 1 push (ids[i], dists[i]) / 0 pop.  Writes the popped
 * (id, distance) pairs in order; returns the number of pops. */
int orc_pq_trace(const int *ops, const int64_t *ids, const float *dists, int n, int64_t *out_ids, float *out_dists);

/* ---- a5-a12: HNSW (src/hnsw_algo.c) ---- */
typedef struct orc_index orc_index;

typedef struct {
    int64_t id;
    float distance;
} orc_result; /* src/hnsw_algo.h:30-33 */

typedef struct {
    int64_t n_dist;     /* distance evaluations (dist_func calls) */
    int64_t n_expanded; /* neighbour rows read by beam/greedy expansion */
    int64_t n_prune;    /* MN-RU prunes */
    int64_t max_cand;   /* high-water mark of the candidates heap */
    int64_t max_visited;
} orc_stats;

orc_index *orc_hnsw_create(int dim, int metric, int M, int ef_construction); /* :181-208 */
void orc_hnsw_destroy(orc_index *idx);
void orc_hnsw_seed_rng(orc_index *idx, unsigned seed);        /* :222-224 */
void orc_hnsw_set_order(orc_index *idx, int order);           /* ORC_ORDER_* */
void orc_hnsw_set_visited(orc_index *idx, int visited_mode);  /* ORC_VISITED_* */
int orc_hnsw_insert(orc_index *idx, int64_t id, const float *vector);        /* :520-666 */
int orc_hnsw_search(orc_index *idx, const float *query, int k, int ef, orc_result *out); /* :670-704 */
int orc_hnsw_delete(orc_index *idx, int64_t id);                               /* :717-805 */
/* Batch-synchronous build schedule of the HIP "fast" build: every node of the
 * batch is searched against the graph frozen at batch start, then linked in batch order.  With
 * n == 1 this is exactly orc_hnsw_insert. */
int orc_hnsw_insert_batch(orc_index *idx, const int64_t *ids, const float *vectors, int n);
/* Draw the next level from the index's xorshift32 stream (:240-248) without inserting. */
int orc_hnsw_random_level(orc_index *idx);

/* graph inspection (for parity checks) */
int orc_hnsw_node_count(const orc_index *idx);
int64_t orc_hnsw_entry_point(const orc_index *idx);
int orc_hnsw_max_level(const orc_index *idx);
int orc_hnsw_node_level(const orc_index *idx, int64_t id);                 /* -1 if absent */
int orc_hnsw_node_deleted(const orc_index *idx, int64_t id);
int orc_hnsw_neighbors(const orc_index *idx, int64_t id, int level, int64_t *out, int cap);
const float *orc_hnsw_vector(const orc_index *idx, int64_t id);            /* NULL if absent/deleted */
void orc_hnsw_get_stats(const orc_index *idx, orc_stats *out);
void orc_hnsw_reset_stats(orc_index *idx);
/* load a graph wholesale (used to hand the oracle the graph a device build produced) */
int orc_hnsw_load_node(orc_index *idx, int64_t id, const float *vector, int level, int deleted);
int orc_hnsw_load_neighbors(orc_index *idx, int64_t id, int level, const int64_t *nbrs, int n);
void orc_hnsw_set_entry(orc_index *idx, int64_t entry, int max_level);
/* bulk variants: nodes in slot order; link rows are slot indices, -1 padded, [n][width] */
int orc_hnsw_load_bulk(orc_index *idx, int n, const int64_t *ids, const float *vectors, const int *levels,
                       const int *deleted);
int orc_hnsw_load_links_bulk(orc_index *idx, int level, const int *rows, int width);

#ifdef __cplusplus
}
#endif
#endif
