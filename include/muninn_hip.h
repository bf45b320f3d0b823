/*
 * muninn_hip.h — C-ABI of libmuninn_hip.so, the MI355X (gfx950) implementation of sqlite-muninn's
 * compute hot path.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * Each entry point names the reference interface it replaces (file:line in the reference repo).
 * The SQLite glue (hnsw_vtab.c, node2vec.c, graph_community.c) binds to these instead of to
 * hnsw_algo.c / vec_math.c; INTEGRATION.md shows the binding.
 *
 * Threading contract (as the reference's, SURVEY §8b "Threading"): one host thread per index at a
 * time; several indexes per process are fine (each owns a HIP stream).
 *
 * Error convention follows src/hnsw_algo.h:55-79: create → NULL on failure; insert/delete → 0 / -1;
 * search → result count.  mn_last_error() returns a thread-local message for the last failure.
 * There is NO CPU fallback: if no gfx950 device is usable every compute entry point fails.
 */
#ifndef MUNINN_HIP_H
#define MUNINN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MN_ABI_VERSION 2

/* src/vec_math.h:13 — values are persisted in "{table}_config" (src/hnsw_vtab.c:189) */
typedef enum { MN_METRIC_L2 = 0, MN_METRIC_COSINE = 1, MN_METRIC_INNER_PRODUCT = 2 } mn_metric;

/* Summation order of the distance inner loop:
 *   MN_ORDER_SSE  — bit-exact to the reference's x86 build (src/vec_math.c:78-143)
 *   MN_ORDER_WAVE — wavefront-native order (coalesced float4 per lane + xor butterfly); within
 *                   ~1e-7 relative of the reference, faster */
typedef enum { MN_ORDER_SSE = 0, MN_ORDER_WAVE = 1 } mn_order;

/* Build schedule for mn_hnsw_insert_batch */
typedef enum {
    MN_BUILD_SEQUENTIAL = 0, /* one node at a time: graph bit-identical to hnsw_insert called in a loop */
    MN_BUILD_BATCHED = 1     /* batch-synchronous: nodes of one call are searched against the graph frozen
                                at call start, then linked in order */
} mn_build_mode;

/* src/hnsw_algo.h:30-33 */
typedef struct {
    int64_t id;
    float distance;
} mn_search_result;

typedef struct mn_index mn_index; /* replaces HnswIndex (src/hnsw_algo.h:36-53); device-resident */

/* ---- library ---- */
int mn_abi_version(void);
const char *mn_last_error(void);
/* number of usable gfx950 devices (0 → every compute call fails) */
int mn_device_count(void);
/* Error convention under memory pressure (src/hnsw_algo.h:55-79: hnsw_create → NULL, hnsw_insert / hnsw_delete → -1,
 * hnsw_search → 0): no C++ exception ever leaves this library.  Every entry point that can allocate is closed by an exception
 * barrier (csrc/mn_guard.hpp) which maps std::bad_alloc and friends to the function's error value and names the failure in the
 * thread's last-error string; a call that may have left its handle half-edited marks the handle unusable, later calls on it
 * fail cleanly.  Test hook: the nth host allocation this library makes from now on throws std::bad_alloc (0 disarms);
 * returns the number of host allocations counted since the previous call. */
long long mn_debug_fault_alloc(long long nth);

/* ---- vec_math.c replacements (a1-a4) ---- */
/* src/vec_math.c:192-204: "l2" | "cosine" | "inner_product" → 0, else -1 */
int mn_vec_parse_metric(const char *name, int *out_metric);
/* vec_get_distance_func(metric)(query, rows[i], dim) for i < n  (src/vec_math.c:78-143,180-190).
 * Host pointers; rows is [n][dim] row-major.  Returns 0 / -1. */
int mn_vec_dist_batch(int metric, int order, const float *query, const float *rows, int64_t n, int dim, float *out);

/* ---- hnsw_algo.c replacements (a5-a12) ---- */
/* hnsw_create (src/hnsw_algo.c:181-208).  M_max0 = 2M, rng seed 42.  device = HIP ordinal.
 * 2 <= M <= 512 (the reference has no upper bound; here a list of 2M + 1 entries is pruned inside one workgroup's LDS;
 * rows of more than 64 links are walked 64 at a time); NULL + mn_last_error() otherwise, or when no gfx950 device is
 * available (there is no CPU fallback).  The first index a process creates on a device also has HIP load the library's kernels
 * there (tens of milliseconds, once), so that no later query or insert pays for it. */
mn_index *mn_hnsw_create(int dim, int metric, int M, int ef_construction);
mn_index *mn_hnsw_create_on(int dim, int metric, int M, int ef_construction, int device);
/* hnsw_destroy (:210-220) */
void mn_hnsw_destroy(mn_index *idx);
/* hnsw_seed_rng (:222-224) */
void mn_hnsw_seed_rng(mn_index *idx, unsigned seed);
/* distance summation order for this index; must be set before the first insert (default SSE) */
int mn_hnsw_set_order(mn_index *idx, int order);

/* hnsw_insert (:520-666): vector is copied.  0, or -1 on duplicate id / failure — including the reference's "table full"
 * failure (:61-74): its node table grows on the LIVE count while soft-deleted nodes keep their entries, so delete + insert
 * churn can fill it; as there, the level draw is consumed and nothing else changes. */
int mn_hnsw_insert(mn_index *idx, int64_t id, const float *vector);
/* n inserts in one call.  mode MN_BUILD_SEQUENTIAL ≡ n × mn_hnsw_insert; MN_BUILD_BATCHED is the
 * batch-synchronous schedule.  vectors is host [n][dim].  Returns 0 / -1 (nothing inserted on -1). */
int mn_hnsw_insert_batch(mn_index *idx, const int64_t *ids, const float *vectors, int64_t n, int mode);
/* hnsw_insert that also reports which edges it added / removed, for a host that persists the graph edge by edge (the "{t}_edges"
 * shadow table: src/hnsw_vtab.c:755-776 rewrites every edge of the new node AND of each of its neighbours per insert; with the
 * log only the changed rows are touched).  *n_log = entries written, or -1 when the log cannot describe this insert: more
 * than cap changes, a list wider than 64 links, or a touched node whose persisted copy is not known to match the index —
 * its lists were edited by mn_hnsw_delete (which the reference never persists, src/hnsw_vtab.c:702-706) or the caller said
 * so with mn_hnsw_log_invalidate.  mn_hnsw_take_dirty is then still complete (and taking a node through it makes the node
 * known again); with a valid log the persist set is emptied.  Same graph and return convention as mn_hnsw_insert. */
typedef struct {
    int op;          /* 1: edge src -> dst added at `level` with `distance`;  2: edge src -> dst removed */
    int level;
    int64_t src, dst;
    float distance;  /* dist_func(src vector, dst vector), as persist_node stores it (src/hnsw_vtab.c:262-280) */
} mn_edge_change;
int mn_hnsw_insert_logged(mn_index *idx, int64_t id, const float *vector, mn_edge_change *log, int cap, int *n_log);
/* the caller's persisted copy of these n nodes — or, ids == NULL, of every node present now (a ROLLBACK took rows away that the
 * index keeps) — is not what the index holds: unknown until rewritten whole.  Ids that are not in the index are ignored. */
int mn_hnsw_log_invalidate(mn_index *idx, const int64_t *ids, int64_t n);
/* Bulk build helper: splits [n] into batches growing with the index (batch ≤ max(1, count/grow_div),
 * capped at max_batch) and calls the batched schedule on each.  grow_div ≤ 0 → 16, max_batch ≤ 0 → 8192. */
int mn_hnsw_build(mn_index *idx, const int64_t *ids, const float *vectors, int64_t n, int grow_div, int max_batch);
/* the same from rows already in HBM on the index's device ([n][dim] f32): same batches, same graph; nothing visits the host */
int mn_hnsw_build_dev(mn_index *idx, const int64_t *ids, const float *d_vectors, int64_t n, int grow_div, int max_batch);
int mn_hnsw_device(mn_index *idx); /* HIP ordinal the index lives on */
/* One MN_BUILD_BATCHED batch in three steps, so that several GPUs that each hold a replica of the index can share
 * its search half (the dominant cost) and still all end up with the graph a single GPU builds:
 *   stage  — add the batch's nodes (ids, levels from the index's own level stream, vectors) on this replica;
 *            returns m = nodes waiting to be searched and linked (n, or n-1 when the first node of an empty index
 *            just became the entry point), -1 on error;
 *   search — search staged nodes [lo, hi) against the graph as it stands and write their selected-neighbour lists
 *            into rows lo..hi of the caller's DEVICE arrays d_sel [m][nlev][row_width] / d_nsel [m][nlev]
 *            (mn_hnsw_batch_dims gives nlev = top layer + 1 and row_width = 2M);
 *   (the caller exchanges the rows between replicas — an all-gather — so that every replica holds all m)
 *   link   — apply the whole batch's links from the full arrays and update entry point / top layer.
 * Every replica must stage the same batches in the same order (the level stream is part of the index state). */
int mn_hnsw_batch_stage(mn_index *idx, const int64_t *ids, const float *vectors, int64_t n);
int mn_hnsw_batch_dims(mn_index *idx, int *nlev, int *row_width);
int mn_hnsw_batch_search(mn_index *idx, int lo, int hi, int *d_sel, int *d_nsel);
int mn_hnsw_batch_link(mn_index *idx, const int *d_sel, const int *d_nsel);

/* hnsw_search (:670-704): ef = max(ef, k); results ascending by distance; returns count ≤ k */
int mn_hnsw_search(mn_index *idx, const float *query, int k, int ef_search, mn_search_result *results);
/* nq independent hnsw_search calls in one launch.  queries host [nq][dim]; out_ids/out_dists host
 * [nq][k] (unused tail: id -1, distance 0); out_counts host [nq].  Returns 0 / -1. */
int mn_hnsw_search_batch(mn_index *idx, const float *queries, int64_t nq, int k, int ef_search, int64_t *out_ids,
                         float *out_dists, int *out_counts);
/* Same with every buffer already resident in device memory (HBM) on idx's device; asynchronous on the
 * index's stream — call mn_hnsw_sync before reading.  */
int mn_hnsw_search_batch_dev(mn_index *idx, const float *d_queries, int64_t nq, int k, int ef_search,
                             int64_t *d_out_ids, float *d_out_dists, int *d_out_counts);
int mn_hnsw_sync(mn_index *idx);

/* hnsw_delete (:717-805): soft delete + neighbour reconnection.  0 / -1 (absent or already deleted).  Reconnection may
 * leave a list longer than M_max — the reference's node_add_neighbor grows lists without bound (:142-163) — which the
 * device rows follow (the table is re-strided when a list outgrows the row; the next insert that touches the list prunes it
 * back to M_max exactly as the reference does, :601-646).  Only the few rows of the deleted node's neighbours move
 * between host and device. */
int mn_hnsw_delete(mn_index *idx, int64_t id);

/* hnsw_get_vector / hnsw_get_node (:226-236): copies dim floats out (index lives in HBM).
 * Returns 0, or -1 if absent or deleted. */
int mn_hnsw_get_vector(mn_index *idx, int64_t id, float *out);

/* ---- state the vtab persists / reloads (replaces direct HnswIndex/HnswNode field access in
 *      src/hnsw_vtab.c:237-341,405-462 and ht_find/node_create/node_add_neighbor, src/hnsw_algo.h:98-104) ---- */
int mn_hnsw_node_count(mn_index *idx);     /* live nodes (HnswIndex.node_count) */
int64_t mn_hnsw_entry_point(mn_index *idx); /* -1 if empty */
int mn_hnsw_max_level(mn_index *idx);
int mn_hnsw_node_level(mn_index *idx, int64_t id);   /* -1 if absent */
int mn_hnsw_node_deleted(mn_index *idx, int64_t id); /* -1 if absent */
/* neighbour ids of `id` at `level` in list order; returns the count (may exceed cap), -1 if absent */
int mn_hnsw_neighbors(mn_index *idx, int64_t id, int level, int64_t *out, int cap);
/* load path of load_index_from_shadow (src/hnsw_vtab.c:286-341): nodes first, then edges, then entry.
 * Host-side staging only: vectors and rows reach the device in bulk at the next compute call.
 * mn_hnsw_load_node: 0 = added; 1 = not added because the node table was full or the id is already present — the
 * reference's loop ignores that failure and carries on without the node (:316), and so should the caller; -1 = error.
 * mn_hnsw_load_neighbors: lists may be longer than M_max (a database whose deletes grew them); -1 only on a real error. */
int mn_hnsw_load_node(mn_index *idx, int64_t id, const float *vector, int level, int deleted);
int mn_hnsw_load_neighbors(mn_index *idx, int64_t id, int level, const int64_t *nbrs, int n);
int mn_hnsw_set_entry(mn_index *idx, int64_t entry_point, int max_level);

/* Bulk export (what a bulk persist_node replacement reads, src/hnsw_vtab.c:237-283; also lets a
 * checker mirror the device graph).  Slots are insertion order; deleted nodes keep their slot. */
int mn_hnsw_slot_count(mn_index *idx);
int mn_hnsw_export_nodes(mn_index *idx, int64_t *ids, int *levels, int *deleted); /* each [slot_count] */
int mn_hnsw_export_vectors(mn_index *idx, float *out);                             /* [slot_count][dim] */
/* neighbour rows of every slot at `level` as slot indices, -1 padded: out is [slot_count][width];
 * rows of nodes whose level < `level` are all -1.  *width = mn_hnsw_row_width(idx, level): 2M at level 0, M above,
 * more once a delete or a loaded database has grown a list past that. */
int mn_hnsw_row_width(mn_index *idx, int level);
int mn_hnsw_export_links(mn_index *idx, int level, int *out, int *width);

/* The set of nodes the reference's xUpdate would have re-persisted (src/hnsw_vtab.c:755-768: the new
 * node and every neighbour it linked to), accumulated on the device over all inserts since the last
 * call, so the SQL layer can write shadow tables once per transaction instead of once per row.
 * Fills ids[0..count) in slot order and clears the set; if count > cap nothing is cleared and the
 * count is returned so the caller can retry with room.  -1 on error. */
int64_t mn_hnsw_take_dirty(mn_index *idx, int64_t *ids, int64_t cap);

/* All edges of the given nodes with the distance persist_node stores next to each one
 * (src/hnsw_vtab.c:268-279: dist_func(node, neighbour), 0.0 when the neighbour is soft-deleted).
 * Fills parallel arrays (source id, target id, level, distance) up to `cap`; returns the number of
 * edges (may exceed cap → call again with more room), -1 on error / unknown id. */
int64_t mn_hnsw_edges_of(mn_index *idx, const int64_t *ids, int n, int64_t *out_src, int64_t *out_dst, int *out_level,
                         float *out_dist, int64_t cap);

/* ---- measurement hooks (bench.py) ---- */
typedef struct {
    double last_kernel_ms;   /* HIP-event time of the last dominant kernel launch on the index's stream (0 after a search of a few
                              * host-resident queries, mn_hnsw_search: that path records no events) */
    int64_t last_n_dist;     /* distance evaluations performed by that launch (device counter) */
    int64_t last_n_expanded; /* neighbour rows read by that launch */
    int64_t last_n_overflow; /* queries whose heaps exceeded workspace (must be 0) */
} mn_launch_stats;
int mn_hnsw_last_launch(mn_index *idx, mn_launch_stats *out);
/* Totals over the batch-synchronous inserts (MN_BUILD_BATCHED / mn_hnsw_build) since the last reset: HIP-event time of
 * the search half (k_beam<BUILD>, the dominant kernel of a build) and of the link half, with the search half's device
 * counters — what the build-side roofline of bench.py is computed from. */
typedef struct {
    double search_ms, link_ms;
    int64_t n_dist, n_expanded; /* distance evaluations / neighbour rows read by the searches */
    int64_t batches, nodes;
} mn_build_stats;
int mn_hnsw_build_stats(mn_index *idx, mn_build_stats *out, int reset);
/* device malloc/free/copies so a non-HIP host (ctypes) can stage HBM-resident inputs */
void *mn_dev_malloc(mn_index *idx, size_t bytes);
void mn_dev_free(mn_index *idx, void *p);
int mn_dev_upload(mn_index *idx, void *dst_dev, const void *src_host, size_t bytes);
int mn_dev_download(mn_index *idx, void *dst_host, const void *src_dev, size_t bytes);
/* exact brute-force top-k on device (ground truth for recall@k; not on the parity path).
 * d_queries device [nq][dim]; out host [nq][k] ids. */
int mn_hnsw_bruteforce_topk(mn_index *idx, const float *d_queries, int64_t nq, int k, int64_t *out_ids);

/* ---- graph_csr.h / graph_community.c replacements (a18-a22) ---- */
typedef struct mn_graph mn_graph; /* device-resident adjacency: GraphData.out / .in (src/graph_load.h:27-37) as two
                                     CsrArray (src/graph_csr.h:27-34: int32 offsets[V+1], int32 targets[E], f64 weights[E]|NULL) */
/* Per-node edge order must be the adjacency-list order (edge-table row order).  weights NULL = 1.0. */
mn_graph *mn_graph_create(int n_nodes, const int *off_out, const int *tgt_out, const double *w_out, const int *off_in,
                          const int *tgt_in, const double *w_in, int device);
/* The same graph from the reference's stored form: the rows of graph_adjacency's shadow tables
 * "{t}_csr_fwd" / "{t}_csr_rev" (src/graph_adjacency.c:182-197), one mn_csr_block per row in block_id order.  A block
 * holds offsets int32[nodes_in_block + 1] rebased to 0, targets int32[edges] as GLOBAL node indices, weights f64[edges]
 * or nothing (src/graph_csr.c:335-400).  Blocks are uploaded to their place in the device arrays directly — the
 * reference's deserialize → merge → GraphData adjacency-list round trip (src/graph_adjacency.c:1458-1530) has no
 * counterpart.  A single monolithic row (block_size 0) is one block. */
typedef struct {
    const void *offsets;
    int offsets_bytes;
    const void *targets;
    int targets_bytes;
    const void *weights; /* NULL / 0 bytes when unweighted */
    int weights_bytes;
} mn_csr_block;
mn_graph *mn_graph_create_blocked(int n_nodes, const mn_csr_block *fwd, int n_fwd, const mn_csr_block *rev, int n_rev, int device);
void mn_graph_destroy(mn_graph *g);
const char *mn_graph_last_error(void);

typedef enum {
    MN_LEIDEN_SEQUENTIAL = 0, /* the reference's in-order sweep with immediate moves: community[] and Q bit-identical */
    MN_LEIDEN_BATCHED = 1     /* parallel schedules: whole-graph synchronous sweeps, or rounds of `batch` nodes */
} mn_leiden_mode;
typedef struct {
    int64_t iterations, moves, move_sweeps, refine_sweeps;
    int n_communities;
    double device_ms;
} mn_leiden_stats;
/* run_leiden (src/graph_community.c:336-429).  use_both = (direction == "both").  community_out[n_nodes] is
 * renumbered 0..K-1 in first-seen order; *modularity_out = Q.  batch (BATCHED only): 0 or 1 → the default schedule, whole-graph
 * synchronous sweeps in which every positive-gain mover applies and every 3rd sweep is "pick-less" (moves to a smaller
 * community id only), finished by the round schedule if 48 sweeps do not settle; < 0 → the same with a pick-less sweep every
 * -batch sweeps; > 1 → rounds of `batch` nodes with the safe-winner commit rule.  Returns 0 / -1. */
int mn_graph_leiden(mn_graph *g, double resolution, int use_both, int mode, int batch, int *community_out,
                    double *modularity_out);
int mn_graph_leiden_stats(mn_graph *g, mn_leiden_stats *out);

/* brandes_compute (src/graph_centrality.c:393-505): node betweenness cb_out[n] and, when eb_out != NULL, the reference's dense
 * edge matrix eb_out[n*n] (EB[v*n + w] = betweenness of v -> w; the reference allocates the same n x n doubles).  BFS for
 * unweighted graphs, Dijkstra for weighted ones; direction 0 = "both" (halved), 1 = "forward", 2 = "reverse";
 * auto_approx > 0 and n > auto_approx → ceil(sqrt(n)) evenly spaced sources, scaled (:420-429); normalized → / ((n-1)(n-2)[/2]).
 * The graph must have been created with the lists the direction traverses (out for 1, in for 2, both for 0), as
 * graph_data_load fills them (src/graph_load.c:144-250).  Bit-identical to the reference.  0 / -1. */
int mn_graph_betweenness(mn_graph *g, int direction, int auto_approx, int normalized, double *cb_out, double *eb_out);
double mn_graph_last_ms(mn_graph *g); /* device time of the last betweenness call */
/* GraphData.out as CSR on the host (off[n+1], tgt[mn_graph_out_edge_count]) — the order graph_edge_betweenness emits rows in */
long long mn_graph_out_edge_count(mn_graph *g);
int mn_graph_out_lists(mn_graph *g, int *off, int *tgt);

/* ---- node2vec.c replacements (a14-a17) ---- */
typedef struct {
    int dim;            /* 1..1024 (src/node2vec.c:447) */
    double p, q;        /* return / in-out parameters */
    int num_walks, walk_length, window, neg_samples;
    double learning_rate;
    int epochs;
    int batch_walks; /* MN_N2V_BATCHED only: walks per batch; <= 0 → clamp(n/64, 1, 16384) */
} mn_n2v_params;
typedef enum {
    MN_N2V_SEQUENTIAL = 0, /* the reference's single serial SGD stream: output bytes identical to the reference's */
    MN_N2V_BATCHED = 1     /* batch-synchronous mini-batch schedule: one wavefront per walk,
                              per-walk RNG streams, deterministic per-row accumulation; bit-identical to the CPU
                              restatement of the same schedule, statistically equivalent to the serial stream */
} mn_n2v_mode;
typedef struct {
    int64_t pairs;    /* (center, context) pairs trained */
    double device_ms;
} mn_n2v_stats;
/* The compute of node2vec_train (src/node2vec.c:486-551): sgns_create (rng 42), walks + SGNS, L2 normalisation.
 * Graph = node2vec.c's own adjacency (first-seen node order, undirected, de-duplicated, :72-138) as CSR:
 * neighbours of node i are adj[off[i] .. off[i+1]) in list order.  out is host [n][dim] f32 — the vectors the
 * reference INSERTs with rowid = i + 1 (:575).  Returns n, or -1. */
int mn_node2vec_train(int n_nodes, const int *off, const int *adj, const mn_n2v_params *prm, int mode, int device, float *out,
                      mn_n2v_stats *stats);
const char *mn_node2vec_last_error(void);
/* node2vec_train's compute AND its output step (src/node2vec.c:540-583: every embedding INSERTed into the output hnsw_index,
 * rowid = first-seen index + 1) without the embeddings leaving HBM: trained (MN_N2V_BATCHED) and normalised on idx's device,
 * then mn_hnsw_build_dev with rowids first_rowid + i.  host_out (or NULL): the embeddings as well, one bulk copy, for a host
 * that persists them.  *build_seconds (or NULL): wall time of the index build.  Same embedding bytes as mn_node2vec_train, same
 * graph as mn_hnsw_build on them.  Returns n, or -1. */
int mn_node2vec_train_into(int n_nodes, const int *off, const int *adj, const mn_n2v_params *prm, int mode, mn_index *idx,
                           int64_t first_rowid, float *host_out, mn_n2v_stats *stats, double *build_seconds);

/* The batched schedule as a session, so that several GPUs can share one training run: every rank produces the
 * samples of its slice of a batch's walks, the (centre, target, err) triples are exchanged (RCCL all-gather, rank
 * order = walk order) and every replica applies the whole batch — the N-GPU embeddings are bit-identical to the
 * 1-GPU ones.  Buffers passed to samples/apply are DEVICE pointers on the session's device; both calls are queued on the
 * session's own stream (in call order) and return at once — mn_n2v_sync before another stream or the host touches the
 * buffers. */
typedef struct mn_n2v_session mn_n2v_session;
mn_n2v_session *mn_n2v_begin(int n_nodes, const int *off, const int *adj, const mn_n2v_params *prm, int device);
int mn_n2v_batch_walks(mn_n2v_session *s);  /* resolved walks per batch */
int mn_n2v_sample_slots(mn_n2v_session *s);   /* sample slots per walk = walk_length * 2*window * (1+neg) */
int mn_n2v_position_slots(mn_n2v_session *s); /* position slots per walk = walk_length */
/* walks of start nodes [lo, hi) (hi - lo <= batch_walks) of pass (epoch, w).  Per walk: `sample_slots` (centre,
 * target, err) triples — the target-side updates, unused slots carry -1 — and `position_slots` (centre, neu1e[dim])
 * entries — the centre-side update of each walk position (src/node2vec.c:347,:383-391), -1 past the walk's end. */
int mn_n2v_samples(mn_n2v_session *s, int epoch, int w, int lo, int hi, int *d_center, int *d_target, float *d_err,
                   int *d_pos_center, float *d_pos_neu);
/* applies ns sample slots and np position slots in the given order (any concatenation of mn_n2v_samples outputs) */
int mn_n2v_apply(mn_n2v_session *s, const int *d_center, const int *d_target, const float *d_err, int64_t ns,
                 const int *d_pos_center, const float *d_pos_neu, int64_t np);
int mn_n2v_sync(mn_n2v_session *s);
int mn_n2v_finish(mn_n2v_session *s, float *out, mn_n2v_stats *stats); /* L2 normalise, download [n][dim] */
/* L2 normalise and leave the embeddings in HBM: *d_out = [n][dim] f32 on the session's device, valid until mn_n2v_end */
int mn_n2v_finish_dev(mn_n2v_session *s, const float **d_out, mn_n2v_stats *stats);
void mn_n2v_end(mn_n2v_session *s);

/* ---- graph_tvf.c's remaining edge-list algorithms ----
 * Nodes are first-seen indices (src of row 0, dst of row 0, src of row 1, ... as pr_adj_find_or_add / uf_find_or_add
 * number them, src/graph_tvf.c:1591-1613,1231-1247); src[e] -> dst[e] are the edge table's rows in order, duplicates and
 * self loops kept as the reference keeps them.  Host arrays. */
typedef struct {
    double device_ms;
    int iterations; /* PageRank: iterations run; components: hook rounds */
    int64_t aux;    /* PageRank: dangling nodes */
} mn_graph_algo_stats;
/* run_pagerank (src/graph_tvf.c:1631-1797): rank_out[n] f64, bit-identical to the reference's sequential push loop
 * (the same additions in the same order, pulled per target).  0 / -1. */
int mn_graph_pagerank(int n_nodes, int64_t n_edges, const int *src, const int *dst, double damping, int iterations, int device,
                      double *rank_out, mn_graph_algo_stats *stats);
typedef enum {
    MN_COMPONENTS_EXACT = 0, /* the reference's union sequence replayed: component_id = its union-find root */
    MN_COMPONENTS_FAST = 1   /* parallel hooking: same partition and sizes, component_id = smallest node index */
} mn_components_mode;
/* run_components (src/graph_tvf.c:1314-1366): component_id[n], component_size[n].  0 / -1. */
int mn_graph_components(int n_nodes, int64_t n_edges, const int *src, const int *dst, int mode, int device, int *component_id,
                        int *component_size, mn_graph_algo_stats *stats);
const char *mn_graph_algo_last_error(void);

/* csr_apply_delta (src/graph_csr.c:175-325), the merge step of graph_adjacency's incremental rebuild
 * (src/graph_adjacency.c:721-1005): old CSR + delta log -> new CSR; per node the log is replayed in order (INSERT appends,
 * DELETE removes the first occurrence by moving the last element into its place), so lists come out in the reference's
 * order.  Host arrays in; new_offsets[max(new_node_count, old_node_count) + 1] is the caller's, *new_targets /
 * *new_weights are malloc'ed here (NULL when the result has no edges; release with mn_host_free).  mn_csr_delta has the
 * layout of CsrDelta (src/graph_csr.h:37-42).  0 / -1. */
typedef struct {
    int32_t src_idx, dst_idx;
    double weight;
    int op; /* 1 = INSERT, 2 = DELETE */
} mn_csr_delta;
int mn_csr_apply_delta(int old_node_count, const int *old_offsets, const int *old_targets, const double *old_weights,
                       int has_weights, const mn_csr_delta *deltas, int delta_count, int new_node_count, int device,
                       int *new_offsets, int **new_targets, double **new_weights, int *new_edge_count);
void mn_host_free(void *p);

/* ---- multi-GPU: one rank per GPU of a node, processes or threads; RCCL over xGMI ----
 * The reference is single-device; these entry points are what its host (hnsw_vtab.c / node2vec.c) would call to use the
 * node's other GPUs.  The only collective is an all-gather of equal-sized device buffers:
 *   RCCL transport: rank 0 calls mn_comm_unique_id and hands the 128 bytes to the other ranks by whatever means the
 *     host has (a file, a socket, MPI, torch.distributed's store); every rank then calls mn_comm_init_rccl.  librccl is
 *     loaded on first use, so single-GPU users never touch it.
 *   host transport: the caller provides the all-gather over HOST buffers (recv = world x bytes_per_rank, rank order) —
 *     rehearsals where several ranks share one GPU (RCCL refuses that), or a host with its own fabric. */
typedef struct mn_comm mn_comm;
#define MN_COMM_ID_BYTES 128
typedef int (*mn_host_allgather_fn)(void *user, const void *send, void *recv, size_t bytes_per_rank);
int mn_comm_unique_id(void *id128);
mn_comm *mn_comm_init_rccl(int world, int rank, const void *id128, int device);
mn_comm *mn_comm_init_host(int world, int rank, mn_host_allgather_fn fn, void *user, int device);
int mn_comm_world(mn_comm *c);
int mn_comm_rank(mn_comm *c);
void mn_comm_destroy(mn_comm *c);
const char *mn_comm_last_error(void);

/* mn_hnsw_build on `world` GPUs that each hold a replica of ONE index: the batches are those of mn_hnsw_build; inside a
 * batch rank r searches a contiguous slice of the batch's nodes against its replica (k_beam<BUILD>, the dominant cost),
 * the selected-neighbour lists are all-gathered and every replica links the whole batch.  Every replica ends with the
 * graph a single GPU builds, bit for bit.  ids / vectors: the SAME host arrays on every rank.  Batches below min_split
 * nodes (<= 0 -> 256) are searched whole by every rank (no exchange).  0 / -1. */
int mn_hnsw_build_shared(mn_index *idx, mn_comm *c, const int64_t *ids, const float *vectors, int64_t n, int grow_div,
                         int max_batch, int min_split);
/* BASELINE config 3: the index is sharded (rowid mod world -> one HNSW graph per GPU, built independently); every rank
 * searches the SAME queries on its shard, the per-shard top-k — k x (int64 id, f32 distance) per query — are all-gathered
 * and merged on the device in the total order (distance, shard rank, position).  Every rank receives the merged result.
 * Device buffers on idx's device; asynchronous on the index's stream like mn_hnsw_search_batch_dev. */
int mn_hnsw_search_sharded_dev(mn_index *idx, mn_comm *c, const float *d_queries, int64_t nq, int k, int ef_search,
                               int64_t *d_out_ids, float *d_out_dists, int *d_out_counts);
/* the same with host buffers */
int mn_hnsw_search_sharded(mn_index *idx, mn_comm *c, const float *queries, int64_t nq, int k, int ef_search, int64_t *out_ids,
                           float *out_dists, int *out_counts);
/* MN_N2V_BATCHED data-parallel over the ranks (BASELINE config 4): every rank holds a replica of both matrices and computes
 * the samples of its contiguous slice of each batch's walks.  A sample is then sent to the ONE rank that owns its target row
 * (a position's neu1e to the owner of its centre row): buckets by destination shard, stable, exchanged all-to-all (ncclSend /
 * ncclRecv over xGMI: 1 / world of the bytes an all-gather moves); every rank sorts and applies what it received to its rows
 * — received buckets stand in rank order = walk order, so a row gets the additions, in the order, one GPU gives it — and the
 * updated row shards of both matrices are all-gathered.  The embeddings are bit-identical to mn_node2vec_train(..,
 * MN_N2V_BATCHED) on one GPU.  stats->pairs is this rank's share.  Returns n / -1. */
int mn_node2vec_train_shared(mn_comm *c, int n_nodes, const int *off, const int *adj, const mn_n2v_params *prm, int device,
                             float *out, mn_n2v_stats *stats);
/* run_leiden (src/graph_community.c:336-429) on the ranks' GPUs — north_star: the per-node local-move sweep "partitioned
 * across the GPUs ... modularity partials".  Every rank holds the whole graph (g on its own device) and the whole state; the
 * evaluation of every synchronous sweep is divided by node range, the decisions (N int32 per sweep) are all-gathered and every
 * replica applies all of them; compute_modularity's per-node terms (:131) are divided and all-gathered the same way.  Every
 * rank returns what mn_graph_leiden(g, .., MN_LEIDEN_BATCHED, batch, ..) returns on one GPU, bit for bit.  0 / -1. */
int mn_graph_leiden_shared(mn_graph *g, mn_comm *c, double resolution, int use_both, int batch, int *community_out,
                           double *modularity_out);

/* ---- the same sharded index (BASELINE config 3) inside ONE process: a C host that owns several GPUs ----
 * rowid mod n -> shard; every shard is an ordinary mn_index on its own GPU (mn_shards_index gives access for persistence:
 * mn_hnsw_take_dirty etc.).  A search uploads the queries to every shard's GPU, searches all shards at once (one stream
 * per GPU, queued by the calling thread), copies the per-shard top-k to the first shard's GPU (hipMemcpyPeerAsync) and
 * merges there with the kernel mn_hnsw_search_sharded uses — the same result, bit for bit, as one rank per GPU over RCCL.
 * devices may repeat an ordinal (several shards on one GPU).  What a loadable extension would bind for
 * MUNINN_DEVICE=0,1,...: hnsw_vtab.c's hnsw_insert / hnsw_search / hnsw_delete call sites (src/hnsw_vtab.c:748, :604,
 * :698) take these instead of the single-index calls.  Errors: mn_shards_last_error(). */
typedef struct mn_shards mn_shards;
mn_shards *mn_shards_create(int dim, int metric, int M, int ef_construction, const int *devices, int n);
void mn_shards_destroy(mn_shards *s);
int mn_shards_count(const mn_shards *s);
mn_index *mn_shards_index(mn_shards *s, int i);
int mn_shards_of(const mn_shards *s, int64_t id); /* ((id mod n) + n) mod n */
int mn_shards_set_order(mn_shards *s, int order);
int mn_shards_insert(mn_shards *s, int64_t id, const float *vector);   /* hnsw_insert on the id's shard */
int mn_shards_delete(mn_shards *s, int64_t id);                         /* hnsw_delete on the id's shard */
/* mn_hnsw_build per shard (input order kept inside every shard), the shards built side by side: one host thread per GPU */
int mn_shards_build(mn_shards *s, const int64_t *ids, const float *vectors, int64_t n, int grow_div, int max_batch);
int mn_shards_search(mn_shards *s, const float *query, int k, int ef_search, mn_search_result *results);
int mn_shards_search_batch(mn_shards *s, const float *queries, int64_t nq, int k, int ef_search, int64_t *out_ids,
                           float *out_dists, int *out_counts);
const char *mn_shards_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MUNINN_HIP_H */
