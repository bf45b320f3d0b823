/*
 * mn_graph_oracle.h — CPU oracle, graph half (Leiden; Node2Vec walk + SGNS).  TEST INFRASTRUCTURE ONLY:
 * see mn_oracle.h for the rules.  Citations are file:line in the reference repository.
 */
#ifndef MN_GRAPH_ORACLE_H
#define MN_GRAPH_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* GraphData.out / GraphData.in (src/graph_load.h:27-37) as CSR (src/graph_csr.h:27-34): per-node edge
 * order is the adjacency-list order; weights NULL = 1.0 */
typedef struct {
    int n;
    const int *off_out, *tgt_out;
    const double *w_out;
    const int *off_in, *tgt_in;
    const double *w_in;
} orc_graph;

typedef struct {
    int64_t iterations, moves, move_sweeps, refine_sweeps;
} orc_leiden_stats;

/* run_leiden (src/graph_community.c:336-429).  batch 0 or 1: the reference's sequential sweeps; batch > 1: the round
 * schedule of the HIP fast mode (rounds of `batch` nodes, safe winners); batch < 0: its whole-graph synchronous sweeps with a
 * pick-less sweep every -batch sweeps (the device's default is -3).  Returns modularity Q. */
double orc_leiden(const orc_graph *g, int *community, double resolution, int use_both, int batch, orc_leiden_stats *st);
/* compute_modularity (src/graph_community.c:109-142) */
double orc_modularity(const orc_graph *g, const int *community, double resolution, double m, int use_both);

/* ---- a14-a17: Node2Vec (src/node2vec.c) ----
 * Undirected, de-duplicated adjacency in first-seen node order (graph_node_index / graph_add_edge,
 * src/node2vec.c:72-109) given as CSR: neighbours of node i are adj[off[i] .. off[i+1]) in list order. */
typedef struct {
    int n;
    const int *off, *adj;
} orc_n2v_graph;

typedef struct {
    int dim;
    double p, q;
    int num_walks, walk_length, window, neg_samples;
    double lr;
    int epochs;
} orc_n2v_params;

/* node2vec_train's compute (src/node2vec.c:486-551): sgns_create (rng 42), the serial walk + SGNS stream,
 * then L2 normalisation.  out is [n][dim] f32 — the bytes the reference INSERTs into the output table. */
int orc_node2vec_train(const orc_n2v_graph *g, const orc_n2v_params *p, float *out, int64_t *n_pairs);
/* the batch-synchronous schedule of the HIP MN_N2V_BATCHED mode (B walks per batch) */
int orc_node2vec_train_batched(const orc_n2v_graph *g, const orc_n2v_params *p, int B, float *out, int64_t *n_pairs);
/* biased_walk (src/node2vec.c:168-226) from an explicit rng state; returns the walk length */
int orc_biased_walk(const orc_n2v_graph *g, int start, double p, double q, int walk_length, int *walk, unsigned *rng);
/* Builds the reference's Graph from an edge list (first-seen indices, undirected, de-duplicated);
 * returns n; off must hold n_max+1 ints, adj 2*n_edges ints, index_of_id n_ids ints (or NULL). */
int orc_n2v_build_graph(int n_edges, const int *src, const int *dst, int n_ids, int *off, int *adj, int *index_of_id);

/* brandes_compute (src/graph_centrality.c:393-505) over orc_graph (out / in lists in adjacency order, weights or NULL):
 * direction 0 "both", 1 "forward", 2 "reverse".  cb[n]; eb = NULL or n*n doubles (zeroed by the caller). */
int orc_betweenness(const orc_graph *g, int direction, int auto_approx, int normalized, double *cb, double *eb);
/* csr_apply_delta (src/graph_csr.c:175-325): per-node replay of the delta log.  delta arrays are parallel
 * (src, dst, weight, op: 1 INSERT, 2 DELETE).  new_off must hold max(new_n, old_n) + 1 ints, new_tgt / new_w room for
 * old edges + delta count.  Returns the new edge count, -1 on error. */
int orc_csr_apply_delta(int old_n, const int *off, const int *tgt, const double *w, int has_weights, int nd, const int *dsrc,
                        const int *ddst, const double *dw, const int *dop, int new_n, int *new_off, int *new_tgt, double *new_w);
/* run_pagerank's power iteration (src/graph_tvf.c:1676-1716) over first-seen node indices; edges src[e] -> dst[e] in
 * edge-table row order (duplicates and self loops kept).  rank_out[n]. */
int orc_pagerank(int n, int n_edges, const int *src, const int *dst, double damping, int iterations, double *rank_out);
/* run_components (src/graph_tvf.c:1314-1366) with the union-find of :1231-1273: component_id = root, component_size */
int orc_components(int n, int n_edges, const int *src, const int *dst, int *component_id, int *component_size);

#ifdef __cplusplus
}
#endif
#endif
