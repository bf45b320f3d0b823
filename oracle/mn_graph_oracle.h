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

/* run_leiden (src/graph_community.c:336-429).  batch <= 1: the reference's sequential sweeps;
 * batch > 1: the batch-synchronous schedule of the HIP fast mode.  Returns modularity Q. */
double orc_leiden(const orc_graph *g, int *community, double resolution, int use_both, int batch, orc_leiden_stats *st);
/* compute_modularity (src/graph_community.c:109-142) */
double orc_modularity(const orc_graph *g, const int *community, double resolution, double m, int use_both);

#ifdef __cplusplus
}
#endif
#endif
