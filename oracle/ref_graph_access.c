/*
 * ref_graph_access.c — compiled INTO oracle/_ref/muninn.so next to the reference's own sources
 * (oracle/Makefile).  TEST INFRASTRUCTURE ONLY.  Builds the reference's GraphData through the
 * reference's own functions (graph_data_init / graph_data_find_or_add / adj_add,
 * src/graph_load.c:26-139) from an integer edge list and calls the reference's run_leiden
 * (src/graph_community.c:336) — no SQL involved, nothing restated.
 */
#include <stdio.h>
#include "graph_load.h"
#include "graph_community.h"

double run_leiden(const GraphData *g, int *community, double resolution, const char *direction);

/* edges in table-row order; node index = first appearance (src before dst), as graph_data_load does
 * (src/graph_load.c:236-243).  direction: 0 "both", 1 "forward", 2 "reverse".
 * out_index[i] receives the reference's node index of integer node id i (or -1). */
double ref_leiden_edges(int n_ids, int n_edges, const int *src, const int *dst, const double *w, int direction,
                        double resolution, int *out_index, int *community, int *out_n) {
    GraphData g;
    graph_data_init(&g);
    int add_forward = direction != 2, add_reverse = direction != 1;
    char a[32], b[32];
    for (int e = 0; e < n_edges; e++) {
        snprintf(a, sizeof a, "%d", src[e]);
        snprintf(b, sizeof b, "%d", dst[e]);
        int si = graph_data_find_or_add(&g, a);
        int di = graph_data_find_or_add(&g, b);
        double wt = w ? w[e] : 1.0;
        if (add_forward)
            adj_add(&g.out[si], di, wt);
        if (add_reverse)
            adj_add(&g.in[di], si, wt);
        g.edge_count++;
    }
    for (int i = 0; i < n_ids; i++) {
        snprintf(a, sizeof a, "%d", i);
        out_index[i] = graph_data_find(&g, a);
    }
    *out_n = g.node_count;
    double Q = run_leiden(&g, community, resolution, direction == 0 ? "both" : (direction == 1 ? "forward" : "reverse"));
    graph_data_destroy(&g);
    return Q;
}


/* the reference's own csr_apply_delta (src/graph_csr.c:175) on caller-supplied arrays; returns the new edge count */
#include <stdlib.h>
#include <string.h>
#include "graph_csr.h"
int ref_csr_apply_delta(int old_n, const int *off, const int *tgt, const double *w, int has_weights, int nd, const int *dsrc,
                        const int *ddst, const double *dw, const int *dop, int new_n, int *new_off, int *new_tgt, double *new_w) {
    CsrArray old, nw;
    memset(&old, 0, sizeof(old));
    old.node_count = old_n;
    old.edge_count = old_n ? off[old_n] : 0;
    old.offsets = (int32_t *)off;
    old.targets = (int32_t *)tgt;
    old.weights = (double *)w;
    old.has_weights = has_weights;
    CsrDelta *dl = (CsrDelta *)calloc((size_t)(nd ? nd : 1), sizeof(CsrDelta));
    for (int d = 0; d < nd; d++) {
        dl[d].src_idx = dsrc[d];
        dl[d].dst_idx = ddst[d];
        dl[d].weight = dw ? dw[d] : 0.0;
        dl[d].op = dop[d];
    }
    if (csr_apply_delta(&old, dl, nd, new_n, &nw) != 0) {
        free(dl);
        return -1;
    }
    free(dl);
    memcpy(new_off, nw.offsets, ((size_t)nw.node_count + 1) * sizeof(int));
    if (nw.edge_count) {
        memcpy(new_tgt, nw.targets, (size_t)nw.edge_count * sizeof(int));
        if (has_weights && nw.weights && new_w)
            memcpy(new_w, nw.weights, (size_t)nw.edge_count * sizeof(double));
    }
    int e = nw.edge_count;
    csr_destroy(&nw);
    return e;
}
