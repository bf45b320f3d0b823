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
