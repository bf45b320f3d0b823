/* mn_csr_adapter.c — csr_apply_delta with the reference's own signature (src/graph_csr.c:175-325, declared in
 * src/graph_csr.h:76-83) over the device implementation mn_csr_apply_delta (libmuninn_hip.so).
 *
 * graph_adjacency's incremental rebuild calls this once per touched block and direction (src/graph_adjacency.c:864,910):
 * old CSR block + that block's slice of the delta log -> new CSR block.  Linking this definition instead of the reference's
 * (oracle/Makefile `integrated` does, with the reference's own graph_adjacency.c unchanged around it) puts that merge on the
 * GPU.  The two structs are restated from src/graph_csr.h:27-42 — same field order and types — not included: this file
 * builds without the reference tree.  Ownership as the reference's: new_csr's arrays are malloc'ed and released by the
 * caller's csr_destroy (free). */
#include "../../include/muninn_hip.h"
#include "mn_nodemap.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { /* src/graph_csr.h:27-34 */
    int32_t node_count;
    int32_t edge_count;
    int32_t *offsets;
    int32_t *targets;
    double *weights;
    int has_weights;
} CsrArray;

typedef struct { /* src/graph_csr.h:37-42; mn_csr_delta has this layout */
    int32_t src_idx;
    int32_t dst_idx;
    double weight;
    int op;
} CsrDelta;

_Static_assert(sizeof(CsrDelta) == sizeof(mn_csr_delta), "CsrDelta and mn_csr_delta must have one layout");

int csr_apply_delta(const CsrArray *old_csr, const CsrDelta *deltas, int delta_count, int32_t new_node_count, CsrArray *new_csr) {
    memset(new_csr, 0, sizeof(CsrArray));
    if (new_node_count < old_csr->node_count) /* :179-180 */
        new_node_count = old_csr->node_count;
    int32_t *offsets = (int32_t *)malloc(((size_t)new_node_count + 1) * sizeof(int32_t));
    if (!offsets)
        return -1;
    int *targets = NULL, n_edges = 0;
    double *weights = NULL;
    static const int32_t zero_off[1] = {0};
    if (mn_csr_apply_delta(old_csr->node_count, old_csr->node_count ? old_csr->offsets : zero_off, old_csr->targets,
                           old_csr->weights, old_csr->has_weights, (const mn_csr_delta *)deltas, delta_count, new_node_count,
                           mn_env_device(), offsets, &targets, old_csr->has_weights ? &weights : NULL, &n_edges) != 0) {
        free(offsets);
        mn_host_free(targets); /* (the merge may have handed these out before it failed) */
        mn_host_free(weights);
        return -1; /* as the reference on allocation failure (:311-324); the caller rolls its savepoint back */
    }
    new_csr->node_count = new_node_count;
    new_csr->edge_count = n_edges;
    new_csr->offsets = offsets;
    new_csr->targets = targets; /* NULL when the result has no edges, as :277-281 */
    new_csr->weights = weights;
    new_csr->has_weights = old_csr->has_weights;
    return 0;
}
