/*
 * mn_graph_oracle.c — CPU oracle for the graph half of the hot path: Leiden local-move /
 * refinement / modularity (src/graph_community.c:75-429) over CSR adjacency (src/graph_csr.h:27-34).
 * TEST INFRASTRUCTURE ONLY (see mn_oracle.h).
 *
 * The reference runs on GraphData adjacency lists (out[] and in[] of {int target; double weight},
 * src/graph_load.h:13-37); here the same two lists are given as CSR (offsets/targets/weights) in
 * the same per-node edge order, so every f64 sum is taken in the reference's order.
 * Build: gcc -O2 -std=c11 -ffp-contract=off.
 */
#include "mn_graph_oracle.h"

#include <stdlib.h>
#include <string.h>

#define ORC_LEIDEN_MAX_SWEEPS 10000 /* batched schedule only; Q increases strictly, so this is a backstop */

static double ew(const double *w, int e) {
    return w ? w[e] : 1.0;
}

/* src/graph_community.c:75-90 */
static double w2c(const orc_graph *g, int v, const int *community, int target, int use_both) {
    double sum = 0.0;
    for (int e = g->off_out[v]; e < g->off_out[v + 1]; e++)
        if (community[g->tgt_out[e]] == target)
            sum += ew(g->w_out, e);
    if (use_both)
        for (int e = g->off_in[v]; e < g->off_in[v + 1]; e++)
            if (community[g->tgt_in[e]] == target)
                sum += ew(g->w_in, e);
    return sum;
}

/* src/graph_community.c:95-104 */
static double wdeg(const orc_graph *g, int v, int use_both) {
    double k = 0.0;
    for (int e = g->off_out[v]; e < g->off_out[v + 1]; e++)
        k += ew(g->w_out, e);
    if (use_both)
        for (int e = g->off_in[v]; e < g->off_in[v + 1]; e++)
            k += ew(g->w_in, e);
    return k;
}

/* src/graph_community.c:109-142 */
double orc_modularity(const orc_graph *g, const int *community, double resolution, double m, int use_both) {
    int N = g->n;
    if (m <= 0)
        return 0.0;
    int max_comm = 0;
    for (int i = 0; i < N; i++)
        if (community[i] > max_comm)
            max_comm = community[i];
    int nc = max_comm + 1;
    double *sum_in = (double *)calloc((size_t)nc, sizeof(double));
    double *sum_tot = (double *)calloc((size_t)nc, sizeof(double));
    for (int i = 0; i < N; i++) {
        int c = community[i];
        sum_tot[c] += wdeg(g, i, use_both);
        sum_in[c] += w2c(g, i, community, c, use_both);
    }
    double Q = 0.0;
    for (int c = 0; c < nc; c++)
        if (sum_tot[c] > 0)
            Q += sum_in[c] / (2.0 * m) - resolution * (sum_tot[c] / (2.0 * m)) * (sum_tot[c] / (2.0 * m));
    free(sum_in);
    free(sum_tot);
    return Q;
}

/* best move of node v against the given state (the loop body of :158-216).  `elig` restricts the
 * candidate edges (refinement: partition[w] == partition[v], :263-266); NULL = all edges. */
static int best_move(const orc_graph *g, int v, const int *label, const double *sum_tot, const double *k, double m,
                     double resolution, int use_both, const int *elig_part, int *scratch) {
    int old = label[v];
    double k_v = k[v];
    double k_v_to_old = w2c(g, v, label, old, use_both);
    int best = old;
    double best_gain = 0.0;
    int n_seen = 0;
    for (int pass = 0; pass < (use_both ? 2 : 1); pass++) {
        const int *off = pass ? g->off_in : g->off_out;
        const int *tgt = pass ? g->tgt_in : g->tgt_out;
        for (int e = off[v]; e < off[v + 1]; e++) {
            int w = tgt[e];
            if (elig_part && elig_part[w] != elig_part[v])
                continue;
            int nc = label[w];
            int seen = 0;
            for (int j = 0; j < n_seen; j++)
                if (scratch[j] == nc) {
                    seen = 1;
                    break;
                }
            if (seen)
                continue; /* a repeated candidate yields the same gain, never strictly greater */
            scratch[n_seen++] = nc;
            if (nc == old)
                continue;
            double k_v_to_t = w2c(g, v, label, nc, use_both);
            double gain = (k_v_to_t - k_v_to_old) / m + resolution * k_v * (sum_tot[old] - k_v - sum_tot[nc]) / (2.0 * m * m);
            if (gain > best_gain) {
                best_gain = gain;
                best = nc;
            }
        }
    }
    return best;
}

static int max_degree(const orc_graph *g, int use_both) {
    int md = 0;
    for (int v = 0; v < g->n; v++) {
        int d = g->off_out[v + 1] - g->off_out[v] + (use_both ? g->off_in[v + 1] - g->off_in[v] : 0);
        if (d > md)
            md = d;
    }
    return md;
}

/* One parallel round of the batch-synchronous schedule (HIP fast mode) over nodes
 * [b, e): every node is evaluated against the frozen state; a mover v commits iff it is the
 * smallest-index mover among (i) the movers touching its old or its target community and (ii) its
 * moving neighbours.  Committed moves touch pairwise disjoint communities and have unchanged
 * inputs, so each realises exactly its computed positive gain (Q strictly increases: no swap
 * cycles) and the round's result is independent of execution order.  Returns moves committed. */
static int batch_round(const orc_graph *g, int b, int e, int *label, double *sum_tot, const double *k, double m,
                       double resolution, int use_both, const int *elig_part, int *scratch, int *dec, int *cmin) {
    int moves = 0;
    for (int v = b; v < e; v++)
        dec[v - b] = best_move(g, v, label, sum_tot, k, m, resolution, use_both, elig_part, scratch);
    for (int v = b; v < e; v++) {
        int old = label[v], best = dec[v - b];
        if (best == old)
            continue;
        if (cmin[old] < 0 || v < cmin[old])
            cmin[old] = v;
        if (cmin[best] < 0 || v < cmin[best])
            cmin[best] = v;
    }
    for (int v = b; v < e; v++) {
        int old = label[v], best = dec[v - b];
        if (best == old)
            continue;
        int win = cmin[old] == v && cmin[best] == v;
        for (int pass = 0; win && pass < (use_both ? 2 : 1); pass++) {
            const int *off = pass ? g->off_in : g->off_out;
            const int *tgt = pass ? g->tgt_in : g->tgt_out;
            for (int x = off[v]; x < off[v + 1]; x++) {
                int w = tgt[x];
                if (w >= b && w < v && dec[w - b] != label[w]) {
                    win = 0;
                    break;
                }
            }
        }
        if (!win)
            dec[v - b] = -1 - best; /* loser: remembered only to clear cmin below */
    }
    for (int v = b; v < e; v++) {
        int old = label[v], d = dec[v - b];
        int best = d < 0 ? -1 - d : d;
        if (best == old)
            continue;
        cmin[old] = -1;
        cmin[best] = -1;
        if (d >= 0) {
            sum_tot[old] -= k[v];
            sum_tot[best] += k[v];
            label[v] = best;
            moves++;
        }
    }
    return moves;
}

/* src/graph_community.c:150-231.  batch <= 1: the reference's Gauss-Seidel sweep.  batch > 1:
 * sweeps of batch_round over consecutive node ranges until a whole sweep commits nothing. */
static int local_moving(const orc_graph *g, int *community, double *sum_tot, const double *k, double m, double resolution,
                        int use_both, int batch, int max_sweeps, int *scratch, int64_t *n_sweeps) {
    int N = g->n, total = 0, improved = 1, sweeps = 0;
    int *dec = batch > 1 ? (int *)malloc((size_t)batch * sizeof(int)) : NULL;
    int *cmin = NULL;
    if (batch > 1) {
        cmin = (int *)malloc((size_t)N * sizeof(int));
        for (int i = 0; i < N; i++)
            cmin[i] = -1;
    }
    while (improved && (batch <= 1 || sweeps < max_sweeps)) {
        improved = 0;
        sweeps++;
        if (batch <= 1) {
            for (int v = 0; v < N; v++) {
                int old = community[v];
                int best = best_move(g, v, community, sum_tot, k, m, resolution, use_both, NULL, scratch);
                if (best != old) { /* :220-227 */
                    sum_tot[old] -= k[v];
                    sum_tot[best] += k[v];
                    community[v] = best;
                    improved = 1;
                    total++;
                }
            }
        } else {
            for (int b = 0; b < N; b += batch) {
                int mv = batch_round(g, b, b + batch < N ? b + batch : N, community, sum_tot, k, m, resolution, use_both,
                                     NULL, scratch, dec, cmin);
                if (mv)
                    improved = 1;
                total += mv;
            }
        }
    }
    free(dec);
    free(cmin);
    if (n_sweeps)
        *n_sweeps += sweeps;
    return total;
}

/* src/graph_community.c:238-312 */
static void refinement(const orc_graph *g, const int *partition, int *refined, const double *k, double m, double resolution,
                       int use_both, int batch, int max_sweeps, int *scratch, int64_t *n_sweeps) {
    int N = g->n;
    double *r_sum_tot = (double *)malloc((size_t)N * sizeof(double));
    for (int i = 0; i < N; i++) {
        refined[i] = i;
        r_sum_tot[i] = k[i];
    }
    int improved = 1, sweeps = 0;
    int *dec = NULL, *cmin = NULL;
    if (batch > 1) {
        dec = (int *)malloc((size_t)batch * sizeof(int));
        cmin = (int *)malloc((size_t)N * sizeof(int));
        for (int i = 0; i < N; i++)
            cmin[i] = -1;
    }
    while (improved && (batch <= 1 || sweeps < max_sweeps)) {
        improved = 0;
        sweeps++;
        if (n_sweeps)
            (*n_sweeps)++;
        if (batch > 1) {
            for (int b = 0; b < N; b += batch)
                if (batch_round(g, b, b + batch < N ? b + batch : N, refined, r_sum_tot, k, m, resolution, use_both, partition,
                                scratch, dec, cmin))
                    improved = 1;
            continue;
        }
        for (int v = 0; v < N; v++) {
            int old = refined[v];
            int best = best_move(g, v, refined, r_sum_tot, k, m, resolution, use_both, partition, scratch);
            if (best != old) {
                r_sum_tot[old] -= k[v];
                r_sum_tot[best] += k[v];
                refined[v] = best;
                improved = 1;
            }
        }
    }
    free(r_sum_tot);
    free(dec);
    free(cmin);
}

/* src/graph_community.c:317-331 */
static int renumber(int *community, int N) {
    int *map = (int *)malloc((size_t)N * sizeof(int));
    for (int i = 0; i < N; i++)
        map[i] = -1;
    int next = 0;
    for (int i = 0; i < N; i++) {
        if (map[community[i]] == -1)
            map[community[i]] = next++;
        community[i] = map[community[i]];
    }
    free(map);
    return next;
}

/* src/graph_community.c:336-429 */
double orc_leiden(const orc_graph *g, int *community, double resolution, int use_both, int batch, orc_leiden_stats *st) {
    int N = g->n;
    if (st)
        memset(st, 0, sizeof(*st));
    if (N == 0)
        return 0.0;
    double *k = (double *)malloc((size_t)N * sizeof(double));
    double m = 0.0;
    for (int i = 0; i < N; i++) {
        k[i] = wdeg(g, i, use_both);
        m += k[i];
    }
    m /= 2.0;
    for (int i = 0; i < N; i++)
        community[i] = i;
    if (m <= 0.0) {
        free(k);
        return 0.0;
    }
    double *sum_tot = (double *)malloc((size_t)N * sizeof(double));
    memcpy(sum_tot, k, (size_t)N * sizeof(double));
    int *refined = (int *)malloc((size_t)N * sizeof(int));
    int *scratch = (int *)malloc(((size_t)max_degree(g, use_both) + 1) * sizeof(int));
    unsigned char *seen = (unsigned char *)malloc((size_t)N);
    for (int iter = 0; iter < 100; iter++) {
        int moves = local_moving(g, community, sum_tot, k, m, resolution, use_both, batch, ORC_LEIDEN_MAX_SWEEPS, scratch, st ? &st->move_sweeps : NULL);
        if (st) {
            st->iterations++;
            st->moves += moves;
        }
        if (moves == 0)
            break;
        refinement(g, community, refined, k, m, resolution, use_both, batch, ORC_LEIDEN_MAX_SWEEPS, scratch, st ? &st->refine_sweeps : NULL);
        int p1 = 0, rf = 0; /* :388-403 */
        memset(seen, 0, (size_t)N);
        for (int i = 0; i < N; i++)
            if (!seen[community[i]]) {
                seen[community[i]] = 1;
                p1++;
            }
        memset(seen, 0, (size_t)N);
        for (int i = 0; i < N; i++)
            if (!seen[refined[i]]) {
                seen[refined[i]] = 1;
                rf++;
            }
        if (rf <= p1) /* :406-408 */
            memcpy(community, refined, (size_t)N * sizeof(int));
        renumber(community, N);
        memset(sum_tot, 0, (size_t)N * sizeof(double));
        for (int i = 0; i < N; i++)
            sum_tot[community[i]] += k[i];
    }
    renumber(community, N);
    double Q = orc_modularity(g, community, resolution, m, use_both);
    free(k);
    free(sum_tot);
    free(refined);
    free(scratch);
    free(seen);
    return Q;
}
