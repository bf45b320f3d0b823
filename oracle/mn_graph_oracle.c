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

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_LEIDEN_MAX_SWEEPS 100000 /* batched schedule only; Q increases strictly, so this is a backstop */

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
                     double resolution, int use_both, const int *elig_part, int *scratch, double *dk_out, int pickless) {
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
            if (pickless && nc > old)
                continue; /* synchronous schedule, every ORC_LEI_PICKLESS-th sweep: only towards a smaller community id */
            double k_v_to_t = w2c(g, v, label, nc, use_both);
            double gain = (k_v_to_t - k_v_to_old) / m + resolution * k_v * (sum_tot[old] - k_v - sum_tot[nc]) / (2.0 * m * m);
            if (gain > best_gain) {
                best_gain = gain;
                best = nc;
                if (dk_out)
                    *dk_out = k_v_to_t - k_v_to_old;
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

/* Fixed-point upper bound of a weighted degree: tallies of movers are kept as integers so that they
 * are exact and order-independent on the device (integer atomics). */
#define ORC_FX 1048576.0
static unsigned long long fx_up(double k) {
    return (unsigned long long)ceil(k * ORC_FX);
}

/* One parallel round of the batch-synchronous schedule (HIP fast mode) over nodes
 * [b, e).  Every node is evaluated against the frozen state (best_move).  Let J[c] / L[c] be the total
 * degree of movers that want to join / leave community c.  A mover v (old → c) is a SAFE winner iff no
 * neighbour with a smaller index inside [b, e) is a mover, and its gain stays positive in the worst
 * order of application: old already drained by every other leaver, c already filled by every other
 * joiner.  Then every applied move has positive gain whatever the order → Q strictly increases.  If a
 * round has movers but no safe winner, the STRICT rule is used instead: v wins iff it is the smallest
 * index among the movers touching old or c (and has no smaller moving neighbour).  Winners are applied
 * in node order.  Returns the number of moves. */
#define ORC_LEI_GROW 4
#define ORC_LEI_GROW_DIV 256
typedef struct {
    int *dec;
    double *dk;
    unsigned char *win;
    int *cmin;                    /* [N] = INT_MAX when idle */
    unsigned long long *Jq, *Lq;  /* [N] zero when idle */
} round_ws;

static int batch_round(const orc_graph *g, int b, int e, int *label, double *sum_tot, const double *k, double m,
                       double resolution, int use_both, const int *elig_part, int *scratch, round_ws *ws) {
    int movers = 0, safe = 0, moves = 0;
    for (int v = b; v < e; v++) {
        ws->dk[v - b] = 0.0;
        ws->dec[v - b] = best_move(g, v, label, sum_tot, k, m, resolution, use_both, elig_part, scratch, &ws->dk[v - b], 0);
    }
    for (int v = b; v < e; v++) {
        int old = label[v], best = ws->dec[v - b];
        if (best == old)
            continue;
        movers++;
        if (v < ws->cmin[old]) ws->cmin[old] = v;
        if (v < ws->cmin[best]) ws->cmin[best] = v;
        ws->Lq[old] += fx_up(k[v]);
        ws->Jq[best] += fx_up(k[v]);
    }
    for (int v = b; v < e; v++) {
        int old = label[v], best = ws->dec[v - b];
        ws->win[v - b] = 0;
        if (best == old)
            continue;
        int free_nb = 1;
        for (int pass = 0; free_nb && pass < (use_both ? 2 : 1); pass++) {
            const int *off = pass ? g->off_in : g->off_out;
            const int *tgt = pass ? g->tgt_in : g->tgt_out;
            for (int x = off[v]; x < off[v + 1]; x++) {
                int w = tgt[x];
                if (w >= b && w < v && ws->dec[w - b] != label[w]) {
                    free_nb = 0;
                    break;
                }
            }
        }
        if (!free_nb)
            continue;
        double Lo = (double)(ws->Lq[old] - fx_up(k[v])) / ORC_FX, Jc = (double)(ws->Jq[best] - fx_up(k[v])) / ORC_FX;
        double gain2 = ws->dk[v - b] / m + resolution * k[v] * (sum_tot[old] - Lo - k[v] - sum_tot[best] - Jc) / (2.0 * m * m);
        int strict = ws->cmin[old] == v && ws->cmin[best] == v;
        ws->win[v - b] = (unsigned char)((gain2 > 0.0 ? 1 : 0) | (strict ? 2 : 0));
        if (gain2 > 0.0)
            safe++;
    }
    const int use_bit = safe > 0 ? 1 : 2;
    for (int v = b; v < e; v++) {
        int old = label[v], best = ws->dec[v - b];
        if (best == old)
            continue;
        ws->cmin[old] = 0x7fffffff;
        ws->cmin[best] = 0x7fffffff;
        ws->Lq[old] = 0;
        ws->Jq[best] = 0;
    }
    for (int v = b; v < e; v++) {
        int old = label[v], best = ws->dec[v - b];
        if (best != old && (ws->win[v - b] & use_bit)) {
            sum_tot[old] -= k[v];
            sum_tot[best] += k[v];
            label[v] = best;
            moves++;
        }
    }
    (void)movers;
    return moves;
}

static void ws_init(round_ws *ws, int N, int batch) {
    ws->dec = (int *)malloc((size_t)batch * sizeof(int));
    ws->dk = (double *)malloc((size_t)batch * sizeof(double));
    ws->win = (unsigned char *)malloc((size_t)batch);
    ws->cmin = (int *)malloc((size_t)N * sizeof(int));
    ws->Jq = (unsigned long long *)calloc((size_t)N, sizeof(unsigned long long));
    ws->Lq = (unsigned long long *)calloc((size_t)N, sizeof(unsigned long long));
    for (int i = 0; i < N; i++)
        ws->cmin[i] = 0x7fffffff;
}
static void ws_free(round_ws *ws) {
    free(ws->dec); free(ws->dk); free(ws->win); free(ws->cmin); free(ws->Jq); free(ws->Lq);
}

/* sweeps of batch_round over consecutive node ranges until a whole sweep commits nothing */
static int batched_phase(const orc_graph *g, int *label, double *sum_tot, const double *k, double m, double resolution,
                         int use_both, const int *elig_part, int batch, int max_sweeps, int *scratch, int64_t *n_sweeps) {
    int N = g->n, total = 0, improved = 1, sweeps = 0;
    round_ws ws;
    /* tail rule of the schedule: rounds ORC_LEI_GROW times larger once the sweep before the previous one committed fewer
     * than N / ORC_LEI_GROW_DIV moves (the device queues a sweep before the previous sweep's count has reached the host) */
    const int batch0 = batch;
    int G_ = ORC_LEI_GROW, T_ = ORC_LEI_GROW_DIV;
    if (getenv("ORC_LEI_GROW")) sscanf(getenv("ORC_LEI_GROW"), "%d,%d", &G_, &T_);
    long long big = (long long)batch0 * G_;
    if (big > (N > batch0 ? N : batch0))
        big = N > batch0 ? N : batch0;
    int moves_prev = -1, moves_prev2 = -1;
    ws_init(&ws, N, (int)big);
    while (improved && sweeps < max_sweeps) {
        improved = 0;
        sweeps++;
        batch = (moves_prev2 >= 0 && moves_prev2 < N / T_) ? (int)big : batch0;
        int tr_rounds = 0, tr_moves = 0;
        for (int b = 0; b < N; b += batch) {
            int mv = batch_round(g, b, b + batch < N ? b + batch : N, label, sum_tot, k, m, resolution, use_both, elig_part,
                                 scratch, &ws);
            if (mv)
                improved = 1, tr_rounds++;
            total += mv;
            tr_moves += mv;
        }
        if (getenv("ORC_LEIDEN_TRACE"))
            fprintf(stderr, "sweep %d%s: rounds of %d: %d moves in %d rounds\n", sweeps, elig_part ? " (refine)" : "", batch, tr_moves, tr_rounds);
        moves_prev2 = moves_prev;
        moves_prev = tr_moves;
    }
    ws_free(&ws);
    if (n_sweeps)
        *n_sweeps += sweeps;
    return total;
}

/* The default schedule of the HIP fast mode since round 4: WHOLE-GRAPH synchronous sweeps.  A sweep
 * evaluates every node against the state frozen at its start (best_move) and then applies EVERY positive-gain mover, in
 * node order (each "sum_tot[old] -= k" then "sum_tot[new] += k", :220-223).  Simultaneous moves can swap two nodes for
 * ever; the device from the GPU Louvain literature (Naim et al., "pick-less") breaks such cycles: in every period-th sweep
 * a node may only move to a community with a SMALLER id.  A phase ends with the first ordinary (not pick-less) sweep that
 * moves nothing — then no node has a positive-gain move, which is also the sequential loop's fixed point (:154-229).
 * Modularity is not monotone under simultaneous moves, so termination is not guaranteed: after ORC_LEI_SYNC_CAP sweeps
 * the phase is finished by the round schedule above (batched_phase: every applied move gains, Q strictly increases).
 * Measured on config 5's graph (500k nodes / 9.27M edges): 16 + 16 sweeps, Q 0.67356, against 856 rounds in 27 sweeps
 * and Q 0.67188 for rounds of 15 625 nodes. */
#define ORC_LEI_PICKLESS 3
#define ORC_LEI_SYNC_CAP 48
static int orc_lei_round_default(int N) { /* mn_graph_leiden's default round size */
    long long b = N / 32;
    if (b < 256) b = 256;
    if (b > 16384) b = 16384;
    return (int)b;
}
static int sync_phase(const orc_graph *g, int *label, double *sum_tot, const double *k, double m, double resolution,
                      int use_both, const int *elig_part, int period, int max_sweeps, int *scratch, int64_t *n_sweeps) {
    int N = g->n, total = 0, sweeps = 0, converged = 0;
    int *dec = (int *)malloc((size_t)N * sizeof(int));
    int cap = ORC_LEI_SYNC_CAP;
    if (getenv("ORC_LEI_SYNC_CAP")) cap = atoi(getenv("ORC_LEI_SYNC_CAP"));
    while (sweeps < cap && sweeps < max_sweeps) {
        sweeps++;
        const int pickless = period > 0 && sweeps % period == 0;
        int moves = 0;
        for (int v = 0; v < N; v++)
            dec[v] = best_move(g, v, label, sum_tot, k, m, resolution, use_both, elig_part, scratch, NULL, pickless);
        for (int v = 0; v < N; v++)
            if (dec[v] != label[v]) {
                sum_tot[label[v]] -= k[v];
                sum_tot[dec[v]] += k[v];
                label[v] = dec[v];
                moves++;
            }
        if (getenv("ORC_LEIDEN_TRACE"))
            fprintf(stderr, "sync sweep %d%s%s: %d moves\n", sweeps, elig_part ? " (refine)" : "", pickless ? " (pick-less)" : "", moves);
        total += moves;
        if (moves == 0 && !pickless) {
            converged = 1;
            break;
        }
    }
    free(dec);
    if (n_sweeps)
        *n_sweeps += sweeps;
    if (!converged)
        total += batched_phase(g, label, sum_tot, k, m, resolution, use_both, elig_part, orc_lei_round_default(N), max_sweeps, scratch, n_sweeps);
    return total;
}

/* src/graph_community.c:150-231.  mode 0: the reference's Gauss-Seidel sweep; mode 1: sync_phase. */
static int local_moving(const orc_graph *g, int *community, double *sum_tot, const double *k, double m, double resolution,
                        int use_both, int batch, int max_sweeps, int *scratch, int64_t *n_sweeps) {
    if (batch > 1)
        return batched_phase(g, community, sum_tot, k, m, resolution, use_both, NULL, batch, max_sweeps, scratch, n_sweeps);
    if (batch < 0)
        return sync_phase(g, community, sum_tot, k, m, resolution, use_both, NULL, -batch, max_sweeps, scratch, n_sweeps);
    int N = g->n, total = 0, improved = 1, sweeps = 0;
    while (improved) {
        improved = 0;
        sweeps++;
        for (int v = 0; v < N; v++) {
            int old = community[v];
            int best = best_move(g, v, community, sum_tot, k, m, resolution, use_both, NULL, scratch, NULL, 0);
            if (best != old) { /* :220-227 */
                sum_tot[old] -= k[v];
                sum_tot[best] += k[v];
                community[v] = best;
                improved = 1;
                total++;
            }
        }
    }
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
    if (batch > 1 || batch < 0) {
        if (batch > 1)
            batched_phase(g, refined, r_sum_tot, k, m, resolution, use_both, partition, batch, max_sweeps, scratch, n_sweeps);
        else
            sync_phase(g, refined, r_sum_tot, k, m, resolution, use_both, partition, -batch, max_sweeps, scratch, n_sweeps);
        free(r_sum_tot);
        return;
    }
    int improved = 1;
    while (improved) {
        improved = 0;
        if (n_sweeps)
            (*n_sweeps)++;
        for (int v = 0; v < N; v++) {
            int old = refined[v];
            int best = best_move(g, v, refined, r_sum_tot, k, m, resolution, use_both, partition, scratch, NULL, 0);
            if (best != old) {
                r_sum_tot[old] -= k[v];
                r_sum_tot[best] += k[v];
                refined[v] = best;
                improved = 1;
            }
        }
    }
    free(r_sum_tot);
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

/* ───────────────────────── a14-a17: Node2Vec (src/node2vec.c) ───────────────────────── */

static unsigned n2v_xorshift32(unsigned *state) { /* :27-34 */
    unsigned x = *state;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    *state = x;
    return x;
}

static double n2v_rand(unsigned *state) { /* :36-38 */
    return (double)n2v_xorshift32(state) / (double)0xFFFFFFFFu;
}

int orc_n2v_build_graph(int n_edges, const int *src, const int *dst, int n_ids, int *off, int *adj, int *index_of_id) {
    /* graph_load_edges (:112-138): node index = first appearance (src before dst), both directions, duplicates dropped */
    int *idx = (int *)malloc((size_t)(n_ids > 0 ? n_ids : 1) * sizeof(int));
    for (int i = 0; i < n_ids; i++)
        idx[i] = -1;
    int n = 0;
    int *es = (int *)malloc((size_t)(n_edges > 0 ? n_edges : 1) * sizeof(int));
    int *ed = (int *)malloc((size_t)(n_edges > 0 ? n_edges : 1) * sizeof(int));
    for (int e = 0; e < n_edges; e++) {
        if (idx[src[e]] < 0)
            idx[src[e]] = n++;
        if (idx[dst[e]] < 0)
            idx[dst[e]] = n++;
        es[e] = idx[src[e]];
        ed[e] = idx[dst[e]];
    }
    /* per-node growable lists with the reference's duplicate check (:96-109) */
    int **lst = (int **)calloc((size_t)(n > 0 ? n : 1), sizeof(int *));
    int *cnt = (int *)calloc((size_t)(n > 0 ? n : 1), sizeof(int));
    int *cap = (int *)calloc((size_t)(n > 0 ? n : 1), sizeof(int));
    for (int e = 0; e < n_edges; e++)
        for (int dir = 0; dir < 2; dir++) {
            int a = dir ? ed[e] : es[e], b = dir ? es[e] : ed[e];
            int dup = 0;
            for (int i = 0; i < cnt[a]; i++)
                if (lst[a][i] == b) {
                    dup = 1;
                    break;
                }
            if (dup)
                continue;
            if (cnt[a] >= cap[a]) {
                cap[a] = cap[a] ? cap[a] * 2 : 8;
                lst[a] = (int *)realloc(lst[a], (size_t)cap[a] * sizeof(int));
            }
            lst[a][cnt[a]++] = b;
        }
    off[0] = 0;
    for (int i = 0; i < n; i++) {
        memcpy(adj + off[i], lst[i], (size_t)cnt[i] * sizeof(int));
        off[i + 1] = off[i] + cnt[i];
        free(lst[i]);
    }
    if (index_of_id)
        memcpy(index_of_id, idx, (size_t)n_ids * sizeof(int));
    free(lst);
    free(cnt);
    free(cap);
    free(idx);
    free(es);
    free(ed);
    return n;
}

static int n2v_is_neighbor(const orc_n2v_graph *g, int node, int target) { /* :154-161 */
    for (int i = g->off[node]; i < g->off[node + 1]; i++)
        if (g->adj[i] == target)
            return 1;
    return 0;
}

int orc_biased_walk(const orc_n2v_graph *g, int start, double p, double q, int walk_length, int *walk, unsigned *rng) {
    walk[0] = start; /* :168-226 */
    int deg0 = g->off[start + 1] - g->off[start];
    if (deg0 == 0)
        return 1;
    int idx = (int)(n2v_rand(rng) * deg0);
    if (idx >= deg0)
        idx = deg0 - 1;
    walk[1] = g->adj[g->off[start] + idx];
    for (int step = 2; step < walk_length; step++) {
        int cur = walk[step - 1], prev = walk[step - 2];
        int c0 = g->off[cur], deg = g->off[cur + 1] - c0;
        if (deg == 0)
            return step;
        double total = 0.0;
        for (int i = 0; i < deg; i++) {
            int x = g->adj[c0 + i];
            double w = (x == prev) ? 1.0 / p : (n2v_is_neighbor(g, prev, x) ? 1.0 : 1.0 / q);
            total += w;
        }
        double r = n2v_rand(rng) * total;
        double cum = 0.0;
        int chosen = g->adj[c0];
        for (int i = 0; i < deg; i++) {
            int x = g->adj[c0 + i];
            double w = (x == prev) ? 1.0 / p : (n2v_is_neighbor(g, prev, x) ? 1.0 : 1.0 / q);
            cum += w;
            if (r <= cum) {
                chosen = x;
                break;
            }
        }
        walk[step] = chosen;
    }
    return walk_length;
}

#define N2V_SIG_SIZE 1000   /* :241 */
#define N2V_MAX_SIG 6.0f    /* :242 */
#define N2V_NEG_TABLE 100000 /* :274 */

static float n2v_sig[N2V_SIG_SIZE + 1];
static int n2v_sig_ready = 0;

static void n2v_init_sig(void) { /* :247-258 */
    if (n2v_sig_ready)
        return;
    for (int i = 0; i <= N2V_SIG_SIZE; i++) {
        float x = (float)i / (float)N2V_SIG_SIZE * 2.0f * N2V_MAX_SIG - N2V_MAX_SIG;
        n2v_sig[i] = 1.0f / (1.0f + expf(-x));
    }
    n2v_sig_ready = 1;
}

static float n2v_fast_sigmoid(float x) { /* :260-271 */
    if (x >= N2V_MAX_SIG)
        return 1.0f;
    if (x <= -N2V_MAX_SIG)
        return 0.0f;
    int idx = (int)((x + N2V_MAX_SIG) / (2.0f * N2V_MAX_SIG) * N2V_SIG_SIZE);
    if (idx < 0)
        idx = 0;
    if (idx > N2V_SIG_SIZE)
        idx = N2V_SIG_SIZE;
    return n2v_sig[idx];
}

const float *orc_n2v_sigmoid_table(void) {
    n2v_init_sig();
    return n2v_sig;
}

void orc_n2v_neg_table(const orc_n2v_graph *g, int *table) { /* build_neg_table :284-303 */
    int N = g->n;
    double total = 0.0;
    for (int i = 0; i < N; i++)
        total += pow((double)(g->off[i + 1] - g->off[i] + 1), 0.75);
    int idx = 0;
    double cum = 0.0;
    for (int i = 0; i < N && idx < N2V_NEG_TABLE; i++) {
        cum += pow((double)(g->off[i + 1] - g->off[i] + 1), 0.75) / total;
        while (idx < N2V_NEG_TABLE && (double)idx / N2V_NEG_TABLE < cum)
            table[idx++] = i;
    }
    while (idx < N2V_NEG_TABLE)
        table[idx++] = N - 1;
}

/* sgns_train_pair (:345-394) */
static void n2v_train_pair(float *syn0, float *syn1neg, const int *neg_table, int dim, int center, int context, int neg,
                           float lr, unsigned *rng, float *neu1e) {
    float *vc = syn0 + (size_t)center * dim;
    memset(neu1e, 0, (size_t)dim * sizeof(float));
    for (int s = 0; s <= neg; s++) {
        int target;
        float label;
        if (s == 0) {
            target = context;
            label = 1.0f;
        } else {
            target = neg_table[n2v_xorshift32(rng) % N2V_NEG_TABLE];
            if (target == center || target == context)
                continue;
            label = 0.0f;
        }
        float *vt = syn1neg + (size_t)target * dim;
        float dot = 0.0f;
        for (int d = 0; d < dim; d++)
            dot += vc[d] * vt[d];
        float sig = n2v_fast_sigmoid(dot);
        float err = (label - sig) * lr;
        for (int d = 0; d < dim; d++)
            neu1e[d] += err * vt[d];
        for (int d = 0; d < dim; d++)
            vt[d] += err * vc[d];
    }
    for (int d = 0; d < dim; d++)
        vc[d] += neu1e[d];
}

int orc_node2vec_train(const orc_n2v_graph *g, const orc_n2v_params *p, float *out, int64_t *n_pairs) {
    int N = g->n, dim = p->dim;
    if (N == 0)
        return 0;
    n2v_init_sig();
    unsigned rng = 42; /* :486 */
    float *syn0 = out;
    float *syn1neg = (float *)calloc((size_t)N * dim, sizeof(float));
    int *neg_table = (int *)malloc(N2V_NEG_TABLE * sizeof(int));
    float *neu1e = (float *)malloc((size_t)dim * sizeof(float));
    int *walk = (int *)malloc((size_t)p->walk_length * sizeof(int));
    for (int i = 0; i < N * dim; i++) /* :323-325 */
        syn0[i] = ((float)n2v_rand(&rng) - 0.5f) / (float)dim;
    orc_n2v_neg_table(g, neg_table);
    int total_words = N * p->num_walks * p->walk_length * p->epochs; /* :503 (32-bit, as the reference) */
    int word_count = 0;
    int64_t pairs = 0;
    for (int epoch = 0; epoch < p->epochs; epoch++)
        for (int w = 0; w < p->num_walks; w++)
            for (int n = 0; n < N; n++) {
                float lr = (float)(p->lr * (1.0 - (double)word_count / (double)total_words)); /* :510-512 */
                if (lr < (float)(p->lr * 0.0001))
                    lr = (float)(p->lr * 0.0001);
                int wlen = orc_biased_walk(g, n, p->p, p->q, p->walk_length, walk, &rng);
                for (int pos = 0; pos < wlen; pos++) {
                    int cs = pos - p->window, ce = pos + p->window;
                    if (cs < 0)
                        cs = 0;
                    if (ce >= wlen)
                        ce = wlen - 1;
                    for (int c = cs; c <= ce; c++) {
                        if (c == pos)
                            continue;
                        n2v_train_pair(syn0, syn1neg, neg_table, dim, walk[pos], walk[c], p->neg_samples, lr, &rng, neu1e);
                        pairs++;
                    }
                    word_count++;
                }
            }
    for (int i = 0; i < N; i++) { /* :540-551 */
        float *emb = syn0 + (size_t)i * dim;
        float norm = 0.0f;
        for (int d = 0; d < dim; d++)
            norm += emb[d] * emb[d];
        norm = sqrtf(norm);
        if (norm > 1e-10f)
            for (int d = 0; d < dim; d++)
                emb[d] /= norm;
    }
    if (n_pairs)
        *n_pairs = pairs;
    free(syn1neg);
    free(neg_table);
    free(neu1e);
    free(walk);
    return N;
}

/* ───────────── batch-synchronous Node2Vec schedule (HIP MN_N2V_BATCHED) ─────────────
 * Per (epoch, w) pass the start nodes are cut into batches of B walks.  Inside a batch:
 *  (1) walk n has its own xorshift32 stream seeded from (epoch, w, n); it draws the walk (biased_walk,
 *      unchanged) and then, in pair order, the negatives of its pairs;
 *  (2) every sample's error is computed against the matrices as they stood at batch start:
 *      err = (label - sigmoid_lut(dot)) * lr_walk, dot in wave order (lane L folds d ≡ L mod 64 with fmaf,
 *      xor butterfly 32..1);
 *  (3) centre side, as the reference's neu1e (src/node2vec.c:347,:383-391) but once per walk position: the
 *      position's neu1e = Σ err·syn1neg_old[t] over all its samples in sample order (fmaf from 0), then
 *      syn0[c] = syn0[c] + neu1e, positions applied in (walk, position) order;
 *      target side: syn1neg[t] += Σ err·syn0_old[c] over the batch's samples with target t, in sample order (fmaf).
 * Final L2 normalisation as the reference (:540-551). */
static unsigned n2v_walk_seed(int epoch, int w, int n) {
    unsigned s = 42u ^ ((unsigned)epoch * 0x9E3779B9u) ^ ((unsigned)w * 0x85EBCA6Bu) ^ ((unsigned)n * 0xC2B2AE35u);
    s ^= s >> 15;
    s *= 0x2C1B3C6Du;
    s ^= s >> 12;
    return s ? s : 1u;
}

static float n2v_dot_wave(const float *a, const float *b, int dim) {
    float part[64];
    for (int l = 0; l < 64; l++) {
        float acc = 0.0f;
        for (int d = l; d < dim; d += 64)
            acc = fmaf(a[d], b[d], acc);
        part[l] = acc;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        float nxt[64];
        for (int l = 0; l < 64; l++)
            nxt[l] = part[l] + part[l ^ m];
        memcpy(part, nxt, sizeof(nxt));
    }
    return part[0];
}

typedef struct {
    int center, target;
    float err;
} n2v_sample;

int orc_node2vec_train_batched(const orc_n2v_graph *g, const orc_n2v_params *p, int B, float *out, int64_t *n_pairs) {
    int N = g->n, dim = p->dim;
    if (N == 0)
        return 0;
    if (B < 1)
        B = 1;
    n2v_init_sig();
    unsigned rng = 42;
    float *syn0 = out;
    float *syn1 = (float *)calloc((size_t)N * dim, sizeof(float));
    float *old0 = (float *)malloc((size_t)N * dim * sizeof(float));
    float *old1 = (float *)malloc((size_t)N * dim * sizeof(float));
    int *neg_table = (int *)malloc(N2V_NEG_TABLE * sizeof(int));
    int *walk = (int *)malloc((size_t)p->walk_length * sizeof(int));
    float *neu = (float *)malloc((size_t)dim * sizeof(float));
    for (int i = 0; i < N * dim; i++)
        syn0[i] = ((float)n2v_rand(&rng) - 0.5f) / (float)dim;
    orc_n2v_neg_table(g, neg_table);
    const size_t cap = (size_t)p->walk_length * 2 * p->window * (1 + p->neg_samples);
    n2v_sample *smp = (n2v_sample *)malloc((size_t)B * cap * sizeof(n2v_sample));
    const double total_words = (double)N * p->num_walks * p->walk_length * p->epochs;
    int64_t pairs = 0;
    for (int epoch = 0; epoch < p->epochs; epoch++)
        for (int w = 0; w < p->num_walks; w++)
            for (int b = 0; b < N; b += B) {
                int e = b + B < N ? b + B : N;
                size_t ns = 0;
                memcpy(old0, syn0, (size_t)N * dim * sizeof(float));
                memcpy(old1, syn1, (size_t)N * dim * sizeof(float));
                for (int n = b; n < e; n++) {
                    unsigned st = n2v_walk_seed(epoch, w, n);
                    double wc = ((double)(epoch * p->num_walks + w) * N + n) * p->walk_length;
                    float lr = (float)(p->lr * (1.0 - wc / total_words));
                    if (lr < (float)(p->lr * 0.0001))
                        lr = (float)(p->lr * 0.0001);
                    int wlen = orc_biased_walk(g, n, p->p, p->q, p->walk_length, walk, &st);
                    for (int pos = 0; pos < wlen; pos++) {
                        int cs = pos - p->window, ce = pos + p->window;
                        if (cs < 0)
                            cs = 0;
                        if (ce >= wlen)
                            ce = wlen - 1;
                        memset(neu, 0, (size_t)dim * sizeof(float));
                        for (int c = cs; c <= ce; c++) {
                            if (c == pos)
                                continue;
                            int center = walk[pos], context = walk[c];
                            pairs++;
                            for (int s = 0; s <= p->neg_samples; s++) {
                                int target;
                                float label;
                                if (s == 0) {
                                    target = context;
                                    label = 1.0f;
                                } else {
                                    target = neg_table[n2v_xorshift32(&st) % N2V_NEG_TABLE];
                                    if (target == center || target == context)
                                        continue;
                                    label = 0.0f;
                                }
                                float dot = n2v_dot_wave(old0 + (size_t)center * dim, old1 + (size_t)target * dim, dim);
                                float err = (label - n2v_fast_sigmoid(dot)) * lr;
                                smp[ns].center = center;
                                smp[ns].target = target;
                                smp[ns].err = err;
                                ns++;
                                const float *t1 = old1 + (size_t)target * dim;
                                for (int d = 0; d < dim; d++) /* the position's neu1e (src/node2vec.c:383-385) */
                                    neu[d] = fmaf(err, t1[d], neu[d]);
                            }
                        }
                        /* centre side, once per position, positions in (walk, pos) order */
                        float *dc = syn0 + (size_t)walk[pos] * dim;
                        for (int d = 0; d < dim; d++)
                            dc[d] = dc[d] + neu[d];
                    }
                }
                /* (3) target side: per-destination accumulation in sample order, from the old centres */
                for (size_t i = 0; i < ns; i++) {
                    float *dt = syn1 + (size_t)smp[i].target * dim;
                    const float *sc0 = old0 + (size_t)smp[i].center * dim;
                    for (int d = 0; d < dim; d++)
                        dt[d] = fmaf(smp[i].err, sc0[d], dt[d]);
                }
            }
    for (int i = 0; i < N; i++) {
        float *emb = syn0 + (size_t)i * dim;
        float norm = 0.0f;
        for (int d = 0; d < dim; d++)
            norm += emb[d] * emb[d];
        norm = sqrtf(norm);
        if (norm > 1e-10f)
            for (int d = 0; d < dim; d++)
                emb[d] /= norm;
    }
    if (n_pairs)
        *n_pairs = pairs;
    free(syn1);
    free(old0);
    free(old1);
    free(neg_table);
    free(walk);
    free(neu);
    free(smp);
    return N;
}


/* ───────────────────────── f-4: PageRank, components (src/graph_tvf.c) ───────────────────────── */

int orc_pagerank(int n, int n_edges, const int *src, const int *dst, double damping, int iterations, double *rank_out) {
    if (n <= 0)
        return 0;
    /* pr_adj_add_edge (:1615-1622): out lists in row order */
    int *cnt = (int *)calloc((size_t)n, sizeof(int)), *off = (int *)calloc((size_t)n + 1, sizeof(int));
    int *tgt = (int *)malloc((size_t)(n_edges ? n_edges : 1) * sizeof(int)), *cur = (int *)malloc((size_t)n * sizeof(int));
    double *a = (double *)malloc((size_t)n * sizeof(double)), *b = (double *)malloc((size_t)n * sizeof(double));
    if (!cnt || !off || !tgt || !cur || !a || !b)
        return -1;
    for (int e = 0; e < n_edges; e++)
        cnt[src[e]]++;
    for (int i = 0; i < n; i++)
        off[i + 1] = off[i] + cnt[i];
    memcpy(cur, off, (size_t)n * sizeof(int));
    for (int e = 0; e < n_edges; e++)
        tgt[cur[src[e]]++] = dst[e];
    double init_rank = 1.0 / n; /* :1684-1686 */
    for (int i = 0; i < n; i++)
        a[i] = init_rank;
    double teleport = (1.0 - damping) / n; /* :1689 */
    for (int it = 0; it < iterations; it++) {
        for (int i = 0; i < n; i++)
            b[i] = teleport;
        for (int i = 0; i < n; i++) {
            if (cnt[i] == 0) { /* dangling: its rank goes to every node (:1694-1698) */
                double share = damping * a[i] / n;
                for (int j = 0; j < n; j++)
                    b[j] += share;
            } else {
                double share = damping * a[i] / cnt[i];
                for (int e = off[i]; e < off[i + 1]; e++)
                    b[tgt[e]] += share;
            }
        }
        double *t = a;
        a = b;
        b = t;
    }
    memcpy(rank_out, a, (size_t)n * sizeof(double));
    free(cnt); free(off); free(tgt); free(cur); free(a); free(b);
    return 0;
}

static int ouf_find(int *parent, int x) { /* :1249-1256 */
    while (parent[x] != x) {
        parent[x] = parent[parent[x]];
        x = parent[x];
    }
    return x;
}

int orc_components(int n, int n_edges, const int *src, const int *dst, int *component_id, int *component_size) {
    if (n <= 0)
        return 0;
    int *parent = (int *)malloc((size_t)n * sizeof(int)), *rank = (int *)calloc((size_t)n, sizeof(int));
    int *size = (int *)calloc((size_t)n, sizeof(int));
    if (!parent || !rank || !size)
        return -1;
    for (int i = 0; i < n; i++)
        parent[i] = i;
    for (int e = 0; e < n_edges; e++) { /* uf_union (:1258-1273) */
        int ra = ouf_find(parent, src[e]), rb = ouf_find(parent, dst[e]);
        if (ra == rb)
            continue;
        if (rank[ra] < rank[rb]) {
            int t = ra;
            ra = rb;
            rb = t;
        }
        parent[rb] = ra;
        if (rank[ra] == rank[rb])
            rank[ra]++;
    }
    for (int i = 0; i < n; i++)
        size[ouf_find(parent, i)]++;
    for (int i = 0; i < n; i++) {
        int r = ouf_find(parent, i);
        component_id[i] = r;
        component_size[i] = size[r];
    }
    free(parent); free(rank); free(size);
    return 0;
}


/* ───────────────────────── f-2: csr_apply_delta (src/graph_csr.c:175-325) ───────────────────────── */

int orc_csr_apply_delta(int old_n, const int *off, const int *tgt, const double *w, int has_weights, int nd, const int *dsrc,
                        const int *ddst, const double *dw, const int *dop, int new_n, int *new_off, int *new_tgt, double *new_w) {
    if (new_n < old_n)
        new_n = old_n; /* :179-180 */
    /* every node's list with room for all its inserts */
    int *cap = (int *)calloc((size_t)new_n + 1, sizeof(int)), *cnt = (int *)calloc((size_t)new_n + 1, sizeof(int));
    int *start = (int *)calloc((size_t)new_n + 1, sizeof(int));
    if (!cap || !cnt || !start)
        return -1;
    for (int i = 0; i < old_n; i++)
        cap[i] = off[i + 1] - off[i];
    for (int d = 0; d < nd; d++)
        if (dsrc[d] >= 0 && dsrc[d] < new_n && dop[d] == 1)
            cap[dsrc[d]]++;
    int total = 0;
    for (int i = 0; i < new_n; i++) {
        start[i] = total;
        total += cap[i];
    }
    int *lt = (int *)malloc((size_t)(total ? total : 1) * sizeof(int));
    double *lw = (double *)malloc((size_t)(total ? total : 1) * sizeof(double));
    if (!lt || !lw)
        return -1;
    for (int i = 0; i < old_n; i++) { /* step 1 */
        cnt[i] = off[i + 1] - off[i];
        for (int j = 0; j < cnt[i]; j++) {
            lt[start[i] + j] = tgt[off[i] + j];
            lw[start[i] + j] = (has_weights && w) ? w[off[i] + j] : 0.0;
        }
    }
    for (int d = 0; d < nd; d++) { /* step 2, log order */
        int s = dsrc[d], t = ddst[d];
        if (s < 0 || s >= new_n || t < 0 || t >= new_n)
            continue;
        int *row = lt + start[s];
        double *wr = lw + start[s];
        if (dop[d] == 2) {
            for (int j = 0; j < cnt[s]; j++)
                if (row[j] == t) {
                    cnt[s]--;
                    if (j < cnt[s]) {
                        row[j] = row[cnt[s]];
                        wr[j] = wr[cnt[s]];
                    }
                    break;
                }
        } else if (dop[d] == 1) {
            row[cnt[s]] = t;
            wr[cnt[s]] = dw ? dw[d] : 0.0;
            cnt[s]++;
        }
    }
    int o = 0; /* step 3 */
    for (int i = 0; i < new_n; i++) {
        new_off[i] = o;
        for (int j = 0; j < cnt[i]; j++) {
            new_tgt[o + j] = lt[start[i] + j];
            if (has_weights && new_w)
                new_w[o + j] = lw[start[i] + j];
        }
        o += cnt[i];
    }
    new_off[new_n] = o;
    free(cap); free(cnt); free(start); free(lt); free(lw);
    return o;
}


/* ───────────────────────── f-4: Brandes betweenness (src/graph_centrality.c:150-505) ───────────────────────── */

typedef struct {
    int node;
    double dist;
} obr_ent;

static int obr_double_eq(double a, double b) { /* :215-217 */
    return fabs(a - b) < 1e-10 * fmax(1.0, fabs(b));
}

int orc_betweenness(const orc_graph *g, int direction, int auto_approx, int normalized, double *cb, double *eb) {
    const int N = g->n;
    if (N <= 0)
        return 0;
    const int use_out = direction != 2, use_in = direction == 2 || direction == 0;
    const int weighted = g->w_out != NULL || g->w_in != NULL;
    /* predecessor slots: one per traversed edge arriving at a node */
    int *poff = (int *)calloc((size_t)N + 1, sizeof(int));
    long etrav = 0;
    if (use_out)
        for (int e = 0; e < g->off_out[N]; e++, etrav++)
            poff[g->tgt_out[e] + 1]++;
    if (use_in)
        for (int e = 0; e < g->off_in[N]; e++, etrav++)
            poff[g->tgt_in[e] + 1]++;
    for (int i = 0; i < N; i++)
        poff[i + 1] += poff[i];
    double *dist = (double *)malloc((size_t)N * sizeof(double)), *sigma = (double *)malloc((size_t)N * sizeof(double));
    double *delta = (double *)malloc((size_t)N * sizeof(double));
    int *stack = (int *)malloc((size_t)N * sizeof(int)), *queue = (int *)malloc((size_t)N * sizeof(int));
    int *pcnt = (int *)malloc((size_t)N * sizeof(int)), *pit = (int *)malloc((size_t)(poff[N] ? poff[N] : 1) * sizeof(int));
    int *settled = (int *)malloc((size_t)N * sizeof(int)), *sources = (int *)malloc((size_t)N * sizeof(int));
    obr_ent *h = (obr_ent *)malloc((size_t)(etrav + 2) * sizeof(obr_ent));
    if (!poff || !dist || !sigma || !delta || !stack || !queue || !pcnt || !pit || !settled || !sources || !h)
        return -1;
    for (int i = 0; i < N; i++)
        cb[i] = 0.0;
    int n_sources = N; /* :417-433 */
    double scale = 1.0;
    if (auto_approx > 0 && N > auto_approx) {
        n_sources = (int)ceil(sqrt((double)N));
        if (n_sources < 1)
            n_sources = 1;
        int step = N / n_sources;
        if (step < 1)
            step = 1;
        n_sources = 0;
        for (int i = 0; i < N && n_sources < (int)ceil(sqrt((double)N)); i += step)
            sources[n_sources++] = i;
        scale = (double)N / (double)n_sources;
    } else {
        for (int i = 0; i < N; i++)
            sources[i] = i;
    }
    for (int si = 0; si < n_sources; si++) {
        const int s = sources[si];
        int ss = 0;
        for (int i = 0; i < N; i++) {
            dist[i] = -1.0;
            sigma[i] = 0.0;
            pcnt[i] = 0;
            settled[i] = 0;
        }
        dist[s] = 0.0;
        sigma[s] = 1.0;
        if (!weighted) { /* sssp_bfs :263-315 */
            int qh = 0, qt = 0;
            queue[qt++] = s;
            while (qh < qt) {
                int v = queue[qh++];
                stack[ss++] = v;
                for (int pass = 0; pass < 2; pass++) {
                    if (pass == 0 ? !use_out : !use_in)
                        continue;
                    const int *off = pass ? g->off_in : g->off_out, *tgt = pass ? g->tgt_in : g->tgt_out;
                    for (int e = off[v]; e < off[v + 1]; e++) {
                        int w = tgt[e];
                        if (dist[w] < 0) {
                            dist[w] = dist[v] + 1.0;
                            queue[qt++] = w;
                        }
                        if (obr_double_eq(dist[w], dist[v] + 1.0)) {
                            if (pcnt[w] == 0 || pit[poff[w] + pcnt[w] - 1] != v) {
                                sigma[w] += sigma[v];
                                pit[poff[w] + pcnt[w]++] = v;
                            }
                        }
                    }
                }
            }
        } else { /* sssp_dijkstra :321-378, dpq :158-212 */
            int hs = 0;
            h[hs].node = s;
            h[hs].dist = 0.0;
            hs++;
            while (hs > 0) {
                obr_ent top = h[0];
                hs--;
                if (hs > 0) {
                    h[0] = h[hs];
                    int i = 0;
                    for (;;) {
                        int left = 2 * i + 1, right = 2 * i + 2, smallest = i;
                        if (left < hs && h[left].dist < h[smallest].dist)
                            smallest = left;
                        if (right < hs && h[right].dist < h[smallest].dist)
                            smallest = right;
                        if (smallest == i)
                            break;
                        obr_ent t = h[i];
                        h[i] = h[smallest];
                        h[smallest] = t;
                        i = smallest;
                    }
                }
                int v = top.node;
                if (settled[v])
                    continue;
                settled[v] = 1;
                stack[ss++] = v;
                for (int pass = 0; pass < 2; pass++) {
                    if (pass == 0 ? !use_out : !use_in)
                        continue;
                    const int *off = pass ? g->off_in : g->off_out, *tgt = pass ? g->tgt_in : g->tgt_out;
                    const double *wt = pass ? g->w_in : g->w_out;
                    for (int e = off[v]; e < off[v + 1]; e++) {
                        int w = tgt[e];
                        double nd = dist[v] + (wt ? wt[e] : 1.0);
                        if (dist[w] < 0 || nd < dist[w] - 1e-10) {
                            dist[w] = nd;
                            sigma[w] = sigma[v];
                            pit[poff[w]] = v;
                            pcnt[w] = 1;
                            int i = hs++;
                            h[i].node = w;
                            h[i].dist = nd;
                            while (i > 0) {
                                int parent = (i - 1) / 2;
                                if (h[parent].dist <= h[i].dist)
                                    break;
                                obr_ent t = h[parent];
                                h[parent] = h[i];
                                h[i] = t;
                                i = parent;
                            }
                        } else if (obr_double_eq(nd, dist[w])) {
                            if (pcnt[w] == 0 || pit[poff[w] + pcnt[w] - 1] != v) {
                                sigma[w] += sigma[v];
                                pit[poff[w] + pcnt[w]++] = v;
                            }
                        }
                    }
                }
            }
        }
        for (int i = 0; i < N; i++)
            delta[i] = 0.0;
        while (ss > 0) { /* :448-462 */
            int w = stack[--ss];
            for (int pi = 0; pi < pcnt[w]; pi++) {
                int v = pit[poff[w] + pi];
                if (sigma[w] > 0) {
                    double flow = (sigma[v] / sigma[w]) * (1.0 + delta[w]);
                    delta[v] += flow;
                    if (eb)
                        eb[(size_t)v * N + w] += flow;
                }
            }
            if (w != s)
                cb[w] += delta[w];
        }
    }
    const long NN = (long)N * N;
    if (scale != 1.0) { /* :466-474 */
        for (int i = 0; i < N; i++)
            cb[i] *= scale;
        if (eb)
            for (long i = 0; i < NN; i++)
                eb[i] *= scale;
    }
    const int undirected = direction == 0;
    if (undirected) {
        for (int i = 0; i < N; i++)
            cb[i] /= 2.0;
        if (eb)
            for (long i = 0; i < NN; i++)
                eb[i] /= 2.0;
    }
    if (normalized && N > 2) {
        double nf = undirected ? (double)(N - 1) * (double)(N - 2) / 2.0 : (double)(N - 1) * (double)(N - 2);
        for (int i = 0; i < N; i++)
            cb[i] /= nf;
        if (eb)
            for (long i = 0; i < NN; i++)
                eb[i] /= nf;
    }
    free(poff); free(dist); free(sigma); free(delta); free(stack); free(queue); free(pcnt); free(pit); free(settled);
    free(sources); free(h);
    return 0;
}
