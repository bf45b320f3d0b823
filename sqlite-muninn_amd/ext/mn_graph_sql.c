/*
 * mn_graph_sql.c — SQL surface of the graph half of the hot path over libmuninn_hip.so:
 *   node2vec_train(edge_table, src_col, dst_col, output_table, dimensions, p, q, num_walks, walk_length,
 *                  window, neg_samples, learning_rate, epochs)          (src/node2vec.c:405-597)
 *   graph_leiden   eponymous table-valued function                      (src/graph_community.c:437-670)
 * Same argument lists, defaults, error strings and output rows as the reference.  The SQL ingest
 * (string node ids → first-seen indices, adjacency in edge-table row order) is host C here as it is
 * there — with a hash map instead of the reference's linear scan (src/node2vec.c:72-77) — and the
 * compute is mn_node2vec_train / mn_graph_leiden.
 */
#include "../../include/muninn_hip.h"
#include "mn_sqlite_abi.h"

#include <stdlib.h>
#include <string.h>

/* Which device schedule the SQL functions use.  MUNINN_GRAPH_MODE=exact → the reference's sequential
 * semantics (results bit-identical to the reference, latency-bound single wavefront); =fast → the
 * batch-synchronous schedules; unset/auto → exact for small inputs (where the reference's own tests
 * live), fast beyond.   */
static int graph_mode_fast(long long work, long long small_limit) {
    const char *e = getenv("MUNINN_GRAPH_MODE");
    if (e && !strcmp(e, "exact"))
        return 0;
    if (e && !strcmp(e, "fast"))
        return 1;
    return work > small_limit;
}

#include "mn_nodemap.h" /* identifiers, string → first-seen index map */

typedef struct {
    int *v;
    double *w;
    int n, cap;
} EList;

static void el_push(EList *l, int v, double w, int with_w) {
    if (l->n >= l->cap) {
        l->cap = l->cap ? l->cap * 2 : 8;
        l->v = (int *)realloc(l->v, (size_t)l->cap * sizeof(int));
        if (with_w)
            l->w = (double *)realloc(l->w, (size_t)l->cap * sizeof(double));
    }
    l->v[l->n] = v;
    if (with_w)
        l->w[l->n] = w;
    l->n++;
}

static EList *lists_grow(EList *ls, int *cap, int need) {
    if (need <= *cap)
        return ls;
    int nc = *cap ? *cap : 256;
    while (nc < need)
        nc *= 2;
    ls = (EList *)realloc(ls, (size_t)nc * sizeof(EList));
    memset(ls + *cap, 0, (size_t)(nc - *cap) * sizeof(EList));
    *cap = nc;
    return ls;
}

static void lists_to_csr(EList *ls, int n, int with_w, int **off, int **tgt, double **w) {
    *off = (int *)malloc(((size_t)n + 1) * sizeof(int));
    (*off)[0] = 0;
    for (int i = 0; i < n; i++)
        (*off)[i + 1] = (*off)[i] + ls[i].n;
    int e = (*off)[n];
    *tgt = (int *)malloc((size_t)(e ? e : 1) * sizeof(int));
    *w = with_w ? (double *)malloc((size_t)(e ? e : 1) * sizeof(double)) : 0;
    for (int i = 0; i < n; i++) {
        memcpy(*tgt + (*off)[i], ls[i].v, (size_t)ls[i].n * sizeof(int));
        if (with_w)
            memcpy(*w + (*off)[i], ls[i].w, (size_t)ls[i].n * sizeof(double));
        free(ls[i].v);
        free(ls[i].w);
    }
    free(ls);
}

/* ───────────────────────── node2vec_train ───────────────────────── */

/* mn_vtab_hnsw.c: the output step straight into a live hnsw_index (1 done, 0 not applicable, -1 error) */
int mn_vtab_hnsw_fill_from_n2v(sqlite3 *db, const char *table, int n, const int *off, const int *adj, const mn_n2v_params *prm,
                               int *inserted, char **err);

static void fn_node2vec_train(sqlite3_context *ctx, int argc, sqlite3_value **argv) {
    if (argc < 13) {
        sqlite3_result_error(ctx, "node2vec_train: requires 13 arguments", -1);
        return;
    }
    const char *edge_table = (const char *)sqlite3_value_text(argv[0]);
    const char *src_col = (const char *)sqlite3_value_text(argv[1]);
    const char *dst_col = (const char *)sqlite3_value_text(argv[2]);
    const char *out_table = (const char *)sqlite3_value_text(argv[3]);
    mn_n2v_params prm;
    prm.dim = sqlite3_value_int(argv[4]);
    prm.p = sqlite3_value_double(argv[5]);
    prm.q = sqlite3_value_double(argv[6]);
    prm.num_walks = sqlite3_value_int(argv[7]);
    prm.walk_length = sqlite3_value_int(argv[8]);
    prm.window = sqlite3_value_int(argv[9]);
    prm.neg_samples = sqlite3_value_int(argv[10]);
    prm.learning_rate = sqlite3_value_double(argv[11]);
    prm.epochs = sqlite3_value_int(argv[12]);
    prm.batch_walks = 0;
    /* src/node2vec.c:427-464 — same checks, same messages */
    if (!ident_ok(edge_table)) { sqlite3_result_error(ctx, "node2vec_train: invalid edge_table name", -1); return; }
    if (!ident_ok(src_col)) { sqlite3_result_error(ctx, "node2vec_train: invalid src_col name", -1); return; }
    if (!ident_ok(dst_col)) { sqlite3_result_error(ctx, "node2vec_train: invalid dst_col name", -1); return; }
    if (!ident_ok(out_table)) { sqlite3_result_error(ctx, "node2vec_train: invalid output_table name", -1); return; }
    if (prm.dim <= 0 || prm.dim > 1024) { sqlite3_result_error(ctx, "node2vec_train: dimensions must be 1-1024", -1); return; }
    if (prm.p <= 0.0 || prm.q <= 0.0) { sqlite3_result_error(ctx, "node2vec_train: p and q must be > 0", -1); return; }
    if (prm.num_walks <= 0 || prm.walk_length <= 0) {
        sqlite3_result_error(ctx, "node2vec_train: num_walks and walk_length must be > 0", -1);
        return;
    }
    if (prm.window <= 0 || prm.neg_samples <= 0) {
        sqlite3_result_error(ctx, "node2vec_train: window and neg_samples must be > 0", -1);
        return;
    }
    if (prm.learning_rate <= 0.0 || prm.epochs <= 0) {
        sqlite3_result_error(ctx, "node2vec_train: learning_rate and epochs must be > 0", -1);
        return;
    }
    sqlite3 *db = sqlite3_context_db_handle(ctx);
    /* graph_load_edges (:112-138): first-seen indices, undirected, duplicates dropped, list order kept */
    char *sql = sqlite3_mprintf("SELECT \"%w\", \"%w\" FROM \"%w\"", src_col, dst_col, edge_table);
    sqlite3_stmt *st = 0;
    int rc = sqlite3_prepare_v2(db, sql, -1, &st, 0);
    sqlite3_free(sql);
    if (rc != SQLITE_OK) {
        sqlite3_result_error(ctx, "node2vec_train: failed to load edges", -1);
        return;
    }
    NodeMap nm;
    nm_init(&nm);
    EList *adj = 0;
    int adj_cap = 0;
    int load_rc = SQLITE_OK, oom = 0;
    while (!oom && (load_rc = sqlite3_step(st)) == SQLITE_ROW) {
        const char *s = (const char *)sqlite3_column_text(st, 0);
        if (!s)
            continue;
        char *scopy = sqlite3_mprintf("%s", s); /* column_text pointers do not survive the next column call */
        const char *d = (const char *)sqlite3_column_text(st, 1);
        if (!d) {
            sqlite3_free(scopy);
            continue;
        }
        int si = scopy ? nm_get(&nm, scopy) : -1;
        int di = si >= 0 ? nm_get(&nm, d) : -1;
        sqlite3_free(scopy);
        if (si < 0 || di < 0) {
            oom = 1;
            break;
        }
        adj = lists_grow(adj, &adj_cap, nm.n);
        for (int dir = 0; dir < 2; dir++) {
            int a = dir ? di : si, b = dir ? si : di;
            int dup = 0;
            for (int i = 0; i < adj[a].n; i++)
                if (adj[a].v[i] == b) {
                    dup = 1;
                    break;
                }
            if (!dup)
                el_push(&adj[a], b, 0.0, 0);
        }
    }
    sqlite3_finalize(st);
    if (oom || load_rc != SQLITE_DONE) { /* a step error is an error, not the end of the rows */
        for (int i = 0; adj && i < adj_cap; i++)
            free(adj[i].v);
        free(adj);
        nm_free(&nm);
        sqlite3_result_error(ctx, oom ? "node2vec_train: out of memory" : "node2vec_train: failed to load edges", -1);
        return;
    }
    const int n = nm.n;
    if (n == 0) {
        nm_free(&nm);
        free(adj);
        sqlite3_result_int(ctx, 0);
        return;
    }
    adj = lists_grow(adj, &adj_cap, n);
    int *off, *tgt;
    double *wdummy;
    lists_to_csr(adj, n, 0, &off, &tgt, &wdummy);
    /* pairs ≈ n · walks · length · 2·window · epochs; the serial stream handles ~0.4 M pairs/s */
    long long work = (long long)n * prm.num_walks * prm.walk_length * 2 * prm.window * prm.epochs;
    int mode = graph_mode_fast(work, 4000000LL) && n >= 512 ? MN_N2V_BATCHED : MN_N2V_SEQUENTIAL;
    if (mode == MN_N2V_BATCHED) {
        /* output table = a live, empty hnsw_index in fast mode: embeddings go from training into the index inside HBM */
        int inserted = 0;
        char *derr = 0;
        int direct = mn_vtab_hnsw_fill_from_n2v(db, out_table, n, off, tgt, &prm, &inserted, &derr);
        if (direct != 0) {
            free(off);
            free(tgt);
            nm_free(&nm);
            if (direct < 0) {
                sqlite3_result_error(ctx, derr ? derr : "node2vec_train: failed", -1);
                sqlite3_free(derr);
            } else {
                sqlite3_result_int(ctx, inserted);
            }
            return;
        }
    }
    float *emb = (float *)malloc((size_t)n * (size_t)prm.dim * sizeof(float));
    int got = emb ? mn_node2vec_train(n, off, tgt, &prm, mode, mn_env_device(), emb, 0) : -1;
    free(off);
    free(tgt);
    nm_free(&nm);
    if (got < 0) {
        free(emb);
        char *m = sqlite3_mprintf("node2vec_train: %s", emb ? mn_node2vec_last_error() : "out of memory");
        sqlite3_result_error(ctx, m, -1);
        sqlite3_free(m);
        return;
    }
    /* :554-583 — INSERT INTO out(rowid, vector) with rowid = first-seen index + 1 */
    sql = sqlite3_mprintf("INSERT INTO \"%w\" (rowid, vector) VALUES (?, ?)", out_table);
    rc = sqlite3_prepare_v2(db, sql, -1, &st, 0);
    sqlite3_free(sql);
    if (rc != SQLITE_OK) {
        free(emb);
        sqlite3_result_error(ctx, "node2vec_train: failed to prepare INSERT for output table (does it exist?)", -1);
        return;
    }
    int inserted = 0;
    for (int i = 0; i < n; i++) {
        sqlite3_bind_int64(st, 1, (sqlite3_int64)(i + 1));
        sqlite3_bind_blob(st, 2, emb + (size_t)i * prm.dim, prm.dim * (int)sizeof(float), SQLITE_TRANSIENT);
        if (sqlite3_step(st) == SQLITE_DONE)
            inserted++;
        sqlite3_reset(st);
    }
    sqlite3_finalize(st);
    free(emb);
    sqlite3_result_int(ctx, inserted);
}

/* ───────────────────────── graph_leiden ───────────────────────── */

typedef struct {
    sqlite3_vtab base;
    sqlite3 *db;
} LeiVtab;

typedef struct {
    sqlite3_vtab_cursor base;
    char **node;
    int *community;
    double Q;
    int n, pos, eof;
} LeiCursor;

enum { LC_NODE = 0, LC_COMM, LC_MOD, LC_EDGE_TABLE, LC_SRC, LC_DST, LC_WEIGHT, LC_RES, LC_DIR, LC_TS, LC_T0, LC_T1 };

static int lei_connect(sqlite3 *db, void *aux, int argc, const char *const *argv, sqlite3_vtab **out, char **err) {
    (void)aux; (void)argc; (void)argv; (void)err;
    int rc = sqlite3_declare_vtab(db, "CREATE TABLE x(  node TEXT, community_id INTEGER, modularity REAL,"
                                      "  edge_table TEXT HIDDEN, src_col TEXT HIDDEN, dst_col TEXT HIDDEN,"
                                      "  weight_col TEXT HIDDEN, resolution REAL HIDDEN,"
                                      "  direction TEXT HIDDEN, timestamp_col TEXT HIDDEN,"
                                      "  time_start HIDDEN, time_end HIDDEN)");
    if (rc != SQLITE_OK)
        return rc;
    LeiVtab *v = (LeiVtab *)sqlite3_malloc((int)sizeof(LeiVtab));
    if (!v)
        return SQLITE_NOMEM;
    memset(v, 0, sizeof(*v));
    v->db = db;
    *out = &v->base;
    return SQLITE_OK;
}

static int lei_disconnect(sqlite3_vtab *v) {
    sqlite3_free(v);
    return SQLITE_OK;
}

/* graph_best_index_common (src/graph_common.h:62-96): hidden columns → bitmask idxNum, argv in column order */
static int lei_best_index(sqlite3_vtab *v, sqlite3_index_info *ii) {
    (void)v;
    int which[9];
    for (int j = 0; j < 9; j++)
        which[j] = -1;
    for (int i = 0; i < ii->nConstraint; i++) {
        if (!ii->aConstraint[i].usable || ii->aConstraint[i].op != SQLITE_INDEX_CONSTRAINT_EQ)
            continue;
        int col = ii->aConstraint[i].iColumn;
        if (col >= LC_EDGE_TABLE && col <= LC_T1)
            which[col - LC_EDGE_TABLE] = i;
    }
    int arg = 1, mask = 0;
    for (int j = 0; j < 9; j++)
        if (which[j] >= 0) {
            ii->aConstraintUsage[which[j]].argvIndex = arg++;
            ii->aConstraintUsage[which[j]].omit = 1;
            mask |= 1 << j;
        }
    ii->idxNum = mask;
    ii->estimatedCost = (mask & 0x7) == 0x7 ? 5000.0 : 1e12;
    return SQLITE_OK;
}

static int lei_open(sqlite3_vtab *v, sqlite3_vtab_cursor **out) {
    (void)v;
    LeiCursor *c = (LeiCursor *)calloc(1, sizeof(LeiCursor));
    if (!c)
        return SQLITE_NOMEM;
    c->eof = 1;
    *out = &c->base;
    return SQLITE_OK;
}

static void lei_clear(LeiCursor *c) {
    for (int i = 0; i < c->n; i++)
        free(c->node[i]);
    free(c->node);
    free(c->community);
    c->node = 0;
    c->community = 0;
    c->n = 0;
}

static int lei_close(sqlite3_vtab_cursor *cur) {
    lei_clear((LeiCursor *)cur);
    free(cur);
    return SQLITE_OK;
}

static const char *safe_text(sqlite3_value *v) {
    return v && sqlite3_value_type(v) != SQLITE_NULL ? (const char *)sqlite3_value_text(v) : 0;
}

/* ── a graph_adjacency table as edge_table (src/graph_community.c:577-581) ──
 * The reference recognises one by the 'edge_table' key in "{t}_config" (is_graph_adjacency,
 * src/graph_adjacency.c:1414-1424).  If its delta log is empty the stored CSR is current and is used as is
 * (:1571-1572): node ids from "{t}_nodes" in idx order and the "{t}_csr_fwd" / "{t}_csr_rev" block rows go
 * straight to the device (mn_graph_create_blocked); otherwise the original edge table named in the config is
 * read with direction 'both' (:1536-1569).  Only shadow tables are read, so the graph_adjacency module itself
 * does not have to be registered. */
static char *adj_cfg(sqlite3 *db, const char *t, const char *key) {
    char *sql = sqlite3_mprintf("SELECT value FROM \"%w_config\" WHERE key=%Q", t, key);
    sqlite3_stmt *st = 0;
    char *out = 0;
    if (sqlite3_prepare_v2(db, sql, -1, &st, 0) == SQLITE_OK && sqlite3_step(st) == SQLITE_ROW && sqlite3_column_text(st, 0))
        out = sqlite3_mprintf("%s", (const char *)sqlite3_column_text(st, 0));
    sqlite3_finalize(st);
    sqlite3_free(sql);
    return out;
}

static long long adj_delta_count(sqlite3 *db, const char *t) {
    char *sql = sqlite3_mprintf("SELECT COUNT(*) FROM \"%w_delta\"", t);
    sqlite3_stmt *st = 0;
    long long n = 0;
    if (sqlite3_prepare_v2(db, sql, -1, &st, 0) == SQLITE_OK && sqlite3_step(st) == SQLITE_ROW)
        n = sqlite3_column_int64(st, 0);
    sqlite3_finalize(st);
    sqlite3_free(sql);
    return n;
}

/* all rows of one CSR shadow table, block_id order; blobs are copied because they must outlive the statement */
static int adj_read_blocks(sqlite3 *db, const char *t, const char *suffix, mn_csr_block **out, int *n_out) {
    char *sql = sqlite3_mprintf("SELECT offsets, targets, weights FROM \"%w%s\" ORDER BY block_id", t, suffix);
    sqlite3_stmt *st = 0;
    int rc = sqlite3_prepare_v2(db, sql, -1, &st, 0);
    sqlite3_free(sql);
    *out = 0;
    *n_out = 0;
    if (rc != SQLITE_OK)
        return rc;
    int cap = 0;
    while (sqlite3_step(st) == SQLITE_ROW) {
        if (*n_out == cap) {
            cap = cap ? cap * 2 : 16;
            *out = (mn_csr_block *)realloc(*out, (size_t)cap * sizeof(mn_csr_block));
        }
        mn_csr_block *b = &(*out)[(*n_out)++];
        memset(b, 0, sizeof(*b));
        for (int col = 0; col < 3; col++) {
            int bytes = sqlite3_column_bytes(st, col);
            const void *src = sqlite3_column_blob(st, col);
            void *copy = 0;
            if (src && bytes > 0) {
                copy = malloc((size_t)bytes);
                memcpy(copy, src, (size_t)bytes);
            } else {
                bytes = 0;
            }
            if (col == 0) { b->offsets = copy; b->offsets_bytes = bytes; }
            else if (col == 1) { b->targets = copy; b->targets_bytes = bytes; }
            else { b->weights = copy; b->weights_bytes = bytes; }
        }
    }
    sqlite3_finalize(st);
    return SQLITE_OK;
}

static void adj_free_blocks(mn_csr_block *b, int n) {
    for (int i = 0; i < n; i++) {
        free((void *)b[i].offsets);
        free((void *)b[i].targets);
        free((void *)b[i].weights);
    }
    free(b);
}

/* the stored CSR of a fresh graph_adjacency table → device graph + node ids; *ids_out owns malloc'd strings */
static mn_graph *adj_load_fresh(sqlite3 *db, const char *t, char ***ids_out, int *n_out, char **err) {
    char *sql = sqlite3_mprintf("SELECT id FROM \"%w_nodes\" ORDER BY idx", t);
    sqlite3_stmt *st = 0;
    int rc = sqlite3_prepare_v2(db, sql, -1, &st, 0);
    sqlite3_free(sql);
    if (rc != SQLITE_OK) {
        *err = sqlite3_mprintf("failed to load nodes");
        return 0;
    }
    char **ids = 0;
    int n = 0, cap = 0;
    while (sqlite3_step(st) == SQLITE_ROW) {
        const char *id = (const char *)sqlite3_column_text(st, 0);
        if (!id)
            continue;
        if (n == cap) {
            cap = cap ? cap * 2 : 256;
            ids = (char **)realloc(ids, (size_t)cap * sizeof(char *));
        }
        size_t len = strlen(id) + 1;
        ids[n] = (char *)malloc(len);
        memcpy(ids[n++], id, len);
    }
    sqlite3_finalize(st);
    *ids_out = ids;
    *n_out = n;
    if (n == 0)
        return 0; /* empty graph: no error */
    mn_csr_block *fwd = 0, *rev = 0;
    int nf = 0, nr = 0;
    mn_graph *g = 0;
    if (adj_read_blocks(db, t, "_csr_fwd", &fwd, &nf) != SQLITE_OK)
        *err = sqlite3_mprintf("failed to load forward CSR");
    else if (adj_read_blocks(db, t, "_csr_rev", &rev, &nr) != SQLITE_OK)
        *err = sqlite3_mprintf("failed to load reverse CSR");
    else if (!(g = mn_graph_create_blocked(n, fwd, nf, rev, nr, mn_env_device())))
        *err = sqlite3_mprintf("graph_leiden: %s", mn_graph_last_error());
    adj_free_blocks(fwd, nf);
    adj_free_blocks(rev, nr);
    return g;
}

/* run_leiden on a device graph and hand the result to the cursor (takes ownership of ids) */
static int lei_run(LeiCursor *c, LeiVtab *vt, mn_graph *g, int n, char **ids, const char *direction, double resolution) {
    c->community = (int *)malloc((size_t)n * sizeof(int));
    int rc = mn_graph_leiden(g, resolution, !strcmp(direction, "both"),
                             graph_mode_fast(n, 2000) ? MN_LEIDEN_BATCHED : MN_LEIDEN_SEQUENTIAL, 0, c->community, &c->Q);
    mn_graph_destroy(g);
    if (rc != 0) {
        vt->base.zErrMsg = sqlite3_mprintf("graph_leiden: %s", mn_graph_last_error());
        for (int i = 0; i < n; i++)
            free(ids[i]);
        free(ids);
        free(c->community);
        c->community = 0;
        return SQLITE_ERROR;
    }
    c->node = ids;
    c->n = n;
    c->eof = 0;
    return SQLITE_OK;
}

/* graph_data_load (src/graph_load.c:144-250) — or, when edge_table names a graph_adjacency table, graph_data_load_from_adjacency
 * (src/graph_adjacency.c:1532-1573) — straight to a device graph: text ids → first-seen indices, out / in lists in row order
 * for the directions loaded, weights when a weight column is given.  SQLITE_OK with *g_out = NULL for an empty graph; on error
 * *err is set (sqlite3_mprintf).  The ids (malloc'ed strings) and the graph belong to the caller.  Shared by graph_leiden and
 * the centrality TVFs (mn_graph_tvf.c). */
int mn_sql_load_graph(sqlite3 *db, const char *who, const char *edge_table, const char *src_col, const char *dst_col,
                      const char *weight_col, const char *direction, const char *ts_col, sqlite3_value *t0, sqlite3_value *t1,
                      mn_graph **g_out, char ***ids_out, int *n_out, char **err) {
    *g_out = 0;
    *ids_out = 0;
    *n_out = 0;
    if (!ident_ok(edge_table) || !ident_ok(src_col) || !ident_ok(dst_col)) {
        *err = sqlite3_mprintf("invalid table/column identifier");
        return SQLITE_ERROR;
    }
    if (weight_col && !ident_ok(weight_col)) {
        *err = sqlite3_mprintf("invalid weight column identifier");
        return SQLITE_ERROR;
    }
    if (ts_col && !ident_ok(ts_col)) {
        *err = sqlite3_mprintf("invalid timestamp column identifier");
        return SQLITE_ERROR;
    }
    const char *load_dir = direction;
    char *a_table = adj_cfg(db, edge_table, "edge_table"), *a_src = 0, *a_dst = 0, *a_w = 0;
    if (a_table) { /* edge_table names a graph_adjacency table */
        if (adj_delta_count(db, edge_table) == 0) {
            sqlite3_free(a_table);
            char **ids = 0, *e2 = 0;
            int n = 0;
            mn_graph *g = adj_load_fresh(db, edge_table, &ids, &n, &e2);
            if (!g) {
                for (int i = 0; i < n; i++)
                    free(ids[i]);
                free(ids);
                if (e2) {
                    *err = e2;
                    return SQLITE_ERROR;
                }
                return SQLITE_OK; /* empty graph */
            }
            *g_out = g;
            *ids_out = ids;
            *n_out = n;
            return SQLITE_OK;
        }
        /* stale: the original edge table, both directions, no time window (src/graph_adjacency.c:1536-1569) */
        a_src = adj_cfg(db, edge_table, "src_col");
        a_dst = adj_cfg(db, edge_table, "dst_col");
        a_w = adj_cfg(db, edge_table, "weight_col");
        if (!a_src || !a_dst) {
            *err = sqlite3_mprintf("graph_adjacency '%s': missing config", edge_table);
            sqlite3_free(a_table); sqlite3_free(a_src); sqlite3_free(a_dst); sqlite3_free(a_w);
            return SQLITE_ERROR;
        }
        edge_table = a_table;
        src_col = a_src;
        dst_col = a_dst;
        weight_col = a_w;
        ts_col = 0;
        load_dir = "both";
    }
    char *sql;
    if (weight_col && ts_col)
        sql = sqlite3_mprintf("SELECT \"%w\", \"%w\", \"%w\" FROM \"%w\" WHERE (\"%w\" >= ?1 OR ?1 IS NULL) AND (\"%w\" <= ?2 OR ?2 IS NULL)",
                              src_col, dst_col, weight_col, edge_table, ts_col, ts_col);
    else if (weight_col)
        sql = sqlite3_mprintf("SELECT \"%w\", \"%w\", \"%w\" FROM \"%w\"", src_col, dst_col, weight_col, edge_table);
    else if (ts_col)
        sql = sqlite3_mprintf("SELECT \"%w\", \"%w\" FROM \"%w\" WHERE (\"%w\" >= ?1 OR ?1 IS NULL) AND (\"%w\" <= ?2 OR ?2 IS NULL)",
                              src_col, dst_col, edge_table, ts_col, ts_col);
    else
        sql = sqlite3_mprintf("SELECT \"%w\", \"%w\" FROM \"%w\"", src_col, dst_col, edge_table);
    sqlite3_stmt *st = 0;
    int rc = sqlite3_prepare_v2(db, sql, -1, &st, 0);
    sqlite3_free(sql);
    const int with_w = weight_col != 0;
    sqlite3_free(a_table); sqlite3_free(a_src); sqlite3_free(a_dst); sqlite3_free(a_w); /* names are in the statement now */
    if (rc != SQLITE_OK) {
        *err = sqlite3_mprintf("failed to prepare: %s", sqlite3_errmsg(db));
        return SQLITE_ERROR;
    }
    if (ts_col) {
        if (t0) sqlite3_bind_value(st, 1, t0); else sqlite3_bind_null(st, 1);
        if (t1) sqlite3_bind_value(st, 2, t1); else sqlite3_bind_null(st, 2);
    }
    int add_fwd = 1, add_rev = 1;
    if (!strcmp(load_dir, "forward")) add_rev = 0;
    else if (!strcmp(load_dir, "reverse")) add_fwd = 0;
    NodeMap nm;
    nm_init(&nm);
    EList *outl = 0, *inl = 0;
    int ocap = 0, icap = 0;
    int load_rc = SQLITE_OK, oom = 0;
    while (!oom && (load_rc = sqlite3_step(st)) == SQLITE_ROW) {
        const char *s = (const char *)sqlite3_column_text(st, 0);
        if (!s)
            continue;
        char *scopy = sqlite3_mprintf("%s", s);
        const char *d = (const char *)sqlite3_column_text(st, 1);
        if (!d) {
            sqlite3_free(scopy);
            continue;
        }
        double w = with_w ? sqlite3_column_double(st, 2) : 1.0;
        int si = scopy ? nm_get(&nm, scopy) : -1;
        int di = si >= 0 ? nm_get(&nm, d) : -1;
        sqlite3_free(scopy);
        if (si < 0 || di < 0) {
            oom = 1;
            break;
        }
        outl = lists_grow(outl, &ocap, nm.n);
        inl = lists_grow(inl, &icap, nm.n);
        if (add_fwd)
            el_push(&outl[si], di, w, with_w);
        if (add_rev)
            el_push(&inl[di], si, w, with_w);
    }
    sqlite3_finalize(st);
    if (oom || load_rc != SQLITE_DONE) { /* a step error is an error, not the end of the rows */
        for (int i = 0; outl && i < ocap; i++)
            free(outl[i].v), free(outl[i].w);
        for (int i = 0; inl && i < icap; i++)
            free(inl[i].v), free(inl[i].w);
        free(outl);
        free(inl);
        nm_free(&nm);
        *err = sqlite3_mprintf("%s: %s", who, oom ? "out of memory" : "failed to read the edge table");
        return oom ? SQLITE_NOMEM : SQLITE_ERROR;
    }
    const int n = nm.n;
    if (n == 0) {
        nm_free(&nm);
        free(outl);
        free(inl);
        return SQLITE_OK;
    }
    outl = lists_grow(outl, &ocap, n);
    inl = lists_grow(inl, &icap, n);
    int *oo, *ot, *io, *it;
    double *ow, *iw;
    lists_to_csr(outl, n, with_w, &oo, &ot, &ow);
    lists_to_csr(inl, n, with_w, &io, &it, &iw);
    mn_graph *g = mn_graph_create(n, oo, ot, ow, io, it, iw, mn_env_device());
    free(oo); free(ot); free(ow); free(io); free(it); free(iw);
    if (!g) {
        *err = sqlite3_mprintf("%s: %s", who, mn_graph_last_error());
        nm_free(&nm);
        return SQLITE_ERROR;
    }
    free(nm.slots);
    *g_out = g;
    *ids_out = nm.ids; /* ownership of the ids moves to the caller */
    *n_out = n;
    return SQLITE_OK;
}

/* src/graph_community.c:516-610 + graph_data_load (src/graph_load.c:144-250) */
static int lei_filter(sqlite3_vtab_cursor *cur, int idxNum, const char *idxStr, int argc, sqlite3_value **argv) {
    (void)idxStr;
    LeiCursor *c = (LeiCursor *)cur;
    LeiVtab *vt = (LeiVtab *)cur->pVtab;
    lei_clear(c);
    c->pos = 0;
    c->eof = 1;
    if (argc < 3)
        return SQLITE_OK;
    const char *edge_table = 0, *src_col = 0, *dst_col = 0, *weight_col = 0, *direction = 0, *ts_col = 0;
    sqlite3_value *t0 = 0, *t1 = 0;
    double resolution = 1.0;
    int pos = 0;
    for (int bit = 0; bit < 9 && pos < argc; bit++) {
        if (!(idxNum & (1 << bit)))
            continue;
        switch (bit + LC_EDGE_TABLE) {
        case LC_EDGE_TABLE: edge_table = safe_text(argv[pos]); break;
        case LC_SRC: src_col = safe_text(argv[pos]); break;
        case LC_DST: dst_col = safe_text(argv[pos]); break;
        case LC_WEIGHT: weight_col = safe_text(argv[pos]); break;
        case LC_RES: resolution = sqlite3_value_double(argv[pos]); break;
        case LC_DIR: direction = safe_text(argv[pos]); break;
        case LC_TS: ts_col = safe_text(argv[pos]); break;
        case LC_T0: t0 = argv[pos]; break;
        case LC_T1: t1 = argv[pos]; break;
        }
        pos++;
    }
    if (!direction)
        direction = "both";
    mn_graph *g = 0;
    char **ids = 0, *err = 0;
    int n = 0;
    if (mn_sql_load_graph(vt->db, "graph_leiden", edge_table, src_col, dst_col, weight_col, direction, ts_col, t0, t1, &g, &ids, &n,
                          &err) != SQLITE_OK) {
        vt->base.zErrMsg = err;
        return SQLITE_ERROR;
    }
    if (!g)
        return SQLITE_OK; /* empty graph */
    return lei_run(c, vt, g, n, ids, direction, resolution); /* ownership of the ids moves to the cursor */
}

static int lei_next(sqlite3_vtab_cursor *cur) {
    LeiCursor *c = (LeiCursor *)cur;
    c->pos++;
    c->eof = c->pos >= c->n;
    return SQLITE_OK;
}
static int lei_eof(sqlite3_vtab_cursor *cur) { return ((LeiCursor *)cur)->eof; }
static int lei_column(sqlite3_vtab_cursor *cur, sqlite3_context *ctx, int col) {
    LeiCursor *c = (LeiCursor *)cur;
    switch (col) {
    case LC_NODE: sqlite3_result_text(ctx, c->node[c->pos], -1, SQLITE_TRANSIENT); break;
    case LC_COMM: sqlite3_result_int(ctx, c->community[c->pos]); break;
    case LC_MOD: sqlite3_result_double(ctx, c->Q); break;
    default: sqlite3_result_null(ctx); break;
    }
    return SQLITE_OK;
}
static int lei_rowid(sqlite3_vtab_cursor *cur, sqlite3_int64 *out) {
    *out = ((LeiCursor *)cur)->pos;
    return SQLITE_OK;
}

static sqlite3_module leiden_module = {
    .iVersion = 0,
    .xCreate = 0, /* eponymous */
    .xConnect = lei_connect,
    .xBestIndex = lei_best_index,
    .xDisconnect = lei_disconnect,
    .xDestroy = lei_disconnect,
    .xOpen = lei_open,
    .xClose = lei_close,
    .xFilter = lei_filter,
    .xNext = lei_next,
    .xEof = lei_eof,
    .xColumn = lei_column,
    .xRowid = lei_rowid,
};

/* one registration per reference registration function, so that an integrated build can keep the reference's
 * sqlite3_muninn_init (src/muninn.c:42-121) and point its calls here */
int mn_register_node2vec(sqlite3 *db) { /* node2vec_register_functions, src/node2vec.c:594-597 */
    return sqlite3_create_function(db, "node2vec_train", 13, SQLITE_UTF8 | SQLITE_DETERMINISTIC, 0, fn_node2vec_train, 0, 0);
}
int mn_register_leiden(sqlite3 *db) { /* community_register_tvfs, src/graph_community.c:668-670 */
    return sqlite3_create_module(db, "graph_leiden", &leiden_module, 0);
}
int mn_register_graph_functions(sqlite3 *db) {
    int rc = mn_register_node2vec(db);
    if (rc == SQLITE_OK)
        rc = mn_register_leiden(db);
    return rc;
}
