/*
 * mn_graph_tvf.c — graph_pagerank and graph_components over libmuninn_hip.so.
 * Same eponymous table-valued functions as the reference (src/graph_tvf.c:1367-1529 and :1719-1889): declared schema,
 * hidden-column constraints → argv mapping, cost rule, defaults (damping 0.85, 20 iterations), identifier check and
 * error strings, rows in first-seen node order.  The SQL ingest (text ids → first-seen indices, edges in row order,
 * NULL rows skipped) is host C; the compute is mn_graph_pagerank / mn_graph_components.
 */
#include "../../include/muninn_hip.h"
#include "mn_sqlite_abi.h"

#include "mn_nodemap.h"

typedef struct {
    sqlite3_vtab base;
    sqlite3 *db;
} TvfVtab;

typedef struct {
    sqlite3_vtab_cursor base;
    NodeMap nodes;
    int have_nodes;
    double *rank;
    int *comp_id, *comp_size;
    int n, pos, eof;
} TvfCursor;

static void cur_clear(TvfCursor *c) {
    if (c->have_nodes)
        nm_free(&c->nodes);
    c->have_nodes = 0;
    free(c->rank);
    free(c->comp_id);
    free(c->comp_size);
    c->rank = 0;
    c->comp_id = c->comp_size = 0;
    c->n = 0;
}

static int tvf_disconnect(sqlite3_vtab *v) {
    sqlite3_free(v);
    return SQLITE_OK;
}

static int tvf_open(sqlite3_vtab *v, sqlite3_vtab_cursor **out) {
    (void)v;
    TvfCursor *c = (TvfCursor *)calloc(1, sizeof(TvfCursor));
    if (!c)
        return SQLITE_NOMEM;
    c->eof = 1;
    *out = &c->base;
    return SQLITE_OK;
}

static int tvf_close(sqlite3_vtab_cursor *cur) {
    cur_clear((TvfCursor *)cur);
    free(cur);
    return SQLITE_OK;
}

static int tvf_next(sqlite3_vtab_cursor *cur) {
    TvfCursor *c = (TvfCursor *)cur;
    c->pos++;
    c->eof = c->pos >= c->n;
    return SQLITE_OK;
}
static int tvf_eof(sqlite3_vtab_cursor *cur) { return ((TvfCursor *)cur)->eof; }
static int tvf_rowid(sqlite3_vtab_cursor *cur, sqlite3_int64 *out) {
    *out = ((TvfCursor *)cur)->pos;
    return SQLITE_OK;
}

static int tvf_connect_with(sqlite3 *db, const char *schema, sqlite3_vtab **out) {
    int rc = sqlite3_declare_vtab(db, schema);
    if (rc != SQLITE_OK)
        return rc;
    TvfVtab *v = (TvfVtab *)sqlite3_malloc((int)sizeof(TvfVtab));
    if (!v)
        return SQLITE_NOMEM;
    memset(v, 0, sizeof(*v));
    v->db = db;
    *out = &v->base;
    return SQLITE_OK;
}

/* the reference's xBestIndex of these two TVFs (src/graph_tvf.c:1394-1418,1745-1769): argvIndex = hidden-column position
 * + 1 — NOT compacted, so naming a later hidden column without the ones before it makes SQLite report
 * "xBestIndex malfunction", there as here */
static int tvf_best_index(sqlite3_index_info *ii, int first_hidden, int last_hidden, int need_all) {
    int mask = 0;
    for (int i = 0; i < ii->nConstraint; i++) {
        if (!ii->aConstraint[i].usable || ii->aConstraint[i].op != SQLITE_INDEX_CONSTRAINT_EQ)
            continue;
        int col = ii->aConstraint[i].iColumn;
        if (col >= first_hidden && col <= last_hidden) {
            ii->aConstraintUsage[i].argvIndex = col - first_hidden + 1;
            ii->aConstraintUsage[i].omit = 1;
            mask |= 1 << (col - first_hidden);
        }
    }
    ii->estimatedCost = (need_all ? mask == 0x7 : (mask & 0x7) == 0x7) ? 1000.0 : 1e12;
    return SQLITE_OK;
}

/* "SELECT src, dst FROM edge_table": text ids → first-seen indices (src of a row before its dst), rows with a NULL skipped */
static int read_edges(sqlite3 *db, const char *t, const char *sc, const char *dc, NodeMap *nm, int **src, int **dst, long long *ne) {
    char *sql = sqlite3_mprintf("SELECT \"%w\", \"%w\" FROM \"%w\"", sc, dc, t);
    if (!sql)
        return SQLITE_NOMEM;
    sqlite3_stmt *st = 0;
    int rc = sqlite3_prepare_v2(db, sql, -1, &st, 0);
    sqlite3_free(sql);
    if (rc != SQLITE_OK)
        return rc;
    long long n = 0, cap = 1024;
    int *s = (int *)malloc((size_t)cap * sizeof(int)), *d = (int *)malloc((size_t)cap * sizeof(int));
    int oom = !s || !d;
    while (!oom && (rc = sqlite3_step(st)) == SQLITE_ROW) {
        const char *a = (const char *)sqlite3_column_text(st, 0);
        const char *b = (const char *)sqlite3_column_text(st, 1);
        if (!a || !b)
            continue;
        if (n >= cap) {
            int *s2 = (int *)realloc(s, (size_t)cap * 2 * sizeof(int));
            if (s2)
                s = s2;
            int *d2 = s2 ? (int *)realloc(d, (size_t)cap * 2 * sizeof(int)) : 0;
            if (d2)
                d = d2;
            if (!s2 || !d2) {
                oom = 1;
                break;
            }
            cap *= 2;
        }
        int si = nm_get(nm, a); /* (a's text pointer is only valid until the next column call: nm_get copies) */
        b = (const char *)sqlite3_column_text(st, 1);
        int di = b ? nm_get(nm, b) : -1;
        if (si < 0 || di < 0) {
            oom = 1;
            break;
        }
        s[n] = si;
        d[n] = di;
        n++;
    }
    sqlite3_finalize(st);
    if (oom || rc != SQLITE_DONE) { /* a step error is an error, not the end of the rows */
        free(s);
        free(d);
        return oom ? SQLITE_NOMEM : rc;
    }
    *src = s;
    *dst = d;
    *ne = n;
    return SQLITE_OK;
}

static int three_idents(TvfCursor *c, sqlite3_vtab_cursor *cur, int argc, sqlite3_value **argv, const char *who, const char **t,
                        const char **sc, const char **dc) {
    cur_clear(c);
    c->eof = 1;
    c->pos = 0;
    if (argc < 3)
        return 1; /* no rows (:1446-1449) */
    *t = (const char *)sqlite3_value_text(argv[0]);
    *sc = (const char *)sqlite3_value_text(argv[1]);
    *dc = (const char *)sqlite3_value_text(argv[2]);
    if (!ident_ok(*t) || !ident_ok(*sc) || !ident_ok(*dc)) {
        cur->pVtab->zErrMsg = sqlite3_mprintf("%s: invalid table/column identifier", who);
        return -1;
    }
    return 0;
}

/* ───────────────────────── graph_components ───────────────────────── */

enum { GC_NODE = 0, GC_ID, GC_SIZE, GC_EDGE_TABLE, GC_SRC, GC_DST };

static int gc_connect(sqlite3 *db, void *aux, int argc, const char *const *argv, sqlite3_vtab **out, char **err) {
    (void)aux; (void)argc; (void)argv; (void)err;
    return tvf_connect_with(db, "CREATE TABLE x(node TEXT, component_id INTEGER, component_size INTEGER,"
                                " edge_table TEXT HIDDEN, src_col TEXT HIDDEN, dst_col TEXT HIDDEN)", out);
}
static int gc_best_index(sqlite3_vtab *v, sqlite3_index_info *ii) {
    (void)v;
    return tvf_best_index(ii, GC_EDGE_TABLE, GC_DST, 1);
}

/* which union sequence: the reference's (component_id = its union-find root) for inputs of its own test sizes, parallel
 * hooking beyond (same partition and sizes, id = smallest node index); MUNINN_GRAPH_MODE=exact|fast forces one */
static int components_mode(long long n_edges) {
    const char *e = getenv("MUNINN_GRAPH_MODE");
    if (e && !strcmp(e, "exact"))
        return MN_COMPONENTS_EXACT;
    if (e && !strcmp(e, "fast"))
        return MN_COMPONENTS_FAST;
    return n_edges > 200000 ? MN_COMPONENTS_FAST : MN_COMPONENTS_EXACT;
}

static int gc_filter(sqlite3_vtab_cursor *cur, int idxNum, const char *idxStr, int argc, sqlite3_value **argv) {
    (void)idxNum; (void)idxStr;
    TvfCursor *c = (TvfCursor *)cur;
    const char *t, *sc, *dc;
    int r = three_idents(c, cur, argc, argv, "graph_components", &t, &sc, &dc);
    if (r)
        return r < 0 ? SQLITE_ERROR : SQLITE_OK;
    nm_init(&c->nodes);
    c->have_nodes = 1;
    int *src = 0, *dst = 0;
    long long ne = 0;
    int rc = read_edges(((TvfVtab *)cur->pVtab)->db, t, sc, dc, &c->nodes, &src, &dst, &ne);
    if (rc != SQLITE_OK) {
        free(src);
        free(dst);
        return rc;
    }
    const int n = c->nodes.n;
    if (n > 0) {
        c->comp_id = (int *)malloc((size_t)n * sizeof(int));
        c->comp_size = (int *)malloc((size_t)n * sizeof(int));
        if (mn_graph_components(n, ne, src, dst, components_mode(ne), mn_env_device(), c->comp_id, c->comp_size, 0) != 0) {
            cur->pVtab->zErrMsg = sqlite3_mprintf("graph_components: %s", mn_graph_algo_last_error());
            free(src);
            free(dst);
            return SQLITE_ERROR;
        }
    }
    free(src);
    free(dst);
    c->n = n;
    c->eof = n == 0;
    return SQLITE_OK;
}

static int gc_column(sqlite3_vtab_cursor *cur, sqlite3_context *ctx, int col) {
    TvfCursor *c = (TvfCursor *)cur;
    switch (col) {
    case GC_NODE: sqlite3_result_text(ctx, c->nodes.ids[c->pos], -1, SQLITE_TRANSIENT); break;
    case GC_ID: sqlite3_result_int(ctx, c->comp_id[c->pos]); break;
    case GC_SIZE: sqlite3_result_int(ctx, c->comp_size[c->pos]); break;
    default: sqlite3_result_null(ctx); break;
    }
    return SQLITE_OK;
}

static sqlite3_module components_module = {
    .iVersion = 0, .xCreate = 0, .xConnect = gc_connect, .xBestIndex = gc_best_index, .xDisconnect = tvf_disconnect,
    .xDestroy = tvf_disconnect, .xOpen = tvf_open, .xClose = tvf_close, .xFilter = gc_filter, .xNext = tvf_next,
    .xEof = tvf_eof, .xColumn = gc_column, .xRowid = tvf_rowid,
};

/* ───────────────────────── graph_pagerank ───────────────────────── */

enum { GPR_NODE = 0, GPR_RANK, GPR_EDGE_TABLE, GPR_SRC, GPR_DST, GPR_DAMPING, GPR_ITER };

static int gpr_connect(sqlite3 *db, void *aux, int argc, const char *const *argv, sqlite3_vtab **out, char **err) {
    (void)aux; (void)argc; (void)argv; (void)err;
    return tvf_connect_with(db, "CREATE TABLE x(node TEXT, rank REAL,"
                                " edge_table TEXT HIDDEN, src_col TEXT HIDDEN, dst_col TEXT HIDDEN,"
                                " damping REAL HIDDEN, iterations INTEGER HIDDEN)", out);
}
static int gpr_best_index(sqlite3_vtab *v, sqlite3_index_info *ii) {
    (void)v;
    return tvf_best_index(ii, GPR_EDGE_TABLE, GPR_ITER, 0);
}

static int gpr_filter(sqlite3_vtab_cursor *cur, int idxNum, const char *idxStr, int argc, sqlite3_value **argv) {
    (void)idxNum; (void)idxStr;
    TvfCursor *c = (TvfCursor *)cur;
    const char *t, *sc, *dc;
    int r = three_idents(c, cur, argc, argv, "graph_pagerank", &t, &sc, &dc);
    if (r)
        return r < 0 ? SQLITE_ERROR : SQLITE_OK;
    double damping = 0.85; /* :1820-1828 */
    int iterations = 20;
    if (argc > 3 && sqlite3_value_type(argv[3]) != SQLITE_NULL)
        damping = sqlite3_value_double(argv[3]);
    if (argc > 4 && sqlite3_value_type(argv[4]) != SQLITE_NULL)
        iterations = sqlite3_value_int(argv[4]);
    nm_init(&c->nodes);
    c->have_nodes = 1;
    int *src = 0, *dst = 0;
    long long ne = 0;
    int rc = read_edges(((TvfVtab *)cur->pVtab)->db, t, sc, dc, &c->nodes, &src, &dst, &ne);
    if (rc != SQLITE_OK) {
        free(src);
        free(dst);
        return rc;
    }
    const int n = c->nodes.n;
    if (n > 0) {
        c->rank = (double *)malloc((size_t)n * sizeof(double));
        if (mn_graph_pagerank(n, ne, src, dst, damping, iterations, mn_env_device(), c->rank, 0) != 0) {
            cur->pVtab->zErrMsg = sqlite3_mprintf("graph_pagerank: %s", mn_graph_algo_last_error());
            free(src);
            free(dst);
            return SQLITE_ERROR;
        }
    }
    free(src);
    free(dst);
    c->n = n;
    c->eof = n == 0;
    return SQLITE_OK;
}

static int gpr_column(sqlite3_vtab_cursor *cur, sqlite3_context *ctx, int col) {
    TvfCursor *c = (TvfCursor *)cur;
    switch (col) {
    case GPR_NODE: sqlite3_result_text(ctx, c->nodes.ids[c->pos], -1, SQLITE_TRANSIENT); break;
    case GPR_RANK: sqlite3_result_double(ctx, c->rank[c->pos]); break;
    default: sqlite3_result_null(ctx); break;
    }
    return SQLITE_OK;
}

static sqlite3_module pagerank_module = {
    .iVersion = 0, .xCreate = 0, .xConnect = gpr_connect, .xBestIndex = gpr_best_index, .xDisconnect = tvf_disconnect,
    .xDestroy = tvf_disconnect, .xOpen = tvf_open, .xClose = tvf_close, .xFilter = gpr_filter, .xNext = tvf_next,
    .xEof = tvf_eof, .xColumn = gpr_column, .xRowid = tvf_rowid,
};

/* ───────────────────────── graph_node_betweenness / graph_edge_betweenness (src/graph_centrality.c:752-1240) ───────────────────────── */

int mn_sql_load_graph(sqlite3 *db, const char *who, const char *edge_table, const char *src_col, const char *dst_col,
                      const char *weight_col, const char *direction, const char *ts_col, sqlite3_value *t0, sqlite3_value *t1,
                      mn_graph **g_out, char ***ids_out, int *n_out, char **err); /* mn_graph_sql.c */

typedef struct {
    sqlite3_vtab_cursor base;
    char **ids;       /* node ids, first-seen order (owned) */
    int n_ids;
    double *cb;       /* node betweenness */
    int *e_src, *e_dst; /* edge rows */
    double *e_c;
    int n, pos, eof;  /* rows */
} BetCursor;

static void bet_clear(BetCursor *c) {
    for (int i = 0; i < c->n_ids; i++)
        free(c->ids[i]);
    free(c->ids);
    free(c->cb);
    free(c->e_src);
    free(c->e_dst);
    free(c->e_c);
    c->ids = 0;
    c->cb = c->e_c = 0;
    c->e_src = c->e_dst = 0;
    c->n_ids = c->n = 0;
}

static int bet_open(sqlite3_vtab *v, sqlite3_vtab_cursor **out) {
    (void)v;
    BetCursor *c = (BetCursor *)calloc(1, sizeof(BetCursor));
    if (!c)
        return SQLITE_NOMEM;
    c->eof = 1;
    *out = &c->base;
    return SQLITE_OK;
}
static int bet_close(sqlite3_vtab_cursor *cur) {
    bet_clear((BetCursor *)cur);
    free(cur);
    return SQLITE_OK;
}
static int bet_next(sqlite3_vtab_cursor *cur) {
    BetCursor *c = (BetCursor *)cur;
    c->pos++;
    c->eof = c->pos >= c->n;
    return SQLITE_OK;
}
static int bet_eof(sqlite3_vtab_cursor *cur) { return ((BetCursor *)cur)->eof; }
static int bet_rowid(sqlite3_vtab_cursor *cur, sqlite3_int64 *out) {
    *out = ((BetCursor *)cur)->pos;
    return SQLITE_OK;
}

/* graph_best_index_common (src/graph_common.h:62-96): argv compacted in column order, idxNum = bitmask of the hidden columns present */
static int common_best_index(sqlite3_index_info *ii, int first_hidden, int last_hidden, double good_cost) {
    int which[32];
    const int ncols = last_hidden - first_hidden + 1;
    for (int j = 0; j < ncols; j++)
        which[j] = -1;
    for (int i = 0; i < ii->nConstraint; i++) {
        if (!ii->aConstraint[i].usable || ii->aConstraint[i].op != SQLITE_INDEX_CONSTRAINT_EQ)
            continue;
        int col = ii->aConstraint[i].iColumn;
        if (col >= first_hidden && col <= last_hidden)
            which[col - first_hidden] = i;
    }
    int arg = 1, mask = 0;
    for (int j = 0; j < ncols; j++)
        if (which[j] >= 0) {
            ii->aConstraintUsage[which[j]].argvIndex = arg++;
            ii->aConstraintUsage[which[j]].omit = 1;
            mask |= 1 << j;
        }
    ii->idxNum = mask;
    ii->estimatedCost = (mask & 0x7) == 0x7 ? good_cost : 1e12;
    return SQLITE_OK;
}

static const char *val_text(sqlite3_value *v) { /* graph_safe_text */
    return v && sqlite3_value_type(v) != SQLITE_NULL ? (const char *)sqlite3_value_text(v) : 0;
}

/* hidden columns, in the order both TVFs declare them */
enum { BH_EDGE_TABLE = 0, BH_SRC, BH_DST, BH_WEIGHT, BH_NORMALIZED, BH_DIRECTION, BH_APPROX, BH_TS, BH_T0, BH_T1, BH_COUNT };

/* shared xFilter: edges = 0 → node rows, 1 → edge rows */
static int bet_filter_common(sqlite3_vtab_cursor *cur, int idxNum, int argc, sqlite3_value **argv, int edges) {
    BetCursor *c = (BetCursor *)cur;
    TvfVtab *vt = (TvfVtab *)cur->pVtab;
    const char *who = edges ? "graph_edge_betweenness" : "graph_node_betweenness";
    bet_clear(c);
    c->pos = 0;
    c->eof = 1;
    if (argc < 3)
        return SQLITE_OK;
    const char *edge_table = 0, *src_col = 0, *dst_col = 0, *weight_col = 0, *direction = 0, *ts_col = 0;
    sqlite3_value *t0 = 0, *t1 = 0;
    int normalized = 0, auto_approx = edges ? 0 : 50000; /* :838 / :1082 */
    int pos = 0;
    for (int bit = 0; bit < BH_COUNT && pos < argc; bit++) {
        if (!(idxNum & (1 << bit)))
            continue;
        switch (bit) {
        case BH_EDGE_TABLE: edge_table = val_text(argv[pos]); break;
        case BH_SRC: src_col = val_text(argv[pos]); break;
        case BH_DST: dst_col = val_text(argv[pos]); break;
        case BH_WEIGHT: weight_col = val_text(argv[pos]); break;
        case BH_NORMALIZED: normalized = sqlite3_value_int(argv[pos]); break;
        case BH_DIRECTION: direction = val_text(argv[pos]); break;
        case BH_APPROX: auto_approx = sqlite3_value_int(argv[pos]); break;
        case BH_TS: ts_col = val_text(argv[pos]); break;
        case BH_T0: t0 = argv[pos]; break;
        case BH_T1: t1 = argv[pos]; break;
        }
        pos++;
    }
    if (!direction)
        direction = "forward"; /* :872-873 */
    mn_graph *g = 0;
    char *err = 0;
    int n = 0;
    if (mn_sql_load_graph(vt->db, who, edge_table, src_col, dst_col, weight_col, direction, ts_col, t0, t1, &g, &c->ids, &n, &err) !=
        SQLITE_OK) {
        vt->base.zErrMsg = err ? err : sqlite3_mprintf("%s: failed to load graph", who);
        return SQLITE_ERROR;
    }
    c->n_ids = n;
    if (!g)
        return SQLITE_OK; /* empty graph: no rows */
    const int dir = !strcmp(direction, "both") ? 0 : !strcmp(direction, "reverse") ? 2 : 1;
    c->cb = (double *)calloc((size_t)n, sizeof(double));
    double *eb = edges ? (double *)calloc((size_t)n * (size_t)n, sizeof(double)) : 0; /* the reference's dense N x N (:1146) */
    if (!c->cb || (edges && !eb)) {
        free(eb);
        mn_graph_destroy(g);
        return SQLITE_NOMEM;
    }
    if (mn_graph_betweenness(g, dir, auto_approx, normalized, c->cb, eb) != 0) {
        vt->base.zErrMsg = sqlite3_mprintf("%s: %s", who, mn_graph_last_error());
        free(eb);
        mn_graph_destroy(g);
        return SQLITE_ERROR;
    }
    if (!edges) {
        c->n = n;
    } else { /* only edges that exist in GraphData.out, in list order, with a positive value (:1172-1182) */
        const long long ne = mn_graph_out_edge_count(g);
        int *off = (int *)malloc(((size_t)n + 1) * sizeof(int)), *tgt = (int *)malloc((size_t)(ne ? ne : 1) * sizeof(int));
        c->e_src = (int *)malloc((size_t)(ne ? ne : 1) * sizeof(int));
        c->e_dst = (int *)malloc((size_t)(ne ? ne : 1) * sizeof(int));
        c->e_c = (double *)malloc((size_t)(ne ? ne : 1) * sizeof(double));
        if (mn_graph_out_lists(g, off, tgt) != 0) {
            vt->base.zErrMsg = sqlite3_mprintf("%s: %s", who, mn_graph_last_error());
            free(off); free(tgt); free(eb);
            mn_graph_destroy(g);
            return SQLITE_ERROR;
        }
        int rows = 0;
        for (int i = 0; i < n; i++)
            for (int x = off[i]; x < off[i + 1]; x++) {
                const double v = eb[(size_t)i * n + tgt[x]];
                if (v > 0.0) {
                    c->e_src[rows] = i;
                    c->e_dst[rows] = tgt[x];
                    c->e_c[rows++] = v;
                }
            }
        c->n = rows;
        free(off);
        free(tgt);
    }
    free(eb);
    mn_graph_destroy(g);
    c->eof = c->n == 0;
    return SQLITE_OK;
}

static int nbet_connect(sqlite3 *db, void *aux, int argc, const char *const *argv, sqlite3_vtab **out, char **err) {
    (void)aux; (void)argc; (void)argv; (void)err;
    return tvf_connect_with(db, "CREATE TABLE x("
                                "  node TEXT, centrality REAL,"
                                "  edge_table TEXT HIDDEN, src_col TEXT HIDDEN, dst_col TEXT HIDDEN,"
                                "  weight_col TEXT HIDDEN, normalized INTEGER HIDDEN,"
                                "  direction TEXT HIDDEN, auto_approx_threshold INTEGER HIDDEN,"
                                "  timestamp_col TEXT HIDDEN, time_start HIDDEN, time_end HIDDEN"
                                ")", out);
}
static int nbet_best_index(sqlite3_vtab *v, sqlite3_index_info *ii) {
    (void)v;
    return common_best_index(ii, 2, 11, 1000.0);
}
static int nbet_filter(sqlite3_vtab_cursor *cur, int idxNum, const char *idxStr, int argc, sqlite3_value **argv) {
    (void)idxStr;
    return bet_filter_common(cur, idxNum, argc, argv, 0);
}
static int nbet_column(sqlite3_vtab_cursor *cur, sqlite3_context *ctx, int col) {
    BetCursor *c = (BetCursor *)cur;
    switch (col) {
    case 0: sqlite3_result_text(ctx, c->ids[c->pos], -1, SQLITE_TRANSIENT); break;
    case 1: sqlite3_result_double(ctx, c->cb[c->pos]); break;
    default: sqlite3_result_null(ctx); break;
    }
    return SQLITE_OK;
}
static sqlite3_module node_betweenness_module = {
    .iVersion = 0, .xCreate = 0, .xConnect = nbet_connect, .xBestIndex = nbet_best_index, .xDisconnect = tvf_disconnect,
    .xDestroy = tvf_disconnect, .xOpen = bet_open, .xClose = bet_close, .xFilter = nbet_filter, .xNext = bet_next,
    .xEof = bet_eof, .xColumn = nbet_column, .xRowid = bet_rowid,
};

static int ebet_connect(sqlite3 *db, void *aux, int argc, const char *const *argv, sqlite3_vtab **out, char **err) {
    (void)aux; (void)argc; (void)argv; (void)err;
    return tvf_connect_with(db, "CREATE TABLE x("
                                "  src TEXT, dst TEXT, centrality REAL,"
                                "  edge_table TEXT HIDDEN, src_col TEXT HIDDEN, dst_col TEXT HIDDEN,"
                                "  weight_col TEXT HIDDEN, normalized INTEGER HIDDEN,"
                                "  direction TEXT HIDDEN, auto_approx_threshold INTEGER HIDDEN,"
                                "  timestamp_col TEXT HIDDEN, time_start HIDDEN, time_end HIDDEN"
                                ")", out);
}
static int ebet_best_index(sqlite3_vtab *v, sqlite3_index_info *ii) {
    (void)v;
    return common_best_index(ii, 3, 12, 1000.0);
}
static int ebet_filter(sqlite3_vtab_cursor *cur, int idxNum, const char *idxStr, int argc, sqlite3_value **argv) {
    (void)idxStr;
    return bet_filter_common(cur, idxNum, argc, argv, 1);
}
static int ebet_column(sqlite3_vtab_cursor *cur, sqlite3_context *ctx, int col) {
    BetCursor *c = (BetCursor *)cur;
    switch (col) {
    case 0: sqlite3_result_text(ctx, c->ids[c->e_src[c->pos]], -1, SQLITE_TRANSIENT); break;
    case 1: sqlite3_result_text(ctx, c->ids[c->e_dst[c->pos]], -1, SQLITE_TRANSIENT); break;
    case 2: sqlite3_result_double(ctx, c->e_c[c->pos]); break;
    default: sqlite3_result_null(ctx); break;
    }
    return SQLITE_OK;
}
static sqlite3_module edge_betweenness_module = {
    .iVersion = 0, .xCreate = 0, .xConnect = ebet_connect, .xBestIndex = ebet_best_index, .xDisconnect = tvf_disconnect,
    .xDestroy = tvf_disconnect, .xOpen = bet_open, .xClose = bet_close, .xFilter = ebet_filter, .xNext = bet_next,
    .xEof = bet_eof, .xColumn = ebet_column, .xRowid = bet_rowid,
};

int mn_register_graph_tvfs(sqlite3 *db) { /* the order of graph_register_tvfs (src/graph_tvf.c:1898-1914) */
    int rc = sqlite3_create_module(db, "graph_components", &components_module, 0);
    if (rc == SQLITE_OK)
        rc = sqlite3_create_module(db, "graph_pagerank", &pagerank_module, 0);
    return rc;
}

int mn_register_betweenness_tvfs(sqlite3 *db) { /* two of centrality_register_tvfs (src/graph_centrality.c:1510-1530) */
    int rc = sqlite3_create_module(db, "graph_node_betweenness", &node_betweenness_module, 0);
    if (rc == SQLITE_OK)
        rc = sqlite3_create_module(db, "graph_edge_betweenness", &edge_betweenness_module, 0);
    return rc;
}
