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
    while (sqlite3_step(st) == SQLITE_ROW) {
        const char *a = (const char *)sqlite3_column_text(st, 0);
        const char *b = (const char *)sqlite3_column_text(st, 1);
        if (!a || !b)
            continue;
        if (n >= cap) {
            cap *= 2;
            s = (int *)realloc(s, (size_t)cap * sizeof(int));
            d = (int *)realloc(d, (size_t)cap * sizeof(int));
        }
        int si = nm_get(nm, a); /* (a's text pointer is only valid until the next column call: nm_get copies) */
        b = (const char *)sqlite3_column_text(st, 1);
        s[n] = si;
        d[n] = nm_get(nm, b);
        n++;
    }
    sqlite3_finalize(st);
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
        if (mn_graph_components(n, ne, src, dst, components_mode(ne), 0, c->comp_id, c->comp_size, 0) != 0) {
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
        if (mn_graph_pagerank(n, ne, src, dst, damping, iterations, 0, c->rank, 0) != 0) {
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

int mn_register_graph_tvfs(sqlite3 *db) { /* the order of graph_register_tvfs (src/graph_tvf.c:1898-1914) */
    int rc = sqlite3_create_module(db, "graph_components", &components_module, 0);
    if (rc == SQLITE_OK)
        rc = sqlite3_create_module(db, "graph_pagerank", &pagerank_module, 0);
    return rc;
}
