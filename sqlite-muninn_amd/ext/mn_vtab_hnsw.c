/*
 * mn_vtab_hnsw.c — the `hnsw_index` virtual table (and its `hnsw0` alias) over the device-resident
 * index of libmuninn_hip.so.
 *
 * Drop-in for the reference's module (src/hnsw_vtab.c): same CREATE arguments and error strings
 * (:80-134), same declared schema (:366-367), same query plans (:498-550), same shadow tables and
 * config keys (:138-199) so databases written by either implementation open in the other, same
 * xUpdate contract (:686-784).  What differs is underneath: the index lives in HBM and every
 * hnsw_* call is the C-ABI of include/muninn_hip.h (k_insert_seq keeps the reference's
 * one-at-a-time insert semantics, so the persisted graph is the one the reference would persist).
 *
 * Persistence.  The reference rewrites the shadow rows of the new node and of each of
 * its neighbours inside every xUpdate (src/hnsw_vtab.c:755-776), which costs more than the insert
 * itself.  Here xUpdate validates and queues the row; the queue is applied to the device index in one
 * call and the shadow tables are written once from the device's accumulated persist set
 * (mn_hnsw_take_dirty).  MUNINN_HNSW_MODE (read when the table is connected) says when:
 *   exact    (default) after every row — shadow tables are current after each INSERT, as the reference's;
 *   deferred at xSync, or earlier whenever something reads the table (xFilter, DELETE, hnsw_search_batch)
 *            or the queue is full.  Rows are applied in arrival order with the reference's one-at-a-time
 *            semantics: the index, the rowids handed out and the COMMITTED shadow tables are identical to
 *            exact mode; only a direct SELECT on "{t}_nodes"/"{t}_edges" inside the open transaction
 *            sees them late;
 *   fast     as deferred, but the queue is built with the batch-synchronous schedule (DESIGN.md §4:
 *            a different, equally good graph; orders of magnitude faster for bulk loads).
 */
#include "../../include/muninn_hip.h"
#include "mn_sqlite_abi.h"
#include "mn_nodemap.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    sqlite3_vtab base;
    sqlite3 *db;
    char *name;
    mn_index *index;
    int dim, metric, m, efc;
    /* rows accepted by xUpdate that are not in the device index yet */
    sqlite3_int64 *pend_ids;
    float *pend_vecs;
    int n_pend, cap_pend;
    sqlite3_int64 *pset_key; /* open-addressing set of the queued rowids */
    unsigned char *pset_used;
    int pset_cap;
    /* sp_mark[n] = rows queued when savepoint n was established (xSavepoint; statement savepoints included), -1 = not seen:
     * ROLLBACK TO n cuts the queue back to it, otherwise xSync would insert and persist rows the user rolled back */
    int *sp_mark;
    int sp_cap;
    int mode; /* MUNINN_HNSW_MODE */
    /* exact mode, one row at a time: the device says which edges the insert added and removed, and only those rows of
     * "{t}_edges" are written (insert_one_delta) — valid for nodes whose shadow rows equal the index.  A ROLLBACK takes rows
     * away that the index (like the reference's in-memory one) keeps, and a DELETE edits lists the reference never persists:
     * the index then reports "no log" for inserts touching such nodes (mn_hnsw_log_invalidate / mn_hnsw_delete) and they
     * get the reference's whole-node rewrite, which heals them */
    int delta_ok;
    mn_edge_change *chlog;
    sqlite3_stmt *st_node, *st_edge_put, *st_edge_del, *st_cfg;
} VtabHnsw;
#define MN_DELTA_CAP 1024

enum { MODE_EXACT = 0, MODE_DEFERRED = 1, MODE_FAST = 2 };
#define MN_PEND_MAX_BYTES ((size_t)256 << 20) /* queue is applied when its vectors reach this size ... */
#define MN_PEND_MAX_ROWS 65536                /* ... or this many rows (fast mode: 16x both) */

typedef struct {
    sqlite3_vtab_cursor base;
    mn_search_result *hits;
    int n_hits, pos;
    sqlite3_int64 point_id;
    int is_point, eof;
    float *vecbuf;
} CurHnsw;

/* live hnsw_index tables of this process, so that hnsw_search_batch can reach an index by name.  Connections may live
 * on different threads (SURVEY §8b "Threading": one host thread per connection, several connections per process), so the
 * list is guarded; it grows as needed (a table beyond a fixed capacity would silently become unreachable). */
#include <pthread.h>
static pthread_mutex_t g_live_mu = PTHREAD_MUTEX_INITIALIZER;
static VtabHnsw **g_live = 0;
static int g_live_cap = 0;

static void live_add(VtabHnsw *v) {
    pthread_mutex_lock(&g_live_mu);
    int slot = -1;
    for (int i = 0; i < g_live_cap && slot < 0; i++)
        if (!g_live[i])
            slot = i;
    if (slot < 0) {
        const size_t nc = g_live_cap > 0 ? 2 * (size_t)g_live_cap : 64;
        VtabHnsw **n = nc <= (1u << 20) ? (VtabHnsw **)realloc(g_live, nc * sizeof(*n)) : 0;
        if (n) {
            memset(n + g_live_cap, 0, (nc - (size_t)g_live_cap) * sizeof(*n));
            slot = g_live_cap;
            g_live = n;
            g_live_cap = (int)nc;
        }
    }
    if (slot >= 0)
        g_live[slot] = v;
    pthread_mutex_unlock(&g_live_mu);
}
static void live_remove(VtabHnsw *v) {
    pthread_mutex_lock(&g_live_mu);
    for (int i = 0; i < g_live_cap; i++)
        if (g_live[i] == v)
            g_live[i] = 0;
    pthread_mutex_unlock(&g_live_mu);
}
static VtabHnsw *live_find(sqlite3 *db, const char *name) {
    VtabHnsw *hit = 0;
    pthread_mutex_lock(&g_live_mu);
    for (int i = 0; i < g_live_cap && !hit; i++)
        if (g_live[i] && g_live[i]->db == db && !strcmp(g_live[i]->name, name))
            hit = g_live[i];
    pthread_mutex_unlock(&g_live_mu);
    return hit;
}

enum { COL_VECTOR = 0, COL_DISTANCE = 1, COL_K = 2, COL_EF = 3 };
enum { PLAN_SCAN = 0, PLAN_KNN = 1, PLAN_POINT = 2 };

static const char *SCHEMA = "CREATE TABLE x(vector BLOB, distance REAL, k INTEGER HIDDEN, ef_search INTEGER HIDDEN)";

typedef struct {
    int dimensions, metric, m, efc;
} Params;

/* key=value arguments of CREATE VIRTUAL TABLE (src/hnsw_vtab.c:80-134) */
static int parse_args(int argc, const char *const *argv, Params *p, char **err) {
    p->dimensions = 0;
    p->metric = MN_METRIC_COSINE;
    p->m = 16;
    p->efc = 200;
    for (int i = 3; i < argc; i++) {
        const char *a = argv[i];
        if (!strncmp(a, "dimensions=", 11)) {
            p->dimensions = atoi(a + 11);
            if (p->dimensions <= 0) {
                *err = sqlite3_mprintf("hnsw_index: dimensions must be > 0, got '%s'", a + 11);
                return SQLITE_ERROR;
            }
        } else if (!strncmp(a, "metric=", 7)) {
            const char *v = a + 7;
            char buf[32];
            size_t n = strlen(v);
            if (n >= 2 && (v[0] == '\'' || v[0] == '"')) {
                n -= 2;
                if (n >= sizeof(buf))
                    n = sizeof(buf) - 1;
                memcpy(buf, v + 1, n);
                buf[n] = 0;
                v = buf;
            }
            if (mn_vec_parse_metric(v, &p->metric) != 0) {
                *err = sqlite3_mprintf("hnsw_index: unknown metric '%s' (use 'l2', 'cosine', or 'inner_product')", v);
                return SQLITE_ERROR;
            }
        } else if (!strncmp(a, "m=", 2)) {
            p->m = atoi(a + 2);
            if (p->m < 2) {
                *err = sqlite3_mprintf("hnsw_index: m must be >= 2, got '%s'", a + 2);
                return SQLITE_ERROR;
            }
        } else if (!strncmp(a, "ef_construction=", 16)) {
            p->efc = atoi(a + 16);
            if (p->efc < 1) {
                *err = sqlite3_mprintf("hnsw_index: ef_construction must be >= 1, got '%s'", a + 16);
                return SQLITE_ERROR;
            }
        } else {
            *err = sqlite3_mprintf("hnsw_index: unknown parameter '%s'", a);
            return SQLITE_ERROR;
        }
    }
    if (p->dimensions == 0) {
        *err = sqlite3_mprintf("hnsw_index: 'dimensions' parameter is required");
        return SQLITE_ERROR;
    }
    return SQLITE_OK;
}

static int run_sql(sqlite3 *db, char *sql) {
    if (!sql)
        return SQLITE_NOMEM;
    int rc = sqlite3_exec(db, sql, 0, 0, 0);
    sqlite3_free(sql);
    return rc;
}

/* src/hnsw_vtab.c:138-181 — byte-compatible shadow schema */
static int make_shadow_tables(sqlite3 *db, const char *t) {
    int rc = run_sql(db, sqlite3_mprintf("CREATE TABLE IF NOT EXISTS \"%w_config\" (key TEXT PRIMARY KEY, value TEXT NOT NULL)", t));
    if (rc == SQLITE_OK)
        rc = run_sql(db, sqlite3_mprintf("CREATE TABLE IF NOT EXISTS \"%w_nodes\" (  id INTEGER PRIMARY KEY,  vector BLOB NOT NULL,"
                                         "  level INTEGER NOT NULL,  deleted INTEGER NOT NULL DEFAULT 0)", t));
    if (rc == SQLITE_OK)
        rc = run_sql(db, sqlite3_mprintf("CREATE TABLE IF NOT EXISTS \"%w_edges\" (  source_id INTEGER NOT NULL,"
                                         "  target_id INTEGER NOT NULL,  level INTEGER NOT NULL,  distance REAL NOT NULL,"
                                         "  PRIMARY KEY (source_id, level, target_id)) WITHOUT ROWID", t));
    if (rc == SQLITE_OK)
        rc = run_sql(db, sqlite3_mprintf("CREATE INDEX IF NOT EXISTS \"%w_edges_rev\" ON \"%w_edges\"(target_id, level)", t, t));
    return rc;
}

/* src/hnsw_vtab.c:183-199 */
static int write_config(VtabHnsw *v) {
    if (v->st_cfg) { /* (the same six upserts, prepared once: persist_delta runs this per row) */
        char ep[24], ml[16];
        snprintf(ep, sizeof(ep), "%lld", (long long)mn_hnsw_entry_point(v->index));
        snprintf(ml, sizeof(ml), "%d", mn_hnsw_max_level(v->index));
        sqlite3_bind_text(v->st_cfg, 1, ep, -1, SQLITE_STATIC);
        sqlite3_bind_text(v->st_cfg, 2, ml, -1, SQLITE_STATIC);
        int rc = sqlite3_step(v->st_cfg) == SQLITE_DONE ? SQLITE_OK : SQLITE_ERROR;
        sqlite3_reset(v->st_cfg);
        return rc;
    }
    return run_sql(v->db, sqlite3_mprintf("INSERT OR REPLACE INTO \"%w_config\" (key, value) VALUES ('dimensions', '%d'),"
                                          " ('metric', '%d'), ('m', '%d'), ('ef_construction', '%d'),"
                                          " ('entry_point', '%lld'), ('max_level', '%d')",
                                          v->name, v->dim, v->metric, v->m, v->efc,
                                          (long long)mn_hnsw_entry_point(v->index), mn_hnsw_max_level(v->index)));
}

/* src/hnsw_vtab.c:201-234 */
static int read_config(sqlite3 *db, const char *t, Params *p, sqlite3_int64 *entry, int *max_level) {
    p->dimensions = 0;
    p->metric = MN_METRIC_COSINE;
    p->m = 16;
    p->efc = 200;
    *entry = -1;
    *max_level = -1;
    char *sql = sqlite3_mprintf("SELECT key, value FROM \"%w_config\"", t);
    sqlite3_stmt *st = 0;
    int rc = sqlite3_prepare_v2(db, sql, -1, &st, 0);
    sqlite3_free(sql);
    if (rc != SQLITE_OK)
        return rc;
    while (sqlite3_step(st) == SQLITE_ROW) {
        const char *k = (const char *)sqlite3_column_text(st, 0);
        const char *val = (const char *)sqlite3_column_text(st, 1);
        if (!k || !val)
            continue;
        if (!strcmp(k, "dimensions")) p->dimensions = atoi(val);
        else if (!strcmp(k, "metric")) p->metric = atoi(val);
        else if (!strcmp(k, "m")) p->m = atoi(val);
        else if (!strcmp(k, "ef_construction")) p->efc = atoi(val);
        else if (!strcmp(k, "entry_point")) *entry = atoll(val);
        else if (!strcmp(k, "max_level")) *max_level = atoi(val);
    }
    sqlite3_finalize(st);
    return SQLITE_OK;
}

static int node_is_live(VtabHnsw *v, sqlite3_int64 id) { /* hnsw_get_node != NULL (src/hnsw_algo.c:226-231) */
    return mn_hnsw_node_level(v->index, id) >= 0 && mn_hnsw_node_deleted(v->index, id) == 0;
}

/* ── queue of accepted rows ── */
static unsigned long long pset_hash(sqlite3_int64 id) {
    unsigned long long h = (unsigned long long)id * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}
static int pset_has(const VtabHnsw *v, sqlite3_int64 id) {
    if (!v->n_pend)
        return 0;
    for (unsigned long long i = pset_hash(id);; i++) {
        int p = (int)(i & (unsigned long long)(v->pset_cap - 1));
        if (!v->pset_used[p])
            return 0;
        if (v->pset_key[p] == id)
            return 1;
    }
}
static void pset_put(VtabHnsw *v, sqlite3_int64 id) {
    for (unsigned long long i = pset_hash(id);; i++) {
        int p = (int)(i & (unsigned long long)(v->pset_cap - 1));
        if (!v->pset_used[p]) {
            v->pset_used[p] = 1;
            v->pset_key[p] = id;
            return;
        }
    }
}
static void pend_clear(VtabHnsw *v) {
    v->n_pend = 0;
    if (v->pset_used)
        memset(v->pset_used, 0, (size_t)v->pset_cap);
    for (int i = 0; i < v->sp_cap; i++) /* the queue starts again: every open savepoint now stands at its beginning */
        if (v->sp_mark[i] > 0)
            v->sp_mark[i] = 0;
}
/* cut the queue back to its first n rows (ROLLBACK TO) and rebuild the rowid set from what is kept */
static void pend_truncate(VtabHnsw *v, int n) {
    if (n >= v->n_pend)
        return;
    v->n_pend = n;
    if (v->pset_used)
        memset(v->pset_used, 0, (size_t)v->pset_cap);
    for (int i = 0; i < n; i++)
        pset_put(v, v->pend_ids[i]);
}
static void sp_forget(VtabHnsw *v, int from) {
    for (int i = from < 0 ? 0 : from; i < v->sp_cap; i++)
        v->sp_mark[i] = -1;
}
static void pend_free(VtabHnsw *v) {
    free(v->pend_ids);
    free(v->pend_vecs);
    free(v->pset_key);
    free(v->pset_used);
    free(v->sp_mark);
    v->sp_mark = 0;
    v->sp_cap = 0;
    v->pend_ids = 0;
    v->pend_vecs = 0;
    v->pset_key = 0;
    v->pset_used = 0;
    v->n_pend = v->cap_pend = v->pset_cap = 0;
}
static int pend_add(VtabHnsw *v, sqlite3_int64 id, const float *vec) {
    if (v->n_pend == v->cap_pend) {
        int nc = v->cap_pend ? v->cap_pend * 2 : 256;
        sqlite3_int64 *ni = (sqlite3_int64 *)realloc(v->pend_ids, (size_t)nc * sizeof(sqlite3_int64));
        if (!ni)
            return SQLITE_NOMEM;
        v->pend_ids = ni;
        float *nv = (float *)realloc(v->pend_vecs, (size_t)nc * v->dim * sizeof(float));
        if (!nv)
            return SQLITE_NOMEM;
        v->pend_vecs = nv;
        v->cap_pend = nc;
    }
    if ((v->n_pend + 1) * 2 > v->pset_cap) { /* keep the set at most half full */
        int nc = v->pset_cap ? v->pset_cap * 2 : 1024;
        sqlite3_int64 *nk = (sqlite3_int64 *)malloc((size_t)nc * sizeof(sqlite3_int64));
        unsigned char *nu = (unsigned char *)calloc((size_t)nc, 1);
        if (!nk || !nu) {
            free(nk);
            free(nu);
            return SQLITE_NOMEM;
        }
        free(v->pset_key);
        free(v->pset_used);
        v->pset_key = nk;
        v->pset_used = nu;
        v->pset_cap = nc;
        for (int i = 0; i < v->n_pend; i++)
            pset_put(v, v->pend_ids[i]);
    }
    v->pend_ids[v->n_pend] = id;
    memcpy(v->pend_vecs + (size_t)v->n_pend * v->dim, vec, (size_t)v->dim * sizeof(float));
    v->n_pend++;
    pset_put(v, id);
    return SQLITE_OK;
}

/* Shadow rows for everything the device marked since the last call: persist_node (src/hnsw_vtab.c:237-283)
 * for every new node and every node that was a neighbour of a new node when it was linked (:755-768),
 * once per node instead of once per insert that touched it.  new_ids/new_vecs = the queue just applied. */
static int persist_marked(VtabHnsw *v, const sqlite3_int64 *new_ids, const float *new_vecs, int n_new) {
    sqlite3_stmt *st = 0;
    char *sql = sqlite3_mprintf("INSERT OR REPLACE INTO \"%w_nodes\" (id, vector, level, deleted) VALUES (?, ?, ?, 0)", v->name);
    int rc = sqlite3_prepare_v2(v->db, sql, -1, &st, 0);
    sqlite3_free(sql);
    if (rc != SQLITE_OK)
        return rc;
    for (int i = 0; i < n_new && rc == SQLITE_OK; i++) {
        sqlite3_bind_int64(st, 1, new_ids[i]);
        sqlite3_bind_blob(st, 2, new_vecs + (size_t)i * v->dim, v->dim * (int)sizeof(float), SQLITE_STATIC);
        sqlite3_bind_int(st, 3, mn_hnsw_node_level(v->index, new_ids[i]));
        rc = sqlite3_step(st) == SQLITE_DONE ? SQLITE_OK : SQLITE_ERROR;
        sqlite3_reset(st);
    }
    sqlite3_finalize(st);
    if (rc != SQLITE_OK)
        return rc;

    int64_t cap = (int64_t)n_new * (1 + v->m) + 1024, nd;
    int64_t *marked = 0;
    for (;;) {
        marked = (int64_t *)malloc((size_t)cap * sizeof(int64_t));
        if (!marked)
            return SQLITE_NOMEM;
        nd = mn_hnsw_take_dirty(v->index, marked, cap);
        if (nd <= cap)
            break;
        free(marked);
        cap = nd;
    }
    if (nd < 0) {
        free(marked);
        sqlite3_free(v->base.zErrMsg);
        v->base.zErrMsg = sqlite3_mprintf("hnsw_index: cannot read the persist set (%s)", mn_last_error());
        return SQLITE_ERROR;
    }
    /* persist_node upserts the node row of every neighbour as well (src/hnsw_vtab.c:243-256).  For a row that exists
     * that changes nothing; it matters after a ROLLBACK, which takes shadow rows away while the index (the reference's
     * in-memory one, ours in HBM) keeps the nodes: the next insert that links to such a node brings its row back. */
    {
        sqlite3_stmt *has = 0, *put = 0;
        sql = sqlite3_mprintf("SELECT 1 FROM \"%w_nodes\" WHERE id = ?", v->name);
        rc = sqlite3_prepare_v2(v->db, sql, -1, &has, 0);
        sqlite3_free(sql);
        if (rc == SQLITE_OK) {
            sql = sqlite3_mprintf("INSERT OR REPLACE INTO \"%w_nodes\" (id, vector, level, deleted) VALUES (?, ?, ?, 0)", v->name);
            rc = sqlite3_prepare_v2(v->db, sql, -1, &put, 0);
            sqlite3_free(sql);
        }
        float *vb = rc == SQLITE_OK ? (float *)malloc((size_t)v->dim * sizeof(float)) : 0;
        for (int64_t i = 0; i < nd && rc == SQLITE_OK && vb; i++) {
            if (pset_has(v, marked[i]))
                continue;
            sqlite3_bind_int64(has, 1, marked[i]);
            const int present = sqlite3_step(has) == SQLITE_ROW;
            sqlite3_reset(has);
            if (present || mn_hnsw_get_vector(v->index, marked[i], vb) != 0)
                continue;
            sqlite3_bind_int64(put, 1, marked[i]);
            sqlite3_bind_blob(put, 2, vb, v->dim * (int)sizeof(float), SQLITE_STATIC);
            sqlite3_bind_int(put, 3, mn_hnsw_node_level(v->index, marked[i]));
            sqlite3_step(put);
            sqlite3_reset(put);
        }
        free(vb);
        sqlite3_finalize(has);
        sqlite3_finalize(put);
        if (rc != SQLITE_OK) {
            free(marked);
            return rc;
        }
    }
    /* old edges of the pre-existing marked nodes (new nodes have none yet) */
    sql = sqlite3_mprintf("DELETE FROM \"%w_edges\" WHERE source_id = ?", v->name);
    rc = sqlite3_prepare_v2(v->db, sql, -1, &st, 0);
    sqlite3_free(sql);
    if (rc != SQLITE_OK) {
        free(marked);
        return rc;
    }
    for (int64_t i = 0; i < nd; i++) {
        if (pset_has(v, marked[i]))
            continue;
        sqlite3_bind_int64(st, 1, marked[i]);
        sqlite3_step(st);
        sqlite3_reset(st);
    }
    sqlite3_finalize(st);

    sql = sqlite3_mprintf("INSERT INTO \"%w_edges\" (source_id, target_id, level, distance) VALUES (?, ?, ?, ?)", v->name);
    rc = sqlite3_prepare_v2(v->db, sql, -1, &st, 0);
    sqlite3_free(sql);
    if (rc != SQLITE_OK) {
        free(marked);
        return rc;
    }
    const int CHUNK = 8192; /* nodes per device round trip */
    int64_t ecap = (int64_t)CHUNK * 3 * v->m + 64;
    int64_t *src = (int64_t *)malloc((size_t)ecap * sizeof(int64_t));
    int64_t *dst = (int64_t *)malloc((size_t)ecap * sizeof(int64_t));
    int *lvl = (int *)malloc((size_t)ecap * sizeof(int));
    float *dist = (float *)malloc((size_t)ecap * sizeof(float));
    for (int64_t pos = 0; pos < nd && rc == SQLITE_OK; pos += CHUNK) {
        int n = (int)(nd - pos < CHUNK ? nd - pos : CHUNK);
        int64_t ne = -1;
        if (src && dst && lvl && dist)
            ne = mn_hnsw_edges_of(v->index, marked + pos, n, src, dst, lvl, dist, ecap);
        if (ne > ecap) { /* more edges than the guess: grow and ask again */
            free(src); free(dst); free(lvl); free(dist);
            ecap = ne;
            src = (int64_t *)malloc((size_t)ecap * sizeof(int64_t));
            dst = (int64_t *)malloc((size_t)ecap * sizeof(int64_t));
            lvl = (int *)malloc((size_t)ecap * sizeof(int));
            dist = (float *)malloc((size_t)ecap * sizeof(float));
            pos -= CHUNK;
            continue;
        }
        if (ne < 0) {
            sqlite3_free(v->base.zErrMsg);
            v->base.zErrMsg = sqlite3_mprintf("hnsw_index: cannot read neighbour lists (%s)", mn_last_error());
            rc = SQLITE_ERROR;
            break;
        }
        for (int64_t e = 0; e < ne; e++) {
            sqlite3_bind_int64(st, 1, src[e]);
            sqlite3_bind_int64(st, 2, dst[e]);
            sqlite3_bind_int(st, 3, lvl[e]);
            sqlite3_bind_double(st, 4, (double)dist[e]);
            sqlite3_step(st);
            sqlite3_reset(st);
        }
    }
    sqlite3_finalize(st);
    free(src); free(dst); free(lvl); free(dist); free(marked);
    if (rc == SQLITE_OK)
        rc = write_config(v);
    return rc;
}

/* MUNINN_PROFILE=1: where a flush's time goes (device insert vs shadow-table SQL), printed when the table disconnects */
static double g_prof_dev_s = 0.0, g_prof_sql_s = 0.0;
static long long g_prof_rows = 0;
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void delta_release(VtabHnsw *v) {
    sqlite3_finalize(v->st_node);
    sqlite3_finalize(v->st_edge_put);
    sqlite3_finalize(v->st_edge_del);
    sqlite3_finalize(v->st_cfg);
    v->st_node = v->st_edge_put = v->st_edge_del = v->st_cfg = 0;
    free(v->chlog);
    v->chlog = 0;
}
static void delta_off(VtabHnsw *v) {
    v->delta_ok = 0;
    delta_release(v);
}
static int delta_prepare(VtabHnsw *v) {
    if (v->st_node)
        return SQLITE_OK;
    v->chlog = (mn_edge_change *)malloc((size_t)MN_DELTA_CAP * sizeof(mn_edge_change));
    if (!v->chlog)
        return SQLITE_NOMEM;
    static const char *const Q[4] = {
        "INSERT OR REPLACE INTO \"%w_nodes\" (id, vector, level, deleted) VALUES (?, ?, ?, 0)",
        "INSERT OR REPLACE INTO \"%w_edges\" (source_id, target_id, level, distance) VALUES (?, ?, ?, ?)",
        "DELETE FROM \"%w_edges\" WHERE source_id = ? AND level = ? AND target_id = ?",
        0};
    sqlite3_stmt **dst[3] = {&v->st_node, &v->st_edge_put, &v->st_edge_del};
    int rc = SQLITE_OK;
    for (int i = 0; i < 3 && rc == SQLITE_OK; i++) {
        char *sql = sqlite3_mprintf(Q[i], v->name);
        rc = sql ? sqlite3_prepare_v2(v->db, sql, -1, dst[i], 0) : SQLITE_NOMEM;
        sqlite3_free(sql);
    }
    if (rc == SQLITE_OK) {
        char *sql = sqlite3_mprintf("INSERT OR REPLACE INTO \"%w_config\" (key, value) VALUES ('dimensions', '%d'),"
                                    " ('metric', '%d'), ('m', '%d'), ('ef_construction', '%d'), ('entry_point', ?1), ('max_level', ?2)",
                                    v->name, v->dim, v->metric, v->m, v->efc);
        rc = sql ? sqlite3_prepare_v2(v->db, sql, -1, &v->st_cfg, 0) : SQLITE_NOMEM;
        sqlite3_free(sql);
    }
    if (rc != SQLITE_OK)
        delta_release(v);
    return rc;
}

/* One exact insert and its shadow rows.  The reference re-persists the new node and every neighbour whole (persist_node,
 * src/hnsw_vtab.c:237-283, called at :755-776: ≈ 1 100 "{t}_edges" rows deleted and re-inserted per row at M = 16); the
 * device reports the edges this insert added and removed (mn_hnsw_insert_logged) and those ≈ 100 rows are written instead —
 * the same table contents, because every other row of the touched nodes is already what persist_node would write back.
 * When the log cannot describe the insert (*logged = 0) the caller does the whole-node rewrite. */
static int insert_one_delta(VtabHnsw *v, sqlite3_int64 id, const float *vec, int *logged, int *dev_rc, double *t_dev) {
    *logged = 0;
    int rc = delta_prepare(v);
    if (rc != SQLITE_OK)
        return rc;
    int n_log = -1;
    *dev_rc = mn_hnsw_insert_logged(v->index, id, vec, v->chlog, MN_DELTA_CAP, &n_log);
    *t_dev = now_s();
    if (*dev_rc != 0 || n_log < 0)
        return SQLITE_OK;
    *logged = 1;
    sqlite3_bind_int64(v->st_node, 1, id);
    sqlite3_bind_blob(v->st_node, 2, vec, v->dim * (int)sizeof(float), SQLITE_STATIC);
    sqlite3_bind_int(v->st_node, 3, mn_hnsw_node_level(v->index, id));
    rc = sqlite3_step(v->st_node) == SQLITE_DONE ? SQLITE_OK : SQLITE_ERROR;
    sqlite3_reset(v->st_node);
    for (int i = 0; i < n_log && rc == SQLITE_OK; i++) {
        const mn_edge_change *e = &v->chlog[i];
        sqlite3_stmt *st = e->op == 1 ? v->st_edge_put : v->st_edge_del;
        sqlite3_bind_int64(st, 1, e->src);
        if (e->op == 1) {
            sqlite3_bind_int64(st, 2, e->dst);
            sqlite3_bind_int(st, 3, e->level);
            sqlite3_bind_double(st, 4, (double)e->distance);
        } else {
            sqlite3_bind_int(st, 2, e->level);
            sqlite3_bind_int64(st, 3, e->dst);
        }
        rc = sqlite3_step(st) == SQLITE_DONE ? SQLITE_OK : SQLITE_ERROR;
        sqlite3_reset(st);
    }
    if (rc == SQLITE_OK)
        rc = write_config(v);
    if (rc != SQLITE_OK)
        delta_off(v); /* rows may be half written: whole-node rewrites from here on */
    return rc;
}

/* apply the queue to the device index (arrival order) and write the shadow tables */
static int flush_pending(VtabHnsw *v) {
    if (v->n_pend == 0)
        return SQLITE_OK;
    const double t0 = now_s();
    if (v->mode == MODE_EXACT && v->n_pend == 1 && v->delta_ok) {
        int logged = 0, dev_rc = 0;
        double t1 = t0;
        int rc = insert_one_delta(v, v->pend_ids[0], v->pend_vecs, &logged, &dev_rc, &t1);
        if (rc == SQLITE_OK && dev_rc != 0) {
            sqlite3_free(v->base.zErrMsg);
            v->base.zErrMsg = sqlite3_mprintf("hnsw_index: insert failed (%s)", mn_last_error());
            rc = SQLITE_ERROR;
        } else if (rc == SQLITE_OK && !logged) {
            rc = persist_marked(v, v->pend_ids, v->pend_vecs, 1);
        }
        g_prof_dev_s += t1 - t0;
        g_prof_sql_s += now_s() - t1;
        g_prof_rows += 1;
        pend_clear(v);
        return rc;
    }
    int r = v->mode == MODE_FAST ? mn_hnsw_build(v->index, (const int64_t *)v->pend_ids, v->pend_vecs, v->n_pend, 0, 0)
                          : mn_hnsw_insert_batch(v->index, (const int64_t *)v->pend_ids, v->pend_vecs, v->n_pend, MN_BUILD_SEQUENTIAL);
    if (r != 0) {
        sqlite3_free(v->base.zErrMsg);
        v->base.zErrMsg = sqlite3_mprintf("hnsw_index: insert failed (%s)", mn_last_error());
        pend_clear(v);
        return SQLITE_ERROR;
    }
    const double t1 = now_s();
    int rc = persist_marked(v, v->pend_ids, v->pend_vecs, v->n_pend);
    g_prof_dev_s += t1 - t0;
    g_prof_sql_s += now_s() - t1;
    g_prof_rows += v->n_pend;
    pend_clear(v);
    return rc;
}

/* load_index_from_shadow (src/hnsw_vtab.c:286-341): nodes in rowid order, edges in primary-key order */
static int load_from_shadow(VtabHnsw *v) {
    sqlite3_stmt *st = 0;
    char *sql = sqlite3_mprintf("SELECT id, vector, level, deleted FROM \"%w_nodes\"", v->name);
    int rc = sqlite3_prepare_v2(v->db, sql, -1, &st, 0);
    sqlite3_free(sql);
    if (rc != SQLITE_OK)
        return rc;
    while (sqlite3_step(st) == SQLITE_ROW) {
        const float *vec = (const float *)sqlite3_column_blob(st, 1);
        /* 1 = the reference's load loop would have gone on without this node too (node table full, src/hnsw_vtab.c:316) */
        if (!vec || sqlite3_column_bytes(st, 1) != v->dim * (int)sizeof(float) ||
            mn_hnsw_load_node(v->index, sqlite3_column_int64(st, 0), vec, sqlite3_column_int(st, 2), sqlite3_column_int(st, 3)) < 0) {
            sqlite3_finalize(st);
            return SQLITE_ERROR;
        }
    }
    sqlite3_finalize(st);
    sql = sqlite3_mprintf("SELECT source_id, target_id, level FROM \"%w_edges\"", v->name);
    rc = sqlite3_prepare_v2(v->db, sql, -1, &st, 0);
    sqlite3_free(sql);
    if (rc != SQLITE_OK)
        return rc;
    while (sqlite3_step(st) == SQLITE_ROW) {
        int64_t dst = sqlite3_column_int64(st, 1);
        int64_t src = sqlite3_column_int64(st, 0);
        int lv = sqlite3_column_int(st, 2);
        if (mn_hnsw_node_level(v->index, src) < lv) { /* :333-336; the row stays until src is rewritten whole */
            mn_hnsw_log_invalidate(v->index, &src, 1);
            continue;
        }
        if (mn_hnsw_load_neighbors(v->index, src, lv, &dst, 1) != 0) { /* never drop an edge silently */
            sqlite3_finalize(st);
            return SQLITE_ERROR;
        }
    }
    sqlite3_finalize(st);
    return SQLITE_OK;
}

static VtabHnsw *new_vtab(sqlite3 *db, const char *name, const Params *p, mn_index *ix) {
    VtabHnsw *v = (VtabHnsw *)sqlite3_malloc((int)sizeof(VtabHnsw));
    if (!v)
        return 0;
    memset(v, 0, sizeof(*v));
    v->db = db;
    v->name = sqlite3_mprintf("%s", name);
    v->index = ix;
    v->dim = p->dimensions;
    v->metric = p->metric;
    v->m = p->m;
    v->efc = p->efc;
    const char *mode = getenv("MUNINN_HNSW_MODE");
    v->mode = mode && !strcmp(mode, "fast") ? MODE_FAST : mode && !strcmp(mode, "deferred") ? MODE_DEFERRED : MODE_EXACT;
    const char *delta = getenv("MUNINN_HNSW_DELTA"); /* =0: whole-node rewrites always, as the reference does them */
    v->delta_ok = !(delta && !strcmp(delta, "0"));
    live_add(v);
    return v;
}

static int x_create(sqlite3 *db, void *aux, int argc, const char *const *argv, sqlite3_vtab **out, char **err) {
    (void)aux;
    Params p;
    int rc = parse_args(argc, argv, &p, err);
    if (rc != SQLITE_OK)
        return rc;
    rc = sqlite3_declare_vtab(db, SCHEMA);
    if (rc != SQLITE_OK)
        return rc;
    rc = make_shadow_tables(db, argv[2]);
    if (rc != SQLITE_OK) {
        *err = sqlite3_mprintf("hnsw_index: failed to create shadow tables");
        return rc;
    }
    mn_index *ix = mn_hnsw_create_on(p.dimensions, p.metric, p.m, p.efc, mn_env_device());
    if (!ix) {
        /* (the reference reports SQLITE_NOMEM here, src/hnsw_vtab.c:380-383: its only cause is malloc; here the usual cause
         * is the device — a wrong MUNINN_DEVICE, no GPU — and SQLITE_NOMEM would drop the message) */
        *err = sqlite3_mprintf("hnsw_index: failed to allocate index (%s)%s", mn_last_error(), mn_env_device_hint());
        return SQLITE_ERROR;
    }
    VtabHnsw *v = new_vtab(db, argv[2], &p, ix);
    if (!v) {
        mn_hnsw_destroy(ix);
        return SQLITE_NOMEM;
    }
    write_config(v);
    *out = &v->base;
    return SQLITE_OK;
}

static int x_connect(sqlite3 *db, void *aux, int argc, const char *const *argv, sqlite3_vtab **out, char **err) {
    (void)aux;
    (void)argc;
    Params p;
    sqlite3_int64 entry;
    int max_level;
    int rc = read_config(db, argv[2], &p, &entry, &max_level);
    if (rc != SQLITE_OK) {
        *err = sqlite3_mprintf("hnsw_index: failed to load config from shadow tables");
        return rc;
    }
    if (p.dimensions == 0) {
        *err = sqlite3_mprintf("hnsw_index: corrupted config — dimensions is 0");
        return SQLITE_ERROR;
    }
    rc = sqlite3_declare_vtab(db, SCHEMA);
    if (rc != SQLITE_OK)
        return rc;
    mn_index *ix = mn_hnsw_create_on(p.dimensions, p.metric, p.m, p.efc, mn_env_device());
    if (!ix) {
        /* (the reference reports SQLITE_NOMEM here, src/hnsw_vtab.c:380-383: its only cause is malloc; here the usual cause
         * is the device — a wrong MUNINN_DEVICE, no GPU — and SQLITE_NOMEM would drop the message) */
        *err = sqlite3_mprintf("hnsw_index: failed to allocate index (%s)%s", mn_last_error(), mn_env_device_hint());
        return SQLITE_ERROR;
    }
    VtabHnsw *v = new_vtab(db, argv[2], &p, ix);
    if (!v) {
        mn_hnsw_destroy(ix);
        return SQLITE_NOMEM;
    }
    rc = load_from_shadow(v);
    if (rc != SQLITE_OK) {
        live_remove(v);
        mn_hnsw_destroy(ix);
        sqlite3_free(v->name);
        sqlite3_free(v);
        *err = sqlite3_mprintf("hnsw_index: failed to load index from shadow tables");
        return rc;
    }
    mn_hnsw_set_entry(ix, entry, max_level);
    *out = &v->base;
    return SQLITE_OK;
}

static int x_disconnect(sqlite3_vtab *vt) {
    VtabHnsw *v = (VtabHnsw *)vt;
    if (getenv("MUNINN_PROFILE") && g_prof_rows)
        fprintf(stderr, "[muninn] %lld rows flushed: device %.3f s, shadow tables %.3f s\n", g_prof_rows, g_prof_dev_s, g_prof_sql_s);
    live_remove(v);
    pend_free(v);
    delta_release(v);
    mn_hnsw_destroy(v->index);
    sqlite3_free(v->name);
    sqlite3_free(v);
    return SQLITE_OK;
}

static int x_destroy(sqlite3_vtab *vt) { /* src/hnsw_vtab.c:471-494 */
    VtabHnsw *v = (VtabHnsw *)vt;
    delta_release(v);
    run_sql(v->db, sqlite3_mprintf("DROP TABLE IF EXISTS \"%w_config\"", v->name));
    run_sql(v->db, sqlite3_mprintf("DROP TABLE IF EXISTS \"%w_nodes\"", v->name));
    run_sql(v->db, sqlite3_mprintf("DROP INDEX IF EXISTS \"%w_edges_rev\"", v->name));
    run_sql(v->db, sqlite3_mprintf("DROP TABLE IF EXISTS \"%w_edges\"", v->name));
    return x_disconnect(vt);
}

/* src/hnsw_vtab.c:498-550 */
static int x_best_index(sqlite3_vtab *vt, sqlite3_index_info *ii) {
    (void)vt;
    int i_match = -1, i_k = -1, i_rowid = -1, i_ef = -1;
    for (int i = 0; i < ii->nConstraint; i++) {
        if (!ii->aConstraint[i].usable)
            continue;
        int col = ii->aConstraint[i].iColumn, op = ii->aConstraint[i].op;
        if (col == COL_VECTOR && op == SQLITE_INDEX_CONSTRAINT_MATCH) i_match = i;
        else if (col == COL_K && op == SQLITE_INDEX_CONSTRAINT_EQ) i_k = i;
        else if (col == COL_EF && op == SQLITE_INDEX_CONSTRAINT_EQ) i_ef = i;
        else if (col == -1 && op == SQLITE_INDEX_CONSTRAINT_EQ) i_rowid = i;
    }
    if (i_match >= 0 && i_k >= 0) {
        int arg = 1;
        ii->idxNum = PLAN_KNN;
        ii->aConstraintUsage[i_match].argvIndex = arg++;
        ii->aConstraintUsage[i_match].omit = 1;
        ii->aConstraintUsage[i_k].argvIndex = arg++;
        ii->aConstraintUsage[i_k].omit = 1;
        if (i_ef >= 0) {
            ii->aConstraintUsage[i_ef].argvIndex = arg++;
            ii->aConstraintUsage[i_ef].omit = 1;
        }
        ii->estimatedCost = 10.0;
        ii->estimatedRows = 10;
    } else if (i_rowid >= 0) {
        ii->idxNum = PLAN_POINT;
        ii->aConstraintUsage[i_rowid].argvIndex = 1;
        ii->aConstraintUsage[i_rowid].omit = 1;
        ii->estimatedCost = 1.0;
        ii->estimatedRows = 1;
    } else {
        ii->idxNum = PLAN_SCAN;
        ii->estimatedCost = 1000000.0;
        ii->estimatedRows = 1000000;
    }
    return SQLITE_OK;
}

static int x_open(sqlite3_vtab *vt, sqlite3_vtab_cursor **out) {
    (void)vt;
    CurHnsw *c = (CurHnsw *)sqlite3_malloc((int)sizeof(CurHnsw));
    if (!c)
        return SQLITE_NOMEM;
    memset(c, 0, sizeof(*c));
    c->eof = 1;
    *out = &c->base;
    return SQLITE_OK;
}

static int x_close(sqlite3_vtab_cursor *cur) {
    CurHnsw *c = (CurHnsw *)cur;
    free(c->hits);
    free(c->vecbuf);
    sqlite3_free(c);
    return SQLITE_OK;
}

/* src/hnsw_vtab.c:572-620 */
static int x_filter(sqlite3_vtab_cursor *cur, int idxNum, const char *idxStr, int argc, sqlite3_value **argv) {
    (void)idxStr;
    CurHnsw *c = (CurHnsw *)cur;
    VtabHnsw *v = (VtabHnsw *)cur->pVtab;
    free(c->hits);
    c->hits = 0;
    c->n_hits = c->pos = 0;
    c->is_point = 0;
    c->eof = 1;
    if (idxNum != PLAN_SCAN) { /* rows queued in this transaction must be visible to the read */
        int frc = flush_pending(v);
        if (frc != SQLITE_OK)
            return frc;
    }
    if (idxNum == PLAN_KNN) {
        const float *q = (const float *)sqlite3_value_blob(argv[0]);
        int qbytes = sqlite3_value_bytes(argv[0]);
        int k = sqlite3_value_int(argv[1]);
        int ef = argc >= 3 ? sqlite3_value_int(argv[2]) : k * 2;
        int want = v->dim * (int)sizeof(float);
        if (qbytes != want) {
            v->base.zErrMsg = sqlite3_mprintf("hnsw_index: expected %d-dim vector (%d bytes), got %d bytes", v->dim, want, qbytes);
            return SQLITE_ERROR;
        }
        if (k <= 0)
            return SQLITE_OK;
        c->hits = (mn_search_result *)malloc((size_t)k * sizeof(mn_search_result));
        if (!c->hits)
            return SQLITE_NOMEM;
        c->n_hits = mn_hnsw_search(v->index, q, k, ef, c->hits);
        c->eof = c->n_hits == 0;
    } else if (idxNum == PLAN_POINT) {
        c->point_id = sqlite3_value_int64(argv[0]);
        c->is_point = 1;
        c->eof = !node_is_live(v, c->point_id);
    }
    return SQLITE_OK;
}

static int x_next(sqlite3_vtab_cursor *cur) {
    CurHnsw *c = (CurHnsw *)cur;
    if (c->is_point || ++c->pos >= c->n_hits)
        c->eof = 1;
    return SQLITE_OK;
}

static int x_eof(sqlite3_vtab_cursor *cur) { return ((CurHnsw *)cur)->eof; }

static int x_rowid(sqlite3_vtab_cursor *cur, sqlite3_int64 *out) {
    CurHnsw *c = (CurHnsw *)cur;
    *out = c->is_point ? c->point_id : c->hits[c->pos].id;
    return SQLITE_OK;
}

static int x_column(sqlite3_vtab_cursor *cur, sqlite3_context *ctx, int col) { /* src/hnsw_vtab.c:648-683 */
    CurHnsw *c = (CurHnsw *)cur;
    VtabHnsw *v = (VtabHnsw *)cur->pVtab;
    sqlite3_int64 id = c->is_point ? c->point_id : c->hits[c->pos].id;
    if (col == COL_VECTOR) {
        if (!c->vecbuf)
            c->vecbuf = (float *)malloc((size_t)v->dim * sizeof(float));
        if (c->vecbuf && mn_hnsw_get_vector(v->index, id, c->vecbuf) == 0)
            sqlite3_result_blob(ctx, c->vecbuf, v->dim * (int)sizeof(float), SQLITE_TRANSIENT);
        else
            sqlite3_result_null(ctx);
    } else if (col == COL_DISTANCE) {
        sqlite3_result_double(ctx, c->is_point ? 0.0 : (double)c->hits[c->pos].distance);
    } else {
        sqlite3_result_null(ctx);
    }
    return SQLITE_OK;
}

/* src/hnsw_vtab.c:686-784 */
static int x_update(sqlite3_vtab *vt, int argc, sqlite3_value **argv, sqlite3_int64 *rowid) {
    VtabHnsw *v = (VtabHnsw *)vt;
    if (argc == 1) { /* DELETE */
        sqlite3_int64 id = sqlite3_value_int64(argv[0]);
        /* hnsw_delete edits neighbour lists that the reference leaves un-persisted (:702-706): what earlier
         * inserts marked must be written before those edits, as the reference wrote it at insert time */
        int frc = flush_pending(v);
        if (frc != SQLITE_OK)
            return frc;
        if (!node_is_live(v, id)) {
            v->base.zErrMsg = sqlite3_mprintf("hnsw_index: rowid %lld not found", (long long)id);
            return SQLITE_ERROR;
        }
        if (mn_hnsw_delete(v->index, id) != 0) {
            v->base.zErrMsg = sqlite3_mprintf("hnsw_index: delete failed (%s)", mn_last_error());
            return SQLITE_ERROR;
        }
        run_sql(v->db, sqlite3_mprintf("UPDATE \"%w_nodes\" SET deleted = 1 WHERE id = %lld", v->name, (long long)id));
        write_config(v);
        return SQLITE_OK;
    }
    if (argc > 1 && sqlite3_value_type(argv[0]) == SQLITE_NULL) { /* INSERT */
        sqlite3_int64 id;
        if (sqlite3_value_type(argv[1]) == SQLITE_NULL) {
            id = mn_hnsw_node_count(v->index) + v->n_pend + 1; /* :722-727 */
            while (node_is_live(v, id) || pset_has(v, id))
                id++;
        } else {
            id = sqlite3_value_int64(argv[1]);
        }
        if (sqlite3_value_type(argv[2]) != SQLITE_BLOB) {
            v->base.zErrMsg = sqlite3_mprintf("hnsw_index: vector must be a BLOB");
            return SQLITE_ERROR;
        }
        const float *vec = (const float *)sqlite3_value_blob(argv[2]);
        int bytes = sqlite3_value_bytes(argv[2]);
        int want = v->dim * (int)sizeof(float);
        if (bytes != want) {
            v->base.zErrMsg = sqlite3_mprintf("hnsw_index: expected %d-dim vector (%d bytes), got %d bytes", v->dim, want, bytes);
            return SQLITE_ERROR;
        }
        /* hnsw_insert refuses an id that is in the table, soft-deleted or not (src/hnsw_algo.c:522-524) */
        if (mn_hnsw_node_level(v->index, id) >= 0 || pset_has(v, id)) {
            v->base.zErrMsg = sqlite3_mprintf("hnsw_index: insert failed (duplicate rowid %lld?)", (long long)id);
            return SQLITE_ERROR;
        }
        int arc = pend_add(v, id, vec);
        if (arc != SQLITE_OK)
            return arc;
        *rowid = id;
        const int scale = v->mode == MODE_FAST ? 16 : 1;
        if (v->mode == MODE_EXACT || v->n_pend >= MN_PEND_MAX_ROWS * scale ||
            (size_t)v->n_pend * v->dim * sizeof(float) >= MN_PEND_MAX_BYTES * (size_t)scale)
            return flush_pending(v);
        return SQLITE_OK;
    }
    v->base.zErrMsg = sqlite3_mprintf("hnsw_index: UPDATE not supported, use DELETE + INSERT");
    return SQLITE_ERROR;
}

/* Transaction hooks exist only to learn when the statement/transaction ends (the reference's module has
 * none, src/hnsw_vtab.c:788-803): xSync applies and persists the queue, xRollback drops it. */
static int x_begin(sqlite3_vtab *vt) {
    sp_forget((VtabHnsw *)vt, 0);
    return SQLITE_OK;
}
static int x_sync(sqlite3_vtab *vt) {
    VtabHnsw *v = (VtabHnsw *)vt;
    if (v->n_pend == 0)
        return SQLITE_OK;
    /* the shadow-table statements below run after xUpdate has returned, so unlike the reference's (which run inside
     * it) they would show through sqlite3_last_insert_rowid(): put the statement's own value back */
    const int can_restore = sqlite3_libversion_number() >= 3018000;
    const sqlite3_int64 last = sqlite3_last_insert_rowid(v->db);
    int rc = flush_pending(v);
    if (can_restore)
        sqlite3_set_last_insert_rowid(v->db, last);
    return rc;
}
static int x_commit(sqlite3_vtab *vt) {
    sp_forget((VtabHnsw *)vt, 0);
    return SQLITE_OK;
}
static int x_rollback(sqlite3_vtab *vt) {
    pend_clear((VtabHnsw *)vt);
    sp_forget((VtabHnsw *)vt, 0);
    mn_hnsw_log_invalidate(((VtabHnsw *)vt)->index, 0, 0); /* shadow rows are gone that the index still holds */
    return SQLITE_OK;
}
/* statement and savepoint rollbacks take shadow rows away just the same (a multi-row INSERT that fails half way inside a
 * transaction): the module is version 2 only to hear about them */
static int x_savepoint(sqlite3_vtab *vt, int n) {
    VtabHnsw *v = (VtabHnsw *)vt;
    if (n < 0)
        return SQLITE_OK;
    if (n >= v->sp_cap) {
        int nc = v->sp_cap ? v->sp_cap : 8;
        while (nc <= n)
            nc *= 2;
        int *nm = (int *)realloc(v->sp_mark, (size_t)nc * sizeof(int));
        if (!nm)
            return SQLITE_NOMEM;
        for (int i = v->sp_cap; i < nc; i++)
            nm[i] = -1;
        v->sp_mark = nm;
        v->sp_cap = nc;
    }
    v->sp_mark[n] = v->n_pend;
    sp_forget(v, n + 1);
    return SQLITE_OK;
}
static int x_release(sqlite3_vtab *vt, int n) {
    sp_forget((VtabHnsw *)vt, n);
    return SQLITE_OK;
}
static int x_rollback_to(sqlite3_vtab *vt, int n) {
    VtabHnsw *v = (VtabHnsw *)vt;
    /* Rows queued since savepoint n go (deferred / fast mode; in exact mode the queue is always empty here).  A savepoint
     * this table never heard of was opened before its transaction began (sqlite3VtabBegin announces only the innermost
     * one), i.e. when the queue was empty. */
    pend_truncate(v, n >= 0 && n < v->sp_cap && v->sp_mark[n] >= 0 ? v->sp_mark[n] : 0);
    sp_forget(v, n + 1);
    mn_hnsw_log_invalidate(v->index, 0, 0);
    return SQLITE_OK;
}

static sqlite3_module hnsw_module = {
    .iVersion = 2,
    .xCreate = x_create,
    .xConnect = x_connect,
    .xBestIndex = x_best_index,
    .xDisconnect = x_disconnect,
    .xDestroy = x_destroy,
    .xOpen = x_open,
    .xClose = x_close,
    .xFilter = x_filter,
    .xNext = x_next,
    .xEof = x_eof,
    .xColumn = x_column,
    .xRowid = x_rowid,
    .xUpdate = x_update,
    .xBegin = x_begin,
    .xSync = x_sync,
    .xCommit = x_commit,
    .xRollback = x_rollback,
    .xSavepoint = x_savepoint,
    .xRelease = x_release,
    .xRollbackTo = x_rollback_to,
};

/* ───────────────────────── hnsw_search_batch: many queries, one launch ─────────────────────────
 * The reference's SQL surface answers one query per xFilter (src/hnsw_vtab.c:586-606).  This ADDITIVE
 * eponymous table-valued function exposes the batched device search without touching hnsw_index:
 *   SELECT query_idx, id, distance FROM hnsw_search_batch
 *    WHERE tbl = 'vec' AND queries = :blob AND k = 10 [AND ef_search = 128];
 * `queries` is nq * dimensions little-endian f32; rows come out grouped by query_idx, ascending distance. */
typedef struct {
    sqlite3_vtab base;
    sqlite3 *db;
} BatchVtab;
typedef struct {
    sqlite3_vtab_cursor base;
    int64_t *ids;
    float *dists;
    int *counts;
    int nq, k, qi, ri;
} BatchCur;
enum { BC_QIDX = 0, BC_ID, BC_DIST, BC_TBL, BC_QUERIES, BC_K, BC_EF };

static int b_connect(sqlite3 *db, void *aux, int argc, const char *const *argv, sqlite3_vtab **out, char **err) {
    (void)aux; (void)argc; (void)argv; (void)err;
    int rc = sqlite3_declare_vtab(db, "CREATE TABLE x(query_idx INTEGER, id INTEGER, distance REAL, tbl TEXT HIDDEN,"
                                      " queries BLOB HIDDEN, k INTEGER HIDDEN, ef_search INTEGER HIDDEN)");
    if (rc != SQLITE_OK)
        return rc;
    BatchVtab *v = (BatchVtab *)sqlite3_malloc((int)sizeof(BatchVtab));
    if (!v)
        return SQLITE_NOMEM;
    memset(v, 0, sizeof(*v));
    v->db = db;
    *out = &v->base;
    return SQLITE_OK;
}
static int b_disconnect(sqlite3_vtab *v) {
    sqlite3_free(v);
    return SQLITE_OK;
}
static int b_best_index(sqlite3_vtab *v, sqlite3_index_info *ii) {
    (void)v;
    int which[4] = {-1, -1, -1, -1};
    for (int i = 0; i < ii->nConstraint; i++) {
        if (!ii->aConstraint[i].usable || ii->aConstraint[i].op != SQLITE_INDEX_CONSTRAINT_EQ)
            continue;
        int col = ii->aConstraint[i].iColumn;
        if (col >= BC_TBL && col <= BC_EF)
            which[col - BC_TBL] = i;
    }
    int arg = 1, mask = 0;
    for (int j = 0; j < 4; j++)
        if (which[j] >= 0) {
            ii->aConstraintUsage[which[j]].argvIndex = arg++;
            ii->aConstraintUsage[which[j]].omit = 1;
            mask |= 1 << j;
        }
    ii->idxNum = mask;
    ii->estimatedCost = (mask & 0x7) == 0x7 ? 100.0 : 1e12;
    return SQLITE_OK;
}
static int b_open(sqlite3_vtab *v, sqlite3_vtab_cursor **out) {
    (void)v;
    BatchCur *c = (BatchCur *)calloc(1, sizeof(BatchCur));
    if (!c)
        return SQLITE_NOMEM;
    *out = &c->base;
    return SQLITE_OK;
}
static void b_clear(BatchCur *c) {
    free(c->ids);
    free(c->dists);
    free(c->counts);
    c->ids = 0;
    c->dists = 0;
    c->counts = 0;
    c->nq = 0;
}
static int b_close(sqlite3_vtab_cursor *cur) {
    b_clear((BatchCur *)cur);
    free(cur);
    return SQLITE_OK;
}
static void b_skip_empty(BatchCur *c) {
    while (c->qi < c->nq && c->ri >= c->counts[c->qi]) {
        c->qi++;
        c->ri = 0;
    }
}
static int b_filter(sqlite3_vtab_cursor *cur, int idxNum, const char *idxStr, int argc, sqlite3_value **argv) {
    (void)idxStr;
    BatchCur *c = (BatchCur *)cur;
    BatchVtab *bv = (BatchVtab *)cur->pVtab;
    b_clear(c);
    c->qi = c->ri = 0;
    if ((idxNum & 0x7) != 0x7 || argc < 3)
        return SQLITE_OK;
    const char *tbl = (const char *)sqlite3_value_text(argv[0]);
    const float *q = (const float *)sqlite3_value_blob(argv[1]);
    int qbytes = sqlite3_value_bytes(argv[1]);
    int k = sqlite3_value_int(argv[2]);
    int ef = (idxNum & 0x8) && argc >= 4 ? sqlite3_value_int(argv[3]) : 2 * k;
    VtabHnsw *v = tbl ? live_find(bv->db, tbl) : 0;
    if (!v && tbl) { /* not connected yet on this connection: touching the table connects it */
        char *sql = sqlite3_mprintf("SELECT rowid FROM \"%w\" WHERE rowid = -1", tbl);
        sqlite3_exec(bv->db, sql, 0, 0, 0);
        sqlite3_free(sql);
        v = live_find(bv->db, tbl);
    }
    if (!v) {
        bv->base.zErrMsg = sqlite3_mprintf("hnsw_search_batch: no hnsw_index table named '%s'", tbl ? tbl : "");
        return SQLITE_ERROR;
    }
    if (flush_pending(v) != SQLITE_OK) {
        bv->base.zErrMsg = sqlite3_mprintf("hnsw_search_batch: %s", v->base.zErrMsg ? v->base.zErrMsg : "pending inserts failed");
        return SQLITE_ERROR;
    }
    int row = v->dim * (int)sizeof(float);
    if (k <= 0 || qbytes <= 0 || qbytes % row != 0) {
        bv->base.zErrMsg = sqlite3_mprintf("hnsw_search_batch: queries must be a multiple of %d bytes (%d-dim f32), got %d", row, v->dim, qbytes);
        return SQLITE_ERROR;
    }
    c->nq = qbytes / row;
    c->k = k;
    c->ids = (int64_t *)malloc((size_t)c->nq * k * sizeof(int64_t));
    c->dists = (float *)malloc((size_t)c->nq * k * sizeof(float));
    c->counts = (int *)malloc((size_t)c->nq * sizeof(int));
    if (!c->ids || !c->dists || !c->counts)
        return SQLITE_NOMEM;
    if (mn_hnsw_search_batch(v->index, q, c->nq, k, ef, c->ids, c->dists, c->counts) != 0) {
        bv->base.zErrMsg = sqlite3_mprintf("hnsw_search_batch: %s", mn_last_error());
        return SQLITE_ERROR;
    }
    b_skip_empty(c);
    return SQLITE_OK;
}
static int b_next(sqlite3_vtab_cursor *cur) {
    BatchCur *c = (BatchCur *)cur;
    c->ri++;
    b_skip_empty(c);
    return SQLITE_OK;
}
static int b_eof(sqlite3_vtab_cursor *cur) {
    BatchCur *c = (BatchCur *)cur;
    return c->qi >= c->nq;
}
static int b_column(sqlite3_vtab_cursor *cur, sqlite3_context *ctx, int col) {
    BatchCur *c = (BatchCur *)cur;
    size_t at = (size_t)c->qi * c->k + c->ri;
    switch (col) {
    case BC_QIDX: sqlite3_result_int(ctx, c->qi); break;
    case BC_ID: sqlite3_result_int64(ctx, c->ids[at]); break;
    case BC_DIST: sqlite3_result_double(ctx, (double)c->dists[at]); break;
    default: sqlite3_result_null(ctx); break;
    }
    return SQLITE_OK;
}
static int b_rowid(sqlite3_vtab_cursor *cur, sqlite3_int64 *out) {
    BatchCur *c = (BatchCur *)cur;
    *out = (sqlite3_int64)c->qi * c->k + c->ri;
    return SQLITE_OK;
}
static sqlite3_module batch_module = {
    .iVersion = 0,
    .xCreate = 0,
    .xConnect = b_connect,
    .xBestIndex = b_best_index,
    .xDisconnect = b_disconnect,
    .xDestroy = b_disconnect,
    .xOpen = b_open,
    .xClose = b_close,
    .xFilter = b_filter,
    .xNext = b_next,
    .xEof = b_eof,
    .xColumn = b_column,
    .xRowid = b_rowid,
};

/* node2vec_train's output step (src/node2vec.c:540-583: INSERT every embedding into the output table) when that table is a
 * live hnsw_index of this connection in fast mode and still empty: the embeddings are trained, normalised and built into the
 * index without leaving HBM (mn_node2vec_train_into) — no per-row INSERT, no second upload — and the shadow tables are written
 * once from one bulk copy.  Same rows, same rowids (first-seen index + 1), same "{t}_nodes" / "{t}_edges" as the generic INSERT
 * path in this mode (MUNINN_N2V_DIRECT=0 forces that path; the tests compare the two).
 * Returns 1: done, *inserted rows; 0: not applicable, the caller INSERTs as usual; -1: error, *err (sqlite3_mprintf). */
/* A failed direct fill may have left some batches of nodes in HBM with no shadow rows behind them; the index was verified
 * empty on entry, so it is simply replaced by a fresh one — a retry (or plain INSERTs) then start from the empty table the
 * shadow tables describe.  If even that fails the old handle stays and says what it holds. */
static void n2v_fill_undo(VtabHnsw *v) {
    if (mn_hnsw_node_count(v->index) == 0)
        return;
    mn_index *fresh = mn_hnsw_create_on(v->dim, v->metric, v->m, v->efc, mn_hnsw_device(v->index));
    if (!fresh)
        return;
    mn_hnsw_destroy(v->index);
    v->index = fresh;
}

int mn_vtab_hnsw_fill_from_n2v(sqlite3 *db, const char *table, int n, const int *off, const int *adj, const mn_n2v_params *prm,
                               int *inserted, char **err) {
    const char *e = getenv("MUNINN_N2V_DIRECT");
    VtabHnsw *v = live_find(db, table);
    if (!v || v->mode != MODE_FAST || v->dim != prm->dim || v->n_pend != 0 || (e && !strcmp(e, "0")))
        return 0;
    sqlite3_stmt *st = 0;
    char *sql = sqlite3_mprintf("SELECT count(*) FROM \"%w_nodes\"", v->name);
    int rc = sqlite3_prepare_v2(db, sql, -1, &st, 0);
    sqlite3_free(sql);
    if (rc != SQLITE_OK)
        return 0;
    const int have = sqlite3_step(st) == SQLITE_ROW ? sqlite3_column_int(st, 0) : 1;
    sqlite3_finalize(st);
    if (have != 0 || mn_hnsw_node_count(v->index) != 0)
        return 0; /* rows exist already: the INSERT path has the duplicate-rowid semantics */
    float *emb = (float *)malloc((size_t)n * (size_t)prm->dim * sizeof(float));
    if (!emb) {
        *err = sqlite3_mprintf("node2vec_train: out of memory");
        return -1;
    }
    if (mn_node2vec_train_into(n, off, adj, prm, MN_N2V_BATCHED, v->index, 1, emb, 0, 0) < 0) {
        free(emb);
        *err = sqlite3_mprintf("node2vec_train: %s", mn_node2vec_last_error());
        n2v_fill_undo(v);
        return -1;
    }
    rc = SQLITE_OK;
    for (int i = 0; i < n && rc == SQLITE_OK; i++) /* (the queue doubles as the "new rows" list of persist_marked) */
        rc = pend_add(v, (sqlite3_int64)i + 1, emb + (size_t)i * prm->dim);
    free(emb);
    if (rc == SQLITE_OK)
        rc = persist_marked(v, v->pend_ids, v->pend_vecs, v->n_pend);
    pend_clear(v);
    if (rc != SQLITE_OK) {
        *err = sqlite3_mprintf("node2vec_train: writing the shadow tables of \"%s\" failed", table);
        n2v_fill_undo(v);
        return -1;
    }
    *inserted = n;
    return 1;
}

int mn_register_hnsw_module(sqlite3 *db) {
    int rc = sqlite3_create_module(db, "hnsw_index", &hnsw_module, 0); /* src/hnsw_vtab.c:805-807 */
    if (rc == SQLITE_OK)
        rc = sqlite3_create_module(db, "hnsw0", &hnsw_module, 0); /* alias named by BASELINE.json */
    if (rc == SQLITE_OK)
        rc = sqlite3_create_module(db, "hnsw_search_batch", &batch_module, 0); /* additive batch surface */
    return rc;
}
