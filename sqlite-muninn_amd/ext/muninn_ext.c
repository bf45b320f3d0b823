/*
 * muninn_ext.c — SQLite loadable-extension entry point.
 *
 * Same entry symbol and contract as the reference (src/muninn.c:42-121): the file is named
 * muninn.so so SQLite derives `sqlite3_muninn_init`; subsystems are registered in a fixed order and
 * the first failure aborts with *pzErrMsg set via sqlite3_mprintf.  Registered here: the
 * hot-path surface of SURVEY §8(b) — `hnsw_index` (+ `hnsw0` alias), `node2vec_train`,
 * `graph_leiden` — and §8 f-4's `graph_components` / `graph_pagerank` / `graph_node_betweenness` / `graph_edge_betweenness`, each backed by libmuninn_hip.so
 * (include/muninn_hip.h).
 */
#include "mn_sqlite_abi.h"

const sqlite3_api_routines *mn_sqlite_api = 0;

int mn_register_hnsw_module(sqlite3 *db);
int mn_register_graph_functions(sqlite3 *db) __attribute__((weak));
int mn_register_graph_tvfs(sqlite3 *db) __attribute__((weak));
int mn_register_betweenness_tvfs(sqlite3 *db) __attribute__((weak));

#ifdef _WIN32
__declspec(dllexport)
#endif
int sqlite3_muninn_init(sqlite3 *db, char **pzErrMsg, const sqlite3_api_routines *pApi) {
    mn_sqlite_api = pApi;
    int rc = mn_register_hnsw_module(db);
    if (rc != SQLITE_OK) {
        *pzErrMsg = sqlite3_mprintf("muninn: failed to register hnsw_index module");
        return rc;
    }
    if (mn_register_graph_functions) {
        rc = mn_register_graph_functions(db);
        if (rc != SQLITE_OK) {
            *pzErrMsg = sqlite3_mprintf("muninn: failed to register graph functions");
            return rc;
        }
    }
    if (mn_register_graph_tvfs) {
        rc = mn_register_graph_tvfs(db);
        if (rc != SQLITE_OK) {
            *pzErrMsg = sqlite3_mprintf("muninn: failed to register graph TVFs");
            return rc;
        }
    }
    if (mn_register_betweenness_tvfs) {
        rc = mn_register_betweenness_tvfs(db);
        if (rc != SQLITE_OK) {
            *pzErrMsg = sqlite3_mprintf("muninn: failed to register centrality TVFs");
            return rc;
        }
    }
    return SQLITE_OK;
}
