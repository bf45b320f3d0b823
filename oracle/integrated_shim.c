/* oracle/integrated_shim.c — TEST INFRASTRUCTURE ONLY (built by `make -C oracle integrated`, build container only).
 *
 * Evidence for SURVEY §8(b) "the other graph_* surface remains loadable": the reference's UNMODIFIED entry point
 * (src/muninn.c) and its UNMODIFIED non-hot translation units (graph_tvf, graph_centrality, graph_adjacency, graph_csr,
 * graph_load, graph_select*, id_validate — compiled where they lie) are linked into one muninn.so together with this
 * repository's re-pointed hot-path files (ext/mn_vtab_hnsw.c, ext/mn_graph_sql.c, ext/mn_graph_tvf.c over
 * libmuninn_hip.so).  The reference's hnsw_vtab.c / hnsw_algo.c / vec_math.c / priority_queue.c / node2vec.c /
 * graph_community.c are NOT in the link.  This file only supplies the three registration functions muninn.c calls for
 * them, forwarding to this repository's registrations.  Nothing built here travels to the GPU box or ships. */
#include "sqlite3ext.h"
extern const sqlite3_api_routines *sqlite3_api; /* defined by SQLITE_EXTENSION_INIT1 in the reference's muninn.c */
const sqlite3_api_routines *mn_sqlite_api = 0; /* what ext/mn_sqlite_abi.h routes every SQLite call through */

int mn_register_hnsw_module(sqlite3 *db);
int mn_register_node2vec(sqlite3 *db);
int mn_register_leiden(sqlite3 *db);
int mn_register_graph_tvfs(sqlite3 *db);
int mn_register_betweenness_tvfs(sqlite3 *db);

int hnsw_register_module(sqlite3 *db) { /* src/hnsw_vtab.h */
    mn_sqlite_api = sqlite3_api;
    return mn_register_hnsw_module(db);
}
int community_register_tvfs(sqlite3 *db) { /* src/graph_community.h; runs after the reference's graph_register_tvfs */
    int rc = mn_register_leiden(db);
    if (rc == 0)
        rc = mn_register_graph_tvfs(db); /* graph_components / graph_pagerank: the device versions replace the reference's */
    if (rc == 0)
        rc = mn_register_betweenness_tvfs(db); /* likewise graph_node_betweenness / graph_edge_betweenness */
    return rc;
}
int node2vec_register_functions(sqlite3 *db) { /* src/node2vec.h */
    return mn_register_node2vec(db);
}
