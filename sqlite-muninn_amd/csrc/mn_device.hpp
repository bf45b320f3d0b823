// mn_device.hpp — device-side views shared by the kernels and the host shim.  gfx950 only.
//
// HBM layout of one index.  Nodes are addressed by SLOT (insertion order,
// int32); int64 rowids only appear at the API boundary (ids[slot]).
//   vectors  [cap][ld]   f32, ld = round_up(dim,4), pad = 0          (src/hnsw_algo.h:19 HnswNode.vector)
//   norms    [cap]       f32 |v|² in the index's summation order (cosine only)
//   links0   [cap][W0]   int32 neighbour slots at layer 0, list order preserved, -1 padded; W0 ≥ M0 = 2M
//   links_up [rows][WU]  int32, layers ≥ 1; node's layer l lives in row up_off[slot] + (l-1); WU ≥ MU = M
//   W0 / WU are row STRIDES; M0 / MU are the reference's M_max0 / M (src/hnsw_algo.c:188: the length a list is pruned
//   back to).  They are equal until a delete's reconnection (:775-782) or a loaded database needs a longer list than
//   M_max — the reference grows lists without bound there — at which point the host re-strides the table.
//   up_off   [cap]       int32 first pool row of the node, -1 when level == 0
//   levels   [cap] int8, deleted [cap] u8, ids [cap] int64, dirty [cap] u8 (nodes to re-persist)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MN_WAVE 64
constexpr int MN_ORDER_SSE_V = 0;  // muninn_hip.h mn_order
constexpr int MN_ORDER_WAVE_V = 1;

struct MnDevIndex {
    const float *vectors;
    const float *norms;
    int *links0;
    int *links_up;
    const int *up_off;
    const signed char *levels;
    const unsigned char *deleted;
    unsigned char *dirty; // [cap] set by the insert kernels on every node whose rows they (re)wrote: the persist set
    const long long *ids;
    int dim, ld, metric, order;
    int W0, WU; // row strides (capacity of a list)
    int M0, MU; // M_max at layer 0 / above: inserts select at most this many and prune over-full lists back to it
    int WX;     // max(W0, WU): stride of per-row scratch that serves both kinds of row
    int n_slots;
    int n_pool_rows;
    int has_deleted; // 0: no soft-deleted node exists — the per-candidate deleted[] gather of the searches is skipped
};

// One launch of the beam-search kernel (search or build flavour).
struct MnSearchArgs {
    // queries: either dense host-uploaded vectors [nq][dim] or rows of the index (build)
    const float *queries;
    const int *query_slots; // build: slot of each batch node
    long long nq;
    int k;  // search: results wanted per query
    int ef; // beam width
    int entry_slot, max_level;
    // search outputs
    long long *out_ids; // [nq][k]
    float *out_dists;   // [nq][k]
    int *out_counts;    // [nq]
    // build outputs: per (query, level) the first min(found, M_max) results
    int *sel;          // [nq][nlev][M0]
    int *nsel;         // [nq][nlev]
    int nlev;          // max_level + 1 at batch start
    const int *up_bm_index; // build: per query, index of its upper-layer bitmap block or -1
    // workspace
    unsigned *bitmap0;       // [nq][bm0_words]
    long long bm0_words;
    unsigned *bitmap_up;     // [n_upper_q][max_level][bmu_words]
    long long bmu_words;
    uint2 *cand_ovf;         // [nq][cand_gcap]
    int cand_gcap;
    uint2 *res_ovf;          // [nq][res_gcap]
    int res_gcap;
    unsigned long long *counters; // [0] n_dist [1] n_expanded [2] overflowed queries
    unsigned long long *q_counters; // or, when not null: [nq][4] the same per query, plain stores (a few queries answered into the
                                    // index's pinned host block: no counter memset before the launch, no copy after it)
    int use_tile;                 // SSE order: stage candidate rows through the LDS tile (coalesced loads)
    int lds_bitmap;               // k_beam_coop, search: the layer-0 visited bitmap lives in LDS (small indexes: one query's
                                  // bitmap fits, and the visited probe stops being a global-memory round trip per expansion)
    // build, speculative exact mode: per query the link rows its search read (mn_beam.hpp log_row_read)
    int *readlog; // [nq][readcap] or null
    int readcap;
    int *nread;   // [nq] rows read (may exceed readcap)
    // k_beam_coop, SSE order: every wavefront of the group has an LDS tile of mn_lat_tile_floats(ld, lat_tile_rows) floats at byte offset
    // lat_tile_off for its share of a distance request (sse_rows_lat_tiled, mn_dist.hpp); 0 rows = none
    int lat_tile_rows;
    unsigned lat_tile_off;
    int no_spec_rows; // k_beam_coop: 1 = request a neighbour's row only after the visited probe has answered (MN_SPEC_ROWS=0, A/B runs)
};
// dynamic LDS a workgroup of this process may ask for: 64 KB, or what the device grants on request (mn_kernels.hip)
size_t mn_lds_optin_limit();
bool mn_lds_grant(const void *kernel, size_t bytes); // asks once per kernel and size; false = stay within 64 KB

// LDS budget per wavefront (items are 8 B: f32 distance bits, int32 slot)
#ifndef MN_CAND_LDS
#define MN_CAND_LDS 256
#endif
#ifndef MN_RES_LDS
#define MN_RES_LDS 256
#endif

// host-callable launchers (mn_kernels.hip)
// force the load of each translation unit's code object (see the definitions)
void mn_module_touch_kernels();
void mn_module_touch_seq();
void mn_module_touch_spec();
void mn_module_touch_build();
void mn_launch_norms(const MnDevIndex &ix, int first_slot, int n, float *norms_out, hipStream_t st);
void mn_launch_dist_batch(int metric, int order, const float *d_query, const float *d_rows, long long n, int dim, int ld,
                          float *d_out, hipStream_t st);
size_t mn_search_lds_bytes(int ld, bool tile);
int mn_launch_search(const MnDevIndex &ix, const MnSearchArgs &a, bool build, hipStream_t st);
void mn_launch_bruteforce(const MnDevIndex &ix, const float *d_queries, long long nq, int k, long long *d_out_ids,
                          float *d_scratch, hipStream_t st);

// MFMA brute force (mn_brute.hip): k <= 16; scratch = one allocation of mn_brute_mfma_scratch_bytes()
size_t mn_brute_mfma_scratch_bytes(const MnDevIndex &ix, long long nq, int k, int *n_chunks_out, int *rows_per_chunk_out);
int mn_launch_bruteforce_mfma(const MnDevIndex &ix, const float *d_queries, long long nq, int k, long long *d_out_ids,
                              void *scratch, hipStream_t st);

// sharded index (config 3): per-shard top-k lists gathered as [world][nq][k] → global top-k per query in the total order
// (distance, shard rank, position)  (mn_kernels.hip)
// synchronises the index's stream; *n_overflow = queries of its last search that exceeded their heap workspace.  0 / -1
int mn_index_search_overflow(struct mn_index *x, long long *n_overflow);
void mn_launch_merge_topk(const long long *g_ids, const float *g_dists, const int *g_counts, int world, long long nq, int k,
                          long long *out_ids, float *out_dists, int *out_counts, hipStream_t st);

// link kernels for the batched build (mn_build.hip)
struct MnLinkArgs {
    int level, M_max;
    int nq, nlev;
    const int *query_slots; // [nq] slot of batch node j
    const int *sel;         // [nq][nlev][M0]
    const int *nsel;        // [nq][nlev]
    // scratch (all device)
    int *t_target;  // [max_tuples]
    int *t_src;     // [max_tuples] batch index j
    int *counters;  // [0] n_tuples [1] n_touched [2] cursor   (zeroed per level)
    int *count;     // [n_slots] zero on entry, zero again on exit
    int *fill;      // [n_slots]
    int *binoff;    // [n_slots]
    int *touched;   // [max_tuples]
    int *bins;      // [max_tuples]
    int *newrows;   // [max_tuples][WX]
    // jointly built graph (mn_hnsw_build_shared), link half divided over the ranks: rank r replays only the targets t with
    // t % world == r and writes each finished row as a record {t, row[W]} (rec_count = records written); the records of all
    // ranks are all-gathered and committed by every replica
    int world, rank;
    int *rec_count; // [1]
    int *records;   // [this rank's targets][1 + WX]
    int *cls_count; // [world] touched targets per residue class (the same on every rank: every rank runs the forward half)
};
void mn_launch_link(const MnDevIndex &ix, const MnLinkArgs &a, int max_tuples, hipStream_t st);
// the same in two halves around the exchange of the divided link step: forward + bins + this rank's share of the replay, then —
// once the records of all ranks are in `all_records` ([world][seg][1 + WX], cls_count valid ones per segment) — the commit
void mn_launch_link_first(const MnDevIndex &ix, const MnLinkArgs &a, int max_tuples, hipStream_t st);
void mn_launch_link_commit_records(const MnDevIndex &ix, const MnLinkArgs &a, int max_tuples, const int *all_records, int seg,
                                   hipStream_t st);

// exact sequential inserts (mn_seq.hip); LDS of the one workgroup (must stay within MN_LDS_LIMIT)
#define MN_LDS_LIMIT (64 * 1024)
size_t mn_insert_seq_lds_bytes(const MnDevIndex &ix);
void mn_launch_insert_seq(const MnDevIndex &ix, const int *d_slots, int n, int ef, int *d_state, unsigned *bitmap0,
                          long long bm0_words, unsigned *bitmap_up, long long bmu_words, uint2 *cand_ovf, int cand_gcap,
                          uint2 *res_ovf, int res_gcap, unsigned long long *counters, hipStream_t st, int *chlog = nullptr,
                          int chcap = 0); // chlog (n == 1 only): the insert's edge changes, mn_seq.hip MnSeqArgs
#define MN_CHLOG_CAP 1024 // entries of one insert's change log (more: the caller falls back to the persist set)
#define MN_CHLOG_INTS 5

// speculative exact inserts (mn_spec.hip): commit a window of searched inserts in order, stop at the first stale one
#define MN_RLOG_INTS 5      // ints per entry of a search's read log (mn_beam.hpp log_row_read)
#define MN_SPEC_SAVE_CAP 8192 // rows a speculative window may rewrite with their old lists kept (more: those rows invalidate as before)
void mn_launch_spec_commit(const MnDevIndex &ix, const int *d_slots, int W, int nlev, const int *sel, const int *nsel,
                           const int *readlog, int readcap, const int *nread, int *stamp0, int *stampU, int *sidx0, int *sidxU,
                           int *saved_rows, int *pre_act, int *pre_cnt, int *pre_row, int *why, int epoch, int *d_ncommit,
                           hipStream_t st);

// rows[r] = (slot, level): writes the row's neighbour slots and dist(slot, neighbour) (mn_kernels.hip)
void mn_launch_edge_rows(const MnDevIndex &ix, const int *d_row_slot, const int *d_row_level, int n_rows, int *d_out_nbr,
                         float *d_out_dist, hipStream_t st);
