// mn_comm.hpp — the one exchange primitive of the multi-GPU paths: an all-gather of equal-sized device
// buffers between the ranks of a node, one rank per GPU.
//   RCCL transport  ncclAllGather on the caller's HIP stream (xGMI); librccl is loaded on first use (dlopen), so a
//                   single-GPU user never pays for it and the library has no link-time dependency on it
//   host transport  the caller supplies the all-gather over HOST buffers (the world-2 gloo rehearsals, where both
//                   ranks share the one GPU of the box — which RCCL refuses — and any host that has its own fabric)
#pragma once
#include "../../include/muninn_hip.h"
#include <hip/hip_runtime.h>
#include <vector>

struct mn_comm {
    int world = 1, rank = 0, device = 0;
    void *nccl = nullptr; // ncclComm_t
    mn_host_allgather_fn host_fn = nullptr;
    void *host_user = nullptr;
    std::vector<unsigned char> h_send, h_recv;
    void *d_a2a = nullptr; // host-transport emulation of the all-to-all: everybody's send buffer, padded
    size_t a2a_bytes = 0;
    int *d_status = nullptr; // [world + 1] status words of mn_comm_agree
};

// d_recv[r * bytes .. (r+1) * bytes) = rank r's d_send, for every r; in place (d_send == d_recv + rank * bytes) allowed.
// Stream-ordered for RCCL; the host transport synchronises the stream.  0 / -1 (message via mn_comm_last_error).
int mn_comm_allgather_dev(mn_comm *c, const void *d_send, void *d_recv, size_t bytes, hipStream_t st);
// All-to-all with per-pair sizes.  cnt[s * world + p] = elements rank s sends to rank p (the whole matrix, on the HOST, the same
// on every rank: callers all-gather their rows first).  d_send holds this rank's buckets back to back in peer order; d_recv
// receives the buckets meant for this rank back to back in source-rank order.  RCCL: one group of ncclSend / ncclRecv over
// xGMI — every byte travels once, to the rank that needs it; host transport: emulated over the caller's all-gather.
int mn_comm_alltoallv_dev(mn_comm *c, const void *d_send, void *d_recv, const long long *cnt, size_t elem_bytes, hipStream_t st);
const char *mn_comm_last_error_str();
// One status word per rank, all-gathered: 0 = every rank is fine, 1 = *failed_rank (the lowest) reported a failure, -1 = the
// exchange itself failed.  A rank whose local step failed MUST still call this: its peers are in the matching collective.
int mn_comm_agree(mn_comm *c, int my_status, hipStream_t st, int *failed_rank);
