// mn_comm.hip — communicator of the multi-GPU paths: RCCL over xGMI (one rank per GPU, processes or threads), or a
// caller-supplied host all-gather for rehearsals.  See mn_comm.hpp.
#include "mn_comm.hpp"
#include "mn_guard.hpp"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>

static thread_local std::string g_cerr;
static void cset_err(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_cerr = buf;
}
const char *mn_comm_last_error_str() { return g_cerr.c_str(); }
extern "C" const char *mn_comm_last_error(void) { return g_cerr.c_str(); }

// the five RCCL entry points used, resolved from librccl.so.1 on first use (signatures: rccl/rccl.h)
namespace {
struct RcclUniqueId { char internal[128]; };
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(RcclUniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclUniqueId, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
bool rccl_load() {
    std::call_once(g_rccl_once, [] {
        // a process that already carries RCCL (torch.distributed's backend "nccl") gets THAT copy: one RCCL per process, so the
        // host's communicators and this library's share its state; only otherwise is the library loaded here
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h)
            h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!h)
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h)
            h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h)
            return;
        Rccl r;
        r.lib = h;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(h, "ncclSend")); // (point-to-point: only the all-to-all needs them)
        r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(h, "ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        if (r.GetUniqueId && r.CommInitRank && r.AllGather && r.CommDestroy && r.GetErrorString)
            g_rccl = r;
    });
    return g_rccl.lib != nullptr;
}
} // namespace

extern "C" int mn_comm_unique_id(void *id128) try {
    if (!rccl_load()) {
        cset_err("mn_comm_unique_id: librccl.so.1 could not be loaded");
        return -1;
    }
    RcclUniqueId id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) {
        cset_err("ncclGetUniqueId: %s", g_rccl.GetErrorString(rc));
        return -1;
    }
    memcpy(id128, id.internal, MN_COMM_ID_BYTES);
    return 0;
} MN_GUARD_END(cset_err, MN_NOTHING, -1)

extern "C" mn_comm *mn_comm_init_rccl(int world, int rank, const void *id128, int device) try {
    if (world < 1 || rank < 0 || rank >= world || !id128) {
        cset_err("mn_comm_init_rccl: bad arguments");
        return nullptr;
    }
    if (!rccl_load()) {
        cset_err("mn_comm_init_rccl: librccl.so.1 could not be loaded");
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        cset_err("mn_comm_init_rccl: HIP device %d not available", device);
        return nullptr;
    }
    RcclUniqueId id;
    memcpy(id.internal, id128, MN_COMM_ID_BYTES);
    void *nc = nullptr;
    const int rc = g_rccl.CommInitRank(&nc, world, id, rank); // (collective: returns when every rank has called it)
    if (rc != 0) {
        cset_err("ncclCommInitRank(world %d, rank %d): %s", world, rank, g_rccl.GetErrorString(rc));
        return nullptr;
    }
    mn_comm *c = new mn_comm();
    c->world = world;
    c->rank = rank;
    c->device = device;
    c->nccl = nc;
    return c;
} MN_GUARD_END(cset_err, MN_NOTHING, nullptr)

extern "C" mn_comm *mn_comm_init_host(int world, int rank, mn_host_allgather_fn fn, void *user, int device) try {
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && !fn)) {
        cset_err("mn_comm_init_host: bad arguments");
        return nullptr;
    }
    mn_comm *c = new mn_comm();
    c->world = world;
    c->rank = rank;
    c->device = device;
    c->host_fn = fn;
    c->host_user = user;
    return c;
} MN_GUARD_END(cset_err, MN_NOTHING, nullptr)

extern "C" int mn_comm_world(mn_comm *c) { return c->world; }
extern "C" int mn_comm_rank(mn_comm *c) { return c->rank; }

extern "C" void mn_comm_destroy(mn_comm *c) {
    if (!c)
        return;
    if (c->d_status) {
        (void)hipSetDevice(c->device);
        (void)hipFree(c->d_status);
    }
    if (c->d_a2a) {
        (void)hipSetDevice(c->device);
        (void)hipFree(c->d_a2a);
    }
    if (c->nccl && g_rccl.lib) {
        (void)hipSetDevice(c->device);
        (void)g_rccl.CommDestroy(c->nccl);
    }
    delete c;
}

int mn_comm_allgather_dev(mn_comm *c, const void *d_send, void *d_recv, size_t bytes, hipStream_t st) {
    if (bytes == 0)
        return 0;
    if (c->world == 1 && !c->nccl) { // nothing to exchange
        if (d_send != d_recv && hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            cset_err("mn_comm: device copy failed");
            return -1;
        }
        return 0;
    }
    if (c->nccl) {
        const int rc = g_rccl.AllGather(d_send, d_recv, bytes, /* ncclChar */ 0, c->nccl, st);
        if (rc != 0) {
            cset_err("ncclAllGather(%zu bytes per rank): %s", bytes, g_rccl.GetErrorString(rc));
            return -1;
        }
        return 0;
    }
    c->h_send.resize(bytes);
    c->h_recv.resize(bytes * (size_t)c->world);
    if (hipMemcpyAsync(c->h_send.data(), d_send, bytes, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        cset_err("mn_comm: staging to the host failed");
        return -1;
    }
    if (c->host_fn(c->host_user, c->h_send.data(), c->h_recv.data(), bytes) != 0) {
        cset_err("mn_comm: the host all-gather callback failed");
        return -1;
    }
    if (hipMemcpyAsync(d_recv, c->h_recv.data(), bytes * (size_t)c->world, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        cset_err("mn_comm: staging from the host failed");
        return -1;
    }
    return 0;
}

int mn_comm_alltoallv_dev(mn_comm *c, const void *d_send, void *d_recv, const long long *cnt, size_t elem, hipStream_t st) {
    const int W = c ? c->world : 1, me = c ? c->rank : 0;
    const unsigned char *snd = static_cast<const unsigned char *>(d_send);
    unsigned char *rcv = static_cast<unsigned char *>(d_recv);
    // this rank's bucket for itself never leaves the device
    size_t soff = 0, roff = 0;
    for (int p = 0; p < me; p++)
        soff += (size_t)cnt[(size_t)me * W + p];
    for (int s = 0; s < me; s++)
        roff += (size_t)cnt[(size_t)s * W + me];
    const size_t self = (size_t)cnt[(size_t)me * W + me];
    if (self && hipMemcpyAsync(rcv + roff * elem, snd + soff * elem, self * elem, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        cset_err("mn_comm: device copy failed");
        return -1;
    }
    if (W == 1)
        return 0;
    if (c->nccl) {
        if (!g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd) {
            cset_err("mn_comm: this librccl has no ncclSend / ncclRecv");
            return -1;
        }
        int rc = g_rccl.GroupStart();
        size_t so = 0, ro = 0;
        for (int p = 0; p < W && rc == 0; p++) {
            const size_t ns = (size_t)cnt[(size_t)me * W + p], nr = (size_t)cnt[(size_t)p * W + me];
            if (p != me && ns)
                rc = g_rccl.Send(snd + so * elem, ns * elem, /* ncclChar */ 0, p, c->nccl, st);
            if (p != me && nr && rc == 0)
                rc = g_rccl.Recv(rcv + ro * elem, nr * elem, 0, p, c->nccl, st);
            so += ns;
            ro += nr;
        }
        const int rc2 = g_rccl.GroupEnd();
        if (rc != 0 || rc2 != 0) {
            cset_err("mn_comm all-to-all (ncclSend / ncclRecv): %s", g_rccl.GetErrorString(rc ? rc : rc2));
            return -1;
        }
        return 0;
    }
    // host transport: every rank's whole send buffer, padded to the longest, goes through the all-gather; the pieces meant for
    // this rank are copied out (a rehearsal path: it moves world times the bytes the RCCL path moves)
    size_t longest = 0, mine = 0;
    for (int s = 0; s < W; s++) {
        size_t tot = 0;
        for (int p = 0; p < W; p++)
            tot += (size_t)cnt[(size_t)s * W + p];
        longest = std::max(longest, tot);
        if (s == me)
            mine = tot;
    }
    if (longest == 0)
        return 0;
    const size_t slot = longest * elem;
    if (c->a2a_bytes < slot * (size_t)(W + 1)) {
        if (c->d_a2a)
            (void)hipFree(c->d_a2a);
        c->d_a2a = nullptr;
        c->a2a_bytes = 0;
        if (hipMalloc(&c->d_a2a, slot * (size_t)(W + 1)) != hipSuccess) {
            cset_err("mn_comm: all-to-all staging allocation failed");
            return -1;
        }
        c->a2a_bytes = slot * (size_t)(W + 1);
    }
    unsigned char *all = static_cast<unsigned char *>(c->d_a2a), *pad = all + slot * (size_t)W;
    if ((mine && hipMemcpyAsync(pad, snd, mine * elem, hipMemcpyDeviceToDevice, st) != hipSuccess) ||
        mn_comm_allgather_dev(c, pad, all, slot, st))
        return -1;
    size_t ro = 0;
    for (int s = 0; s < W; s++) {
        size_t so = 0;
        for (int p = 0; p < me; p++)
            so += (size_t)cnt[(size_t)s * W + p];
        const size_t n = (size_t)cnt[(size_t)s * W + me];
        if (s != me && n && hipMemcpyAsync(rcv + ro * elem, all + slot * (size_t)s + so * elem, n * elem, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            cset_err("mn_comm: device copy failed");
            return -1;
        }
        ro += n;
    }
    return 0;
}

// Every rank contributes one status word; every rank learns all of them.  A rank whose local step failed must still make
// this call — its peers are already inside (or about to enter) the matching collective and would wait for ever — and all
// ranks then fail together.  Returns 0 (all ranks fine), 1 (*failed_rank = the lowest rank that reported a failure), or -1
// when the exchange itself failed.
int mn_comm_agree(mn_comm *c, int my_status, hipStream_t st, int *failed_rank) {
    *failed_rank = -1;
    if (!c || (c->world == 1 && !c->nccl)) {
        if (my_status)
            *failed_rank = c ? c->rank : 0;
        return my_status ? 1 : 0;
    }
    if (!c->d_status && hipMalloc(&c->d_status, (size_t)(c->world + 1) * sizeof(int)) != hipSuccess) {
        cset_err("mn_comm: status buffer allocation failed");
        return -1;
    }
    std::vector<int> h((size_t)c->world, 0);
    int *mine = c->d_status + c->world; // (send and receive do not overlap: RCCL's in-place rule is about exact offsets)
    if (hipMemcpyAsync(mine, &my_status, sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { // (my_status is a stack word)
        cset_err("mn_comm: status upload failed");
        return -1;
    }
    if (mn_comm_allgather_dev(c, mine, c->d_status, sizeof(int), st))
        return -1;
    if (hipMemcpyAsync(h.data(), c->d_status, (size_t)c->world * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        cset_err("mn_comm: status download failed");
        return -1;
    }
    for (int r = 0; r < c->world; r++)
        if (h[(size_t)r]) {
            *failed_rank = r;
            return 1;
        }
    return 0;
}
