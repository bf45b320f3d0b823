// tools/n2v_kernels.hip — a few batches of config 4's two halves through the session API of the C-ABI, so that each half can
// be put under `rocprofv3 --pmc` on its own (the whole pipeline crashes or hangs rocprofv3's counter collection on this image,
// DESIGN.md §7.2).   usage: n2v_kernels graph.csr samples|both [batches]
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/n2v_kernels tools/n2v_kernels.hip -Iinclude -Lsqlite-muninn_amd -lmuninn_hip -Wl,-rpath,$PWD/sqlite-muninn_amd
#include "../include/muninn_hip.h"
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 3)
        return 2;
    FILE *f = fopen(argv[1], "rb");
    int n = 0;
    long long nadj = 0;
    if (!f || fread(&n, 4, 1, f) != 1 || fread(&nadj, 8, 1, f) != 1)
        return 2;
    std::vector<int> off((size_t)n + 1), adj((size_t)nadj);
    if (fread(off.data(), 4, off.size(), f) != off.size() || fread(adj.data(), 4, adj.size(), f) != adj.size())
        return 2;
    fclose(f);
    const bool both = strcmp(argv[2], "both") == 0;
    const int batches = argc > 3 ? atoi(argv[3]) : 4;
    mn_n2v_params prm = {};
    prm.dim = 128;
    prm.p = prm.q = 1.0;
    prm.num_walks = 10;
    prm.walk_length = 80;
    prm.window = 5;
    prm.neg_samples = 5;
    prm.learning_rate = 0.025;
    prm.epochs = 1;
    mn_n2v_session *S = mn_n2v_begin(n, off.data(), adj.data(), &prm, 0);
    if (!S) {
        fprintf(stderr, "%s\n", mn_node2vec_last_error());
        return 1;
    }
    const int B = mn_n2v_batch_walks(S), cap = mn_n2v_sample_slots(S), pcap = mn_n2v_position_slots(S);
    int *c, *t, *pc;
    float *e, *pn;
    if (hipMalloc(&c, (size_t)B * cap * 4) || hipMalloc(&t, (size_t)B * cap * 4) || hipMalloc(&e, (size_t)B * cap * 4) ||
        hipMalloc(&pc, (size_t)B * pcap * 4) || hipMalloc(&pn, (size_t)B * pcap * prm.dim * 4))
        return 1;
    long long pairs = 0;
    for (int b = 0; b < batches && (long long)b * B < n; b++) {
        const int lo = b * B, hi = lo + B < n ? lo + B : n;
        if (mn_n2v_samples(S, 0, 0, lo, hi, c, t, e, pc, pn) || mn_n2v_sync(S))
            return 1;
        if (both && (mn_n2v_apply(S, c, t, e, (long long)(hi - lo) * cap, pc, pn, (long long)(hi - lo) * pcap) || mn_n2v_sync(S)))
            return 1;
    }
    (void)pairs;
    printf("{\"batches\": %d, \"walks_per_batch\": %d, \"sample_slots_per_walk\": %d, \"mode\": \"%s\"}\n", batches, B, cap, argv[2]);
    mn_n2v_end(S);
    return 0;
}
