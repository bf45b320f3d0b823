// tools/n2v_bench.cpp — BASELINE config 4 (Node2Vec, MN_N2V_BATCHED) driven straight through the C-ABI, with nothing else in
// the process: what `rocprofv3 --pmc FETCH_SIZE -- ./n2v_bench graph.csr` can profile (under the Python driver rocprofv3's
// counter collection dies in its own launch interception).  Plain C++: links libmuninn_hip.so only.
//   g++ -O2 -o /tmp/n2v_bench tools/n2v_bench.cpp -Iinclude -Lsqlite-muninn_amd -lmuninn_hip -Wl,-rpath,$PWD/sqlite-muninn_amd
// graph.csr (written by `bench_graph.py --workload node2vec --dump-csr PATH`): int32 n, int64 nadj, int32 off[n+1], int32 adj[nadj]
#include "../include/muninn_hip.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: n2v_bench graph.csr [runs] [batch_walks] [seq]\n");
        return 2;
    }
    FILE *f = fopen(argv[1], "rb");
    int n = 0;
    long long nadj = 0;
    if (!f || fread(&n, 4, 1, f) != 1 || fread(&nadj, 8, 1, f) != 1) {
        fprintf(stderr, "n2v_bench: cannot read %s\n", argv[1]);
        return 2;
    }
    std::vector<int> off((size_t)n + 1), adj((size_t)nadj);
    if (fread(off.data(), 4, off.size(), f) != off.size() || fread(adj.data(), 4, adj.size(), f) != adj.size()) {
        fprintf(stderr, "n2v_bench: short file\n");
        return 2;
    }
    fclose(f);
    const int runs = argc > 2 ? atoi(argv[2]) : 1;
    mn_n2v_params prm = {};
    prm.dim = 128; // config 4: p = q = 1, 10 walks x 80, window 5, neg 5, lr 0.025, 1 epoch
    prm.p = prm.q = 1.0;
    prm.num_walks = 10;
    prm.walk_length = 80;
    prm.window = 5;
    prm.neg_samples = 5;
    prm.learning_rate = 0.025;
    prm.epochs = 1;
    prm.batch_walks = argc > 3 ? atoi(argv[3]) : 0;
    std::vector<float> out((size_t)n * prm.dim);
    for (int r = 0; r < runs; r++) {
        mn_n2v_stats st = {};
        const auto t0 = std::chrono::steady_clock::now();
        if (mn_node2vec_train(n, off.data(), adj.data(), &prm, argc > 4 ? MN_N2V_SEQUENTIAL : MN_N2V_BATCHED, 0, out.data(), &st) < 0) {
            fprintf(stderr, "n2v_bench: %s\n", mn_node2vec_last_error());
            return 1;
        }
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("{\"run\": %d, \"nodes\": %d, \"adjacency_entries\": %lld, \"pairs\": %lld, \"device_ms\": %.3f, \"wall_s\": %.3f, "
               "\"pairs_per_s\": %.0f}\n", r, n, nadj, (long long)st.pairs, st.device_ms, wall, st.pairs / wall);
        fflush(stdout);
    }
    return 0;
}
