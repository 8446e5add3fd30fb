// Host-side driver of the C-ABI for the AddressSanitizer run (tests/test_host_asan_cpu.py): reads a manifest (the 14 dims, the
// number of tensors, then "name numel" per line), allocates every tensor with exactly its element count, packs the blob, checks
// the error paths and the workspace planners.  No GPU call.
#include "genvox_amd.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "r");
    if (!f) return 2;
    gvx_dims d;
    int* di = reinterpret_cast<int*>(&d);
    for (int i = 0; i < 14; ++i) if (fscanf(f, "%d", &di[i]) != 1) return 2;
    int n = 0;
    if (fscanf(f, "%d", &n) != 1) return 2;
    std::vector<std::string> names(n);
    std::vector<std::vector<float>> bufs(n);
    std::vector<gvx_weight_desc> table(n);
    for (int i = 0; i < n; ++i) {
        char nm[256]; long long numel;
        if (fscanf(f, "%255s %lld", nm, &numel) != 2) return 2;
        names[i] = nm;
        bufs[i].resize(numel);
        for (long long k = 0; k < numel; ++k) bufs[i][k] = 0.001f * (float)((k * 7 + i) % 1000) + 0.5f;
    }
    fclose(f);
    for (int i = 0; i < n; ++i) table[i] = gvx_weight_desc{names[i].c_str(), bufs[i].data(), (int64_t)bufs[i].size()};
    gvx_model* m = nullptr;
    if (gvx_model_create(&d, &m) != GVX_OK) { printf("create: %s\n", gvx_last_error()); return 1; }
    std::vector<char> blob(gvx_model_blob_bytes(m));
    if (gvx_model_pack_weights(m, table.data(), n, blob.data()) != GVX_OK) { printf("pack: %s\n", gvx_last_error()); return 1; }
    // error paths: a missing tensor, a tensor of the wrong size
    if (gvx_model_pack_weights(m, table.data(), n - 1, blob.data()) != GVX_ERR_MISSING_WEIGHT) return 1;
    table[0].numel -= 1;
    if (gvx_model_pack_weights(m, table.data(), n, blob.data()) != GVX_ERR_SHAPE) return 1;
    size_t ws = 0;
    for (int B : {1, 7, 32, 64}) for (int L : {1, 33, 128}) for (int T : {1, 9, 800}) ws += gvx_workspace_bytes(m, B, L, T) + gvx_postnet_workspace_bytes(m, B, T);
    printf("ok blob %zu ws-sum %zu\n", blob.size(), ws);
    gvx_model_destroy(m);
    return 0;
}
