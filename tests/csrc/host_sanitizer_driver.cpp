// Driver for a HOST-ONLY AddressSanitizer + UBSan build of the library's C ABI (tests/test_oracle_sanitizers.py): every source is compiled
// with `hipcc --cuda-host-only -fsanitize=address,undefined` -- no device code, no GPU needed -- and the entry points that answer from the
// descriptor alone (plan construction, tile choice, workspace sizes, kernel-coverage queries, argument validation) are swept over a few
// thousand convolution problems, valid and invalid. This is the part of the product that runs on the host: the tap / class / tile tables of
// build_plan, the magic multipliers, the split-K and weight-gradient workspace arithmetic. (GPU ASAN / XNACK are not available on the pool.)
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include "../../include/nndet_amd.h"

static thread_local uint32_t rng = 2463534242u;
static uint32_t rnd() { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return rng; }
static int pick(const int* v, int n) { return v[rnd() % n]; }

static int sweep() {
    rng = 2463534242u;
    const int chans[] = {1, 2, 3, 27, 32, 33, 48, 64, 96, 128, 162, 256, 320};
    const int dts[] = {NNDET_F32, NNDET_BF16, NNDET_F16, 77};
    uint64_t h = 0;
    long n_ok = 0;
    for (int it = 0; it < 6000; ++it) {
        NndetConv c;
        memset(&c, 0, sizeof(c));
        c.dtype = pick(dts, it % 50 == 0 ? 4 : 3);
        c.transposed = (rnd() % 6) == 0;
        c.batch = (int)(rnd() % 5) + ((rnd() % 40) == 0 ? -1 : 0);
        c.cin = pick(chans, 13); c.cout = pick(chans, 13);
        c.cin_p = c.cin == 1 && (rnd() & 1) ? 1 : (c.cin + 31) / 32 * 32;
        c.cout_p = (c.cout + 31) / 32 * 32;
        if ((rnd() % 60) == 0) c.cout_p += 8;                       // invalid padding
        const int ks = c.transposed ? 2 : ((rnd() % 4) == 0 ? 1 : 3);
        for (int i = 0; i < 3; ++i) {
            c.k[i] = ks; c.s[i] = c.transposed ? 2 : ((rnd() % 3) == 0 ? 2 : 1); c.p[i] = c.transposed ? 0 : (ks == 3 ? 1 : 0);
            if ((rnd() % 80) == 0) c.k[i] = 5;
            if (c.transposed && (rnd() % 7) == 0) { c.k[i] = 1; c.s[i] = 1; }
        }
        const int dims[3] = {(int)(rnd() % 170) + 1, (int)(rnd() % 170) + 1, (int)(rnd() % 100) + 1};
        c.in_d = dims[0]; c.in_h = dims[1]; c.in_w = dims[2];
        const int in[3] = {c.in_d, c.in_h, c.in_w};
        int out[3];
        for (int i = 0; i < 3; ++i) out[i] = c.transposed ? in[i] * c.s[i] : (in[i] + 2 * c.p[i] - c.k[i]) / c.s[i] + 1;
        if ((rnd() % 50) == 0) out[rnd() % 3] += 1;                 // inconsistent shapes
        c.out_d = out[0]; c.out_h = out[1]; c.out_w = out[2];
        float dummy_ss[4] = {1.f, 0.f, 1.f, 0.f};
        if ((rnd() % 9) == 0) c.in_affine = dummy_ss;                // (only its presence matters to the queries)
        h = h * 31 + nndet_packed_weight_elems(&c, 0) + 3 * nndet_packed_weight_elems(&c, 1);
        h = h * 31 + nndet_conv3d_splitk_workspace_bytes(&c, 0) + 5 * nndet_conv3d_splitk_workspace_bytes(&c, 1) + 7 * nndet_conv3d_splitk_workspace_bytes(&c, 2);
        h = h * 31 + nndet_conv3d_wgrad_workspace_bytes(&c);
        h = h * 31 + (uint64_t)nndet_conv3d_dgrad_fuses_bias(&c) + 2 * (uint64_t)nndet_conv3d_dgrad_fuses_norm_reduce(&c) +
            4 * (uint64_t)nndet_conv3d_dgrad_normred_supported(&c) + 8 * (uint64_t)nndet_conv3d_forward_norm_input_fused(&c) +
            16 * (uint64_t)nndet_stem_block_supported(&c);
        h = h * 31 + nndet_stem_block_backward_workspace_bytes(&c);
        // compute entry points with missing buffers: must be refused by the argument checks, before anything is launched
        if (nndet_conv3d_forward(&c, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == 0) { fprintf(stderr, "NULL buffers accepted\n"); return 2; }
        if (nndet_conv3d_backward_data(&c, nullptr, nullptr, nullptr, nullptr) == 0) { fprintf(stderr, "NULL buffers accepted\n"); return 2; }
        if (nndet_conv3d_backward_weight(&c, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr) == 0) { fprintf(stderr, "NULL buffers accepted\n"); return 2; }
        ++n_ok;
    }
    // ragged batches: item tables at the limits
    NndetItems items;
    memset(&items, 0, sizeof(items));
    items.n_items = NNDET_MAX_ITEMS;
    int64_t row = 0;
    for (int i = 0; i < NNDET_MAX_ITEMS; ++i) {
        items.dims[i][0] = (int)(rnd() % 40) + 1; items.dims[i][1] = (int)(rnd() % 40) + 1; items.dims[i][2] = (int)(rnd() % 24) + 1;
        items.row_off[i] = row;
        row += (int64_t)items.dims[i][0] * items.dims[i][1] * items.dims[i][2];
    }
    NndetConv c;
    memset(&c, 0, sizeof(c));
    c.dtype = NNDET_BF16; c.batch = 1; c.cin = c.cout = c.cin_p = c.cout_p = 128;
    for (int i = 0; i < 3; ++i) { c.k[i] = 3; c.s[i] = 1; c.p[i] = 1; }
    c.in_d = c.out_d = 8; c.in_h = c.out_h = 8; c.in_w = c.out_w = 8;
    if (nndet_conv3d_forward_items(&c, &items, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == 0) { fprintf(stderr, "NULL buffers accepted (items)\n"); return 2; }
    items.n_items = NNDET_MAX_ITEMS + 1;
    if (nndet_conv3d_forward_items(&c, &items, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == 0) { fprintf(stderr, "too many items accepted\n"); return 2; }
    h = h * 31 + nndet_hnm_sample_workspace_bytes(32, 0.33, 1, 20.0) + (uint64_t)nndet_hnm_neg_capacity(32, 0.33, 1);
    // the box-op workspace queries over the sizes of BASELINE.json configs[4] and the degenerate ones
    const int64_t ns[] = {0, 1, 63, 64, 65, 1000, 10000, 100000, 1186650, 4746600};
    for (int64_t n : ns) {
        h = h * 31 + nndet_nms3d_workspace_bytes(n) + 3 * nndet_wbc3d_workspace_bytes(n);
        for (int G : {0, 1, 3, 2000})
            for (int L : {1, 4, 5}) h = h * 31 + nndet_atss3d_workspace_bytes(G, n, L, 4);
        for (int B : {1, 4})
            for (int C : {1, 3}) h = h * 31 + nndet_postprocess3d_workspace_bytes(B, n, C, 10000);
    }
    for (int pc : {0, 1, 32, 512})
        for (double ratio : {0.0, 0.33, 3.0})
            for (double pool : {1.0, 20.0}) h = h * 31 + nndet_hnm_sample_workspace_bytes(pc, ratio, 1, pool) + (uint64_t)nndet_hnm_neg_capacity(pc, ratio, 1);
    printf("ok %ld problems, checksum %llu\n", n_ok, (unsigned long long)h);
    return 0;
}

#ifdef SWEEP_THREADS      // ThreadSanitizer build: the same sweep from several threads at once (the library's host side keeps per-process caches:
#include <thread>         // environment switches read once, per-device attribute flags, plan tables on the stack)
#include <vector>
int main() {
    std::vector<std::thread> th;
    int rc[SWEEP_THREADS] = {0};
    for (int t = 0; t < SWEEP_THREADS; ++t) th.emplace_back([t, &rc] { rc[t] = sweep(); });
    for (auto& x : th) x.join();
    for (int t = 0; t < SWEEP_THREADS; ++t) if (rc[t]) return rc[t];
    return 0;
}
#else
int main() { return sweep(); }
#endif
