// microbenchmark (round 3): where row_exponents_kernel's 0.18 ms per optimiser step go. Includes the kernel's own source and
// launches it on an 8 x 256 network with view directions, wall_clock64() samples of thread 0 per phase and layer.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I nerf-projects_amd/csrc -DNERF_ROWEXP_STAMPS \
//       profiles/microbench/row_exponents_bench.hip -o row_exponents_bench && ./row_exponents_bench
#include "mlp_kernel_h2.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

int main() {
    using namespace nerf;
    const int D = 8, W = 256;
    EqualiseRefs r{};
    struct L { int out, in; };
    std::vector<L> lin;
    for (int i = 0; i < D; ++i) lin.push_back({W, i == 0 ? 63 : (i == 5 ? 319 : W)});
    lin.push_back({128, 283});     // views
    lin.push_back({256, 256});     // feature
    lin.push_back({1, 256});       // alpha
    lin.push_back({3, 128});       // rgb
    r.n = (int)lin.size();
    unsigned off = 0;
    for (int k = 0; k < r.n; ++k) {
        r.out[k] = lin[k].out;
        r.in[k] = lin[k].in;
        r.w_off[k] = off;
        off += lin[k].out * lin[k].in;
        r.b_off[k] = off;
        off += lin[k].out;
        r.col_src[k] = -1;
    }
    auto reads = [&](int k, int src, int col0, int n) { r.col_src[k] = src; r.hid_col0[k] = col0; r.n_hid[k] = n; };
    int n = 0;
    for (int i = 0; i < D; ++i) {
        r.scale_rows[i] = 1;
        if (i > 0) reads(i, i - 1, lin[i].in - W, W);
        r.order[n++] = i;
    }
    r.scale_rows[D + 1] = 1; reads(D + 1, D - 1, 0, W); reads(D + 2, D - 1, 0, W);
    r.scale_rows[D] = 1; reads(D, D + 1, 0, W); reads(D + 3, D, 0, 128);
    r.order[n++] = D + 1; r.order[n++] = D + 2; r.order[n++] = D; r.order[n++] = D + 3;

    std::vector<float> h(off);
    srand(3);
    for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    EqualiseBatch b{};
    b.n = 2;
    for (int i = 0; i < 2; ++i) {
        float* p; int* e;
        (void)hipMalloc((void**)&p, off * sizeof(float));
        (void)hipMemcpy(p, h.data(), off * sizeof(float), hipMemcpyHostToDevice);
        (void)hipMalloc((void**)&e, kMaxLinears * 256 * sizeof(int));
        b.params[i] = p; b.row_exp[i] = e; b.out[i] = nullptr; b.refs[i] = r;
    }
    (void)hipMalloc((void**)&b.stamps, 4096 * sizeof(unsigned long long));
    float* big; const size_t big_n = 256u << 20;      // a buffer larger than the caches, written between runs
    (void)hipMalloc((void**)&big, big_n);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipMemset(big, rep, big_n);
        (void)hipMemcpy((void*)b.params[0], h.data(), off * sizeof(float), hipMemcpyHostToDevice);
        (void)hipMemcpy((void*)b.params[1], h.data(), off * sizeof(float), hipMemcpyHostToDevice);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(row_exponents_kernel, dim3(2), dim3(kRowExpThreads), 0, 0, b);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> st(4096);
        (void)hipMemcpy(st.data(), b.stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        printf("run %d: %.1f us\n", rep, ms * 1e3);
        if (rep == 3) {
            int q = 0;
            for (int idx = 0; idx < r.n; ++idx) {
                const int k = r.order[idx];
                if (r.scale_rows[k]) {
                    printf("  linear %2d (%3d x %3d): loads+norms %5.2f  sync %5.2f  histogram %5.2f  median %5.2f us", k, r.out[k], r.in[k],
                           (st[q + 1] - st[q]) * 0.01, (st[q + 2] - st[q + 1]) * 0.01, (st[q + 3] - st[q + 2]) * 0.01, (st[q + 4] - st[q + 3]) * 0.01);
                    if (idx + 1 < r.n) printf("  write+sync %5.2f", (st[q + 5] - st[q + 4]) * 0.01);
                    printf("\n");
                    q += 5;
                } else {
                    q += 1;
                }
            }
        }
    }
    return 0;
}
