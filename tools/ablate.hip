// ablation microbenchmark (profiling tool, not product): times k_ugemm<256> on a C2-sized synthetic
// tile table with parts of its main loop compiled out (-DCCSP_ABLATE=n):
//   0 full kernel  1 no global loads in the loop  2 + no LDS stores  3 + no barriers
#include "../diffusion-ccsp_amd/csrc/ccsp_hip.hip"
#include <cstdlib>
int main(int argc, char** argv) {
    const int H = 256, R = 19840, NT = R / 64, N = 2304, C = 13;
    float *A, *W, *U, *base, *tau; int *node, *r0, *nr, *ts;
    hipMalloc(&A, (size_t)N * H * 4); hipMalloc(&W, (size_t)C * 2 * 2 * H * H * 4); hipMalloc(&U, (size_t)R * 2 * H * 4);
    hipMalloc(&base, (size_t)R * 2 * H * 4); hipMalloc(&tau, (size_t)C * 2 * H * 4);
    hipMalloc(&node, R * 4); hipMalloc(&r0, NT * 4); hipMalloc(&nr, NT * 4); hipMalloc(&ts, NT * 4);
    std::vector<float> h((size_t)C * 4 * H * H); for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(A, h.data(), (size_t)N * H * 4, hipMemcpyHostToDevice);
    hipMemcpy(base, h.data(), (size_t)R * 2 * H * 4 < h.size() * 4 ? (size_t)R * 2 * H * 4 : h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(tau, h.data(), (size_t)C * 2 * H * 4, hipMemcpyHostToDevice);
    std::vector<int> hn(R), a(NT), b(NT), c(NT);
    for (int i = 0; i < R; ++i) hn[i] = (i * 7) % N;
    for (int i = 0; i < NT; ++i) { a[i] = i * 64; b[i] = 64; c[i] = (i * 26) / NT; }
    hipMemcpy(node, hn.data(), R * 4, hipMemcpyHostToDevice); hipMemcpy(r0, a.data(), NT * 4, hipMemcpyHostToDevice);
    hipMemcpy(nr, b.data(), NT * 4, hipMemcpyHostToDevice); hipMemcpy(ts, c.data(), NT * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 50;
    for (int w = 0; w < 2; ++w) {
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i)
            hipLaunchKernelGGL((k_rowgemm<256, 512>), dim3(NT * 4 < 768 ? NT * 4 : 768), dim3(256), 0, 0, NT * 4, A, node, r0, nr, ts, W, (size_t)2 * H * H, base, tau, U);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * R * 2 * H * H;
    printf("CCSP_ABLATE=%d  k_ugemm<256> %d tiles: %.2f us/launch  %.1f TFLOP/s  (%s)\n", CCSP_ABLATE, NT, 1e3 * ms / reps,
           fl / (ms / reps * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    // bf16x3 variant on the same tile table
    unsigned short *A3, *W3;
    hipMalloc(&A3, (size_t)3 * N * H * 2); hipMalloc(&W3, (size_t)3 * C * 2 * 2 * H * H * 2);
    hipLaunchKernelGGL(k_split3, dim3(((long)N * H + 255) / 256), dim3(256), 0, 0, (long)N * H, A, A3);
    hipLaunchKernelGGL(k_split3, dim3(((long)C * 4 * H * H + 255) / 256), dim3(256), 0, 0, (long)C * 4 * H * H, W, W3);
    for (int w = 0; w < 2; ++w) {
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i)
            hipLaunchKernelGGL((k_rowgemm_bf<256, 512>), dim3(NT * 4), dim3(256), 0, 0, A3, (size_t)N * H, node, r0, nr, ts, W3,
                               (size_t)C * 4 * H * H, (size_t)2 * H * H, base, tau, U);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    hipEventElapsedTime(&ms, e0, e1);
    printf("CCSP_ABLATE=%d  k_rowgemm_bf<256,512>: %.2f us/launch  %.1f TFLOP/s-equivalent  (%s)\n", CCSP_ABLATE, 1e3 * ms / reps,
           fl / (ms / reps * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    return 0;
}
