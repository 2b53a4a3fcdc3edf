// What do s_memtime ticks mean on this chip?  One wave: Delta s_memtime and Delta s_memrealtime (100 MHz) around (a) a loop of s_nop (known shader
// cycles: s_nop 15 = 16 cycles) and (b) a chain of dependent MFMAs (32x32x16 f16: 8 passes = 32 cycles each), alone and with every CU busy
// (grid of 1024 such waves).  -> tick rate of s_memtime, shader clock idle and under MFMA load.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/clock_probe tools/clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void k_probe(int mode, int iters, unsigned long long* out, float* sink) {
    floatx16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    half8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (_Float16)(threadIdx.x * 0.001f); b[r] = (_Float16)1.0f; }
    const unsigned long long m0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    if (mode == 0) {
        for (int i = 0; i < iters; ++i) asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");      // 64 cycles per iteration (+ loop)
    } else {
        for (int i = 0; i < iters; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);                                        // dependent: 8 passes each (+ 2 for the RAW)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 4" ::: "memory");
    const unsigned long long m1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = m1 - m0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (mode == 1) sink[blockIdx.x * 64 + threadIdx.x] = acc[0] + acc[7];
}

int main() {
    unsigned long long* d; float* s;
    hipMalloc(&d, 4096 * 16); hipMalloc(&s, 4096 * 64 * 4);
    std::vector<unsigned long long> h(8192);
    for (int mode = 0; mode < 2; ++mode)
        for (int grid : {1, 1024, 4096}) {
            const int iters = 20000;
            hipLaunchKernelGGL(k_probe, dim3(grid), dim3(64), 0, 0, mode, iters, d, s);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(k_probe, dim3(grid), dim3(64), 0, 0, mode, iters, d, s);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, grid * 16, hipMemcpyDeviceToHost);
            double mt = 0, rt = 0;
            for (int i = 0; i < grid; ++i) { mt += h[2 * i]; rt += h[2 * i + 1]; }
            mt /= grid; rt /= grid;
            const double us = rt / 100.0;                       // 100 MHz
            const double cyc = mode == 0 ? 64.0 * iters : 2.0 * 32.0 * iters;      // shader cycles the loop needs at least
            printf("%-28s grid %4d: %.1f us; s_memtime ticks %.0f -> %.1f MHz; loop = %.0f shader cycles -> shader clock >= %.0f MHz; ticks per shader cycle %.3f\n",
                   mode == 0 ? "s_nop loop" : "dependent MFMA 32x32x16 f16", grid, us, mt, mt / us, cyc, cyc / us, mt / cyc);
        }
    return 0;
}
