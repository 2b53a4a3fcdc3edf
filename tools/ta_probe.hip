// What does a global_load_dwordx4 cost at ISSUE, by address pattern?  Four waves per CU (one workgroup of 256 threads per CU), every wave issues
// NL loads back to back (no use of the data until the end) from an L2-resident region, patterns:
//   0  16 rows x 64 bytes per instruction (lane: row l >> 2, piece l & 3; rows 512 bytes apart): HALF lines -- the register staging of k_rowgemm_h2
//   1   8 rows x 128 bytes per instruction (lane: row l >> 3, piece l & 7): FULL lines, same bytes per instruction
//   2  64 lanes x 16 bytes contiguous (1 KB): the best case
// Reported: shader cycles per instruction and wave (s_memtime around the issue loop, and around issue + landing).
// build: hipcc --offload-arch=gfx950 -O2 -o tools/ta_probe tools/ta_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(256) void k_ta(const char* __restrict__ buf, size_t region, int rounds, unsigned long long* out, float* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const char* base = buf + (size_t)blockIdx.x * region + (size_t)wave * (region / 4);
    size_t off;
    if (PAT == 0) off = (size_t)(lane >> 2) * 512 + (lane & 3) * 16;
    else if (PAT == 1) off = (size_t)(lane >> 3) * 512 + (lane & 7) * 16;
    else off = (size_t)lane * 16;
    f4 v[8];
    float acc = 0.f;
    unsigned long long t_issue = 0, t_all = 0;
    for (int r = 0; r < rounds; ++r) {
        // 8 instructions per round; successive instructions move on by 16 rows (PAT 0: the other plane's half line is the NEXT instruction, as in the interleaved layout)
        const char* p = base + (size_t)(r & 3) * 8192 * 4;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long m0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const char* q;
            if (PAT == 0) q = p + off + (size_t)(k >> 1) * 8192 + (k & 1) * 64;      // 16 rows per instruction, the two halves of their lines in turn
            else if (PAT == 1) q = p + off + (size_t)k * 4096;                      // 8 rows per instruction
            else q = p + off + (size_t)k * 1024;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[k]) : "v"(q) : "memory");
        }
        const unsigned long long m1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
        const unsigned long long m2 = __builtin_amdgcn_s_memtime();
        t_issue += m1 - m0; t_all += m2 - m0;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k][0] + v[k][3];
    }
    if (lane == 0) { out[2 * (blockIdx.x * 4 + wave)] = t_issue; out[2 * (blockIdx.x * 4 + wave) + 1] = t_all; }
    sink[blockIdx.x * 256 + tid] = acc;
}

int main() {
    const int grid = 256, rounds = 64;
    const size_t region = 4 * 4 * 8192 * 4;             // per workgroup: 4 waves x 4 windows x 32 KB
    char* buf; unsigned long long* d; float* s;
    hipMalloc(&buf, grid * region); hipMemset(buf, 0, grid * region);
    hipMalloc(&d, grid * 4 * 16); hipMalloc(&s, grid * 256 * 4);
    std::vector<unsigned long long> h(grid * 8);
    for (int pat = 0; pat < 3; ++pat)
        for (int rep = 0; rep < 2; ++rep) {
            if (pat == 0) hipLaunchKernelGGL(k_ta<0>, dim3(grid), dim3(256), 0, 0, buf, region, rounds, d, s);
            else if (pat == 1) hipLaunchKernelGGL(k_ta<1>, dim3(grid), dim3(256), 0, 0, buf, region, rounds, d, s);
            else hipLaunchKernelGGL(k_ta<2>, dim3(grid), dim3(256), 0, 0, buf, region, rounds, d, s);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, grid * 4 * 16, hipMemcpyDeviceToHost);
            double ti = 0, ta = 0;
            for (int i = 0; i < grid * 4; ++i) { ti += h[2 * i]; ta += h[2 * i + 1]; }
            ti /= grid * 4.0 * rounds * 8; ta /= grid * 4.0 * rounds * 8;
            printf("pattern %d (%s): issue %.1f cycles per instruction and wave, issue + landing %.1f (4 waves per CU: x 1/4 per instruction at the CU)\n", pat,
                   pat == 0 ? "16 rows x 64 B, half lines in turn" : pat == 1 ? "8 rows x 128 B, full lines" : "1 KB contiguous", ti, ta);
        }
    return 0;
}
