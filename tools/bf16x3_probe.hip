// numeric probe (tool, not product): is a 3-way bf16 split with 6 cross products on the bf16 matrix
// cores (fp32 accumulate) as accurate as the fp32 MFMA chain for this path's dot products (K = 256)?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rn(float x) {   // round-to-nearest-even bf16 bits
    unsigned int u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }

// C[32x32] = A[32xK] * B[32xK]^T, one wave.  mode 0: fp32 MFMA; mode 1: bf16x3, 6 products
__global__ void probe(const float* A, const float* B, float* C, int K, int mode) {
    const int lane = threadIdx.x;
    floatx16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2) {
            const float a = A[(lane & 31) * K + k + (lane >> 5)], b = B[(lane & 31) * K + k + (lane >> 5)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 a[3], b[3];
            for (int e = 0; e < 8; ++e) {
                const float av = A[(lane & 31) * K + k + 8 * (lane >> 5) + e], bv = B[(lane & 31) * K + k + 8 * (lane >> 5) + e];
                float r = av;
                for (int p = 0; p < 3; ++p) { unsigned short h = bf16_rn(r); r -= bf16_f(h); a[p][e] = __builtin_bit_cast(__bf16, h); }
                r = bv;
                for (int p = 0; p < 3; ++p) { unsigned short h = bf16_rn(r); r -= bf16_f(h); b[p][e] = __builtin_bit_cast(__bf16, h); }
            }
            // smallest terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
}

int main() {
    const int K = 256;
    std::vector<float> A(32 * K), B(32 * K), C(32 * 32);
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
    for (int trial = 0; trial < 3; ++trial) {
        const float sa = trial == 0 ? 1.f : (trial == 1 ? 1e8f : 1e-6f);   // O(1) data, huge transients, tiny values
        for (auto& v : A) v = sa * ((float)rand() / RAND_MAX * 2 - 1) * (rand() % 7 == 0 ? 30.f : 1.f);
        for (auto& v : B) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.1f;
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 2; ++mode) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, mode);
            hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            double emax = 0, scale = 0;
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
                double ref = 0, mag = 0;
                for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[j * K + k]; mag += fabs((double)A[i * K + k] * B[j * K + k]); }
                emax = fmax(emax, fabs(C[i * 32 + j] - ref) / mag); scale = fmax(scale, mag);
            }
            printf("scale %.0e  %-22s max |err| / sum|a b| = %.3e\n", sa, mode ? "bf16x3 (6 products)" : "fp32 MFMA", emax);
        }
    }
    return 0;
}
