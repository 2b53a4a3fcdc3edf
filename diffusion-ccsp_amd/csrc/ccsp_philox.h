// Device statement of the build-owned noise stream (see diffusion-ccsp_amd/noise.py for the
// specification and the numpy statement; oracle/ccsp_oracle.c holds the host C statement).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ccsp {

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

// Box-Muller on 24-bit uniforms (exact in fp32): u1 in (0,1], u2 in [0,1)
__device__ __forceinline__ void box_muller(uint32_t ra, uint32_t rb, float& z0, float& z1) {
    const float u1 = (float)((ra >> 8) + 1u) * (1.0f / 16777216.0f);
    const float u2 = (float)(rb >> 8) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincospif(2.0f * u2, &s, &c);
    z0 = rad * c;
    z1 = rad * s;
}

// randn(N,P)[row, col] of call index `call` (stream 0)
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t row, uint32_t call, int col) {
    const u32x4 r = philox4x32_10((uint32_t)row, call, (uint32_t)(col >> 2), 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    float z0, z1;
    if ((col & 2) == 0) box_muller(r.x, r.y, z0, z1); else box_muller(r.z, r.w, z0, z1);
    return (col & 1) ? z1 : z0;
}

// rand(N)[row] of uniform-call index `call` (stream 1), in [0,1)
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t row, uint32_t call) {
    const u32x4 r = philox4x32_10((uint32_t)row, call, 0u, 1u, (uint32_t)seed, (uint32_t)(seed >> 32));
    return (float)(r.x >> 8) * (1.0f / 16777216.0f);
}

}  // namespace ccsp
