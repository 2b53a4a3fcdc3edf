// helpers of the energy mode that k_edge<H, true> needs (included inside the anonymous namespace of
// ccsp_hip.hip, before k_edge; the kernels themselves are in ccsp_energy.h)
#pragma once

__device__ __forceinline__ float silu_grad_fast(float v) {       // d/dv [v sigmoid(v)]
    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
    return sg * (1.0f + v * (1.0f - sg));
}

// deterministic block sum of one float per thread (256 threads); result valid in thread 0
__device__ __forceinline__ float block_sum_256(float v, float* red /*[256] LDS*/) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
#pragma unroll
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    return red[0];
}

