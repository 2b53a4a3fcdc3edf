// helpers of the energy mode that k_edge<H, true> needs (included inside the anonymous namespace of
// ccsp_hip.hip, before k_edge; the kernels themselves are in ccsp_energy.h)
#pragma once

__device__ __forceinline__ float silu_grad_fast(float v) {       // d/dv [v sigmoid(v)]
    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
    return sg * (1.0f + v * (1.0f - sg));
}

// deterministic block sum of one float per thread (256 threads), every thread gets the result: a butterfly of wavefront
// shuffles inside each of the four waves (fixed order), then the four wave sums through LDS -- two barriers instead of the nine
// of an LDS tree.  `red` needs 4 floats and may be reused right after the call.
__device__ __forceinline__ float block_sum_256(float v, float* red /*[>= 4] LDS*/) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return tot;
}
