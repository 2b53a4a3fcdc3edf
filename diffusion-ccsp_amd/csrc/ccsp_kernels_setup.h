// ccsp_kernels_setup.h -- one-time model kernels and the node encoders (geometry / grasp / pose; fp32 MFMA and f16-pipe forms).
// A fragment of the ONE translation unit csrc/ccsp_hip.hip (included there, at this position, inside its namespaces): not a standalone header.
// ------------------------------------------------------------------------------------------
// one-time model kernels
// ------------------------------------------------------------------------------------------

// fp32 -> three bf16 terms (ccsp_bf16x3.h explains the scheme)
__device__ __forceinline__ unsigned short bf16_rn_bits(float x) {          // round-to-nearest-even
    unsigned int u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_bits_f(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }

// x -> (x1, x2, x3) bf16 bit patterns.  Inf/NaN stay in x1 (x - x1 is NaN/0 there, harmless: NaN is data)
__device__ __forceinline__ void split3(float x, unsigned short& h1, unsigned short& h2, unsigned short& h3) {
    h1 = bf16_rn_bits(x);
    const float r1 = x - bf16_bits_f(h1);
    h2 = bf16_rn_bits(r1);
    const float r2 = r1 - bf16_bits_f(h2);
    h3 = bf16_rn_bits(r2);
}

// SinusoidalPosEmb (denoise_fn.py:38-50) for every t: e[t, :] fp32, evaluated like the reference
__global__ void k_sinusoid(int T, int H, float* __restrict__ e) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = H / 2;
    if (idx >= T * half) return;
    const int t = idx / half, k = idx % half;
    const float c = (float)(-(log(10000.0) / (double)(half - 1)));
    const float f = expf((float)k * c);
    const float a = (float)t * f;
    e[(size_t)t * H + k] = sinf(a);
    e[(size_t)t * H + half + k] = cosf(a);
}

// y[r, o] = act(b[o] + sum_k x[r,k] W[o,k]); one thread per output (set-up only, not hot)
__global__ void k_linear_rows(int R, int K, int O, const float* __restrict__ x, int ldx, const float* __restrict__ W, int ldw,
                              const float* __restrict__ b, int act /*0 none, 1 mish, 2 silu*/, float* __restrict__ y, int ldy) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)R * O) return;
    const int r = (int)(idx / O), o = (int)(idx % O);
    const float* xr = x + (size_t)r * ldx;
    const float* wr = W + (size_t)o * ldw;
    float acc = 0.0f;
    for (int k = 0; k < K; ++k) acc = fmaf(xr[k], wr[k], acc);
    acc += b ? b[o] : 0.0f;
    if (act == 1) acc = mish_f(acc);
    if (act == 2) acc = silu_f(acc);
    y[(size_t)r * ldy + o] = acc;
}

// SinusoidalPosEmb for arbitrary (float) t values: e[r, :] like k_sinusoid (operator API, not hot)
__global__ void k_sinusoid_values(int R, int H, const float* __restrict__ tv, float* __restrict__ e) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = H / 2;
    if (idx >= R * half) return;
    const int r = idx / half, k = idx % half;
    const float c = (float)(-(log(10000.0) / (double)(half - 1)));
    const float a = tv[r] * expf((float)k * c);
    e[(size_t)r * H + k] = sinf(a);
    e[(size_t)r * H + half + k] = cosf(a);
}

// type MLP of ONE constraint type on caller-supplied embeddings (ConstraintDiffuser._process_constraint, denoise_fn.py:341-356):
// h[r, o] = SiLU(b[o] + [grasp_a | geom_a geom_b | pose_a pose_b | time] . W[o, :]) from the per-segment weight slices
__global__ void k_type_mlp_rows(int R, int H, const float* __restrict__ gr /*[R,H] or null*/, const float* __restrict__ ge /*[R,2,H]*/,
                                const float* __restrict__ pe /*[R,2,H]*/, const float* __restrict__ te /*[R,H]*/,
                                const float* __restrict__ Wr, const float* __restrict__ Wg0, const float* __restrict__ Wg1,
                                const float* __restrict__ Wp0, const float* __restrict__ Wp1, const float* __restrict__ Wt,
                                const float* __restrict__ bias, float* __restrict__ h /*[R,2H]*/) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)R * 2 * H) return;
    const int r = (int)(idx / (2 * H)), o = (int)(idx % (2 * H));
    float acc = 0.0f;                                     // segments in the order of the concatenated input
    if (gr) for (int k = 0; k < H; ++k) acc = fmaf(gr[(size_t)r * H + k], Wr[(size_t)o * H + k], acc);
    for (int k = 0; k < H; ++k) acc = fmaf(ge[((size_t)r * 2) * H + k], Wg0[(size_t)o * H + k], acc);
    for (int k = 0; k < H; ++k) acc = fmaf(ge[((size_t)r * 2 + 1) * H + k], Wg1[(size_t)o * H + k], acc);
    for (int k = 0; k < H; ++k) acc = fmaf(pe[((size_t)r * 2) * H + k], Wp0[(size_t)o * H + k], acc);
    for (int k = 0; k < H; ++k) acc = fmaf(pe[((size_t)r * 2 + 1) * H + k], Wp1[(size_t)o * H + k], acc);
    for (int k = 0; k < H; ++k) acc = fmaf(te[(size_t)r * H + k], Wt[(size_t)o * H + k], acc);
    h[idx] = silu_f(acc + bias[o]);
}

// dst[r, c] = src[r, col0 + c]  (weight re-layout)
__global__ void k_copy_cols(int R, int Ccols, const float* __restrict__ src, int lds, int col0, float* __restrict__ dst, int ldd) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)R * Ccols) return;
    const int r = (int)(idx / Ccols), c = (int)(idx % Ccols);
    dst[(size_t)r * ldd + c] = src[(size_t)r * lds + col0 + c];
}

// dst[c, r] = src[r, c]
__global__ void k_transpose(int R, int Ccols, const float* __restrict__ src, float* __restrict__ dst) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)R * Ccols) return;
    const int r = (int)(idx / Ccols), c = (int)(idx % Ccols);
    dst[(size_t)c * R + r] = src[idx];
}

// ------------------------------------------------------------------------------------------
// node encoder: Linear(in->H/2) SiLU Linear(H/2->H) SiLU  (denoise_fn.py:227-250)
// ------------------------------------------------------------------------------------------

struct EncW {
    const float* W0;   // [H/2, in_dim]
    const float* b0;   // [H/2]
    const float* W2T;  // [H/2, H]   (transposed: lanes read consecutive output columns)
    const float* b2;   // [H]
    int in_dim;
    const float* W2F;  // layer-2 weight in v_mfma_f32_16x16x4_f32 B-fragment order (k_pack_enc_frag), or null
    const unsigned short* W2H;   // the same weight * 2^w2_exp as two fp16 planes in v_mfma_f32_16x16x32_f16 fragment order, or null
    int w2_exp;
    float c1, c2;                // |layer-1 pre-activation| <= c1 * max|x| + c2  (largest absolute row sum of W0, largest |b0|)
};

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// W2 [H, H/2] (nn.Linear weight) -> B fragments of 16x16x4: for wave w (H/4 columns), k-step ks,
// lane l, column tile j:  W2F[((w*KS + ks)*64 + l)*TPW + j] = W2[w*16*TPW + j*16 + (l&15)][ks*4 + (l>>4)]
__global__ void k_pack_enc_frag(int H, const float* __restrict__ W2, float* __restrict__ W2F) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int TPW = H / 64, KS = H / 8;
    if (idx >= H * (H / 2)) return;
    const int j = idx % TPW, l = (idx / TPW) % 64, ks = (idx / (TPW * 64)) % KS, w = idx / (TPW * 64 * KS);
    const int col = w * 16 * TPW + j * 16 + (l & 15), k = ks * 4 + (l >> 4);
    W2F[idx] = W2[(size_t)col * (H / 2) + k];
}

// pose encoder of a 16-node tile on the matrix cores: layer 1 (P -> H/2) on the VALU into LDS,
// layer 2 (H/2 -> H) as 16 x H x H/2 with v_mfma_f32_16x16x4_f32 (M = the 16 nodes of the tile).
// s1 has row stride H/2 + 1 (conflict-free A-fragment reads).  All 256 threads participate.
// The node kernel is latency-bound (144 workgroups, a chain of dependent global loads), so every
// weight the encoder needs that does not depend on the data is requested at kernel entry
// (enc_prefetch) and is in flight while the CSR reduction and the pose update run.
template <int H>
struct EncPrefetch {
    static constexpr int TPW = H / 64, KS = H / 8, PF = KS >= 16 ? 8 : KS / 2;   // PF k-steps of layer-2 fragments prefetched
    float w0[8], b0;                                          // layer-1 row of this thread's output column
    float wf[PF][TPW];
    float b2[TPW][4];                                         // bias of this lane's 4 consecutive output columns per tile
};

template <int H>
__device__ __forceinline__ void enc_prefetch(const EncW w, EncPrefetch<H>& pf) {
    using PFT = EncPrefetch<H>;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = tid % (H / 2);
#pragma unroll
    for (int d = 0; d < 8; ++d) pf.w0[d] = d < w.in_dim ? w.W0[j * w.in_dim + d] : 0.0f;
    pf.b0 = w.b0[j];
    const float* wf = w.W2F + ((size_t)wave * PFT::KS * 64 + lane) * PFT::TPW;
#pragma unroll
    for (int ks = 0; ks < PFT::PF; ++ks)
#pragma unroll
        for (int q = 0; q < PFT::TPW; ++q) pf.wf[ks][q] = wf[(size_t)ks * 64 * PFT::TPW + q];
#pragma unroll
    for (int q = 0; q < PFT::TPW; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) pf.b2[q][r] = w.b2[wave * 16 * PFT::TPW + q * 16 + 4 * (lane >> 4) + r];
    __builtin_amdgcn_sched_barrier(0);      // keep these loads at kernel entry (hipcc would sink them to first use)
}

// f16x2 operands (ccsp_f16x2.h explains the scheme): exponent e with amax * 2^e in [2^13, 2^14), 0 for zero /
// denormal / Inf / NaN; and the two fp16 terms of an already scaled value
__device__ __forceinline__ int h2_scale_exp(float amax) {
    const int be = (int)((__float_as_uint(amax) >> 23) & 0xffu);
    return (be == 0 || be == 255) ? 0 : 140 - be;
}
__device__ __forceinline__ void split2h(float xs, unsigned short& h1, unsigned short& h2) {
    const _Float16 a = (_Float16)xs;
    const _Float16 b = (_Float16)(xs - (float)a);
    h1 = __builtin_bit_cast(unsigned short, a);
    h2 = __builtin_bit_cast(unsigned short, b);
}

// The f16x2 operand planes of the pose embeddings (EncOut::h2): product build = both planes of a row's 32-column chunk side by side, [N][H / 32][2][32]
// (one 128-byte L2 -> L1 line per (row, K chunk) of the forward row GEMM: k_rowgemm_h2, ILA); experiments build = planar [2][N][H], which its
// other consumers (k_rowgemm_h2d, the fused evaluation kernels) read.
#ifdef CCSP_EXPERIMENTS
#define CCSP_A_INTERLEAVED 0
#else
#define CCSP_A_INTERLEAVED 1
#endif
struct EncOut {
    float* f32;                 // [N, H] embeddings, or null
    unsigned short* bf3;        // [3][N][H] bf16 planes (ccsp_bf16x3.h), or null
    unsigned short* h2;         // fp16 planes of the row scaled by 2^h2_exp[n] (ccsp_f16x2.h; layout: CCSP_A_INTERLEAVED), or null
    int* h2_exp;                // [N]
};

// Epilogue of the MFMA pose encoders.  The weight fragment is the A operand, so the product comes out transposed: C/D layout
// of 16x16 is col = lane & 15 -> the node, row = (lane >> 4) * 4 + reg -> four CONSECUTIVE output columns per lane
// (v[tile][reg]).  One 16-byte store per tile (and 8 bytes per 2-byte plane) instead of four scattered ones: the 2-byte
// plane stores of the untransposed layout cost 5 % of the whole chain (tools/ab.sh).
template <int H>
__device__ __forceinline__ void enc_store_tile(const float (&v)[H / 64][4], float (*smax)[NODE_TILE], int node0, int N, const EncOut out, int n_lim = -1 /*nodes >= n_lim are not this block's (default: N)*/) {
    constexpr int TPW = H / 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = node0 + (lane & 15);
    if (n_lim < 0) n_lim = N;
    int e2 = 0;
    if (out.h2) {
        // largest |element| of every node row: in-lane over the lane's 4 TPW columns, the four lanes of the wave that
        // share the node (lane & 15), then the four waves through LDS (fmaxf skips NaN; an Inf row gets exponent 0)
        float m = 0.0f;
#pragma unroll
        for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(v[j][r]));
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        if (lane < NODE_TILE) smax[wave][lane] = m;
        __syncthreads();
        CCSP_TRK(2, 4);
        m = fmaxf(fmaxf(smax[0][lane & 15], smax[1][lane & 15]), fmaxf(smax[2][lane & 15], smax[3][lane & 15]));
        e2 = h2_scale_exp(m);
        if (n < n_lim && wave == 0 && lane < NODE_TILE) out.h2_exp[n] = e2;
    }
    if (n < n_lim) {
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            const int c0 = wave * 16 * TPW + j * 16 + 4 * (lane >> 4);
            const size_t o = (size_t)n * H + c0;
            const size_t pl = (size_t)N * H;
            if (out.f32) *reinterpret_cast<float4*>(out.f32 + o) = float4{v[j][0], v[j][1], v[j][2], v[j][3]};
            if (out.bf3) {                                    // operand planes of k_rowgemm_bf*, written by the producer
                unsigned short h1[4], h2[4], h3[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) split3(v[j][r], h1[r], h2[r], h3[r]);
                *reinterpret_cast<uint2*>(out.bf3 + o) = make_uint2(h1[0] | ((unsigned)h1[1] << 16), h1[2] | ((unsigned)h1[3] << 16));
                *reinterpret_cast<uint2*>(out.bf3 + pl + o) = make_uint2(h2[0] | ((unsigned)h2[1] << 16), h2[2] | ((unsigned)h2[3] << 16));
                *reinterpret_cast<uint2*>(out.bf3 + 2 * pl + o) = make_uint2(h3[0] | ((unsigned)h3[1] << 16), h3[2] | ((unsigned)h3[3] << 16));
            }
            if (out.h2) {                                     // operand planes of k_rowgemm_h2
                unsigned short h1[4], h2[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) split2h(ldexpf(v[j][r], e2), h1[r], h2[r]);
#if CCSP_A_INTERLEAVED
                const size_t oi = (size_t)n * (2 * H) + (size_t)(c0 >> 5) * 64 + (c0 & 31);
                *reinterpret_cast<uint2*>(out.h2 + oi) = make_uint2(h1[0] | ((unsigned)h1[1] << 16), h1[2] | ((unsigned)h1[3] << 16));
                *reinterpret_cast<uint2*>(out.h2 + oi + 32) = make_uint2(h2[0] | ((unsigned)h2[1] << 16), h2[2] | ((unsigned)h2[3] << 16));
#else
                *reinterpret_cast<uint2*>(out.h2 + o) = make_uint2(h1[0] | ((unsigned)h1[1] << 16), h1[2] | ((unsigned)h1[3] << 16));
                *reinterpret_cast<uint2*>(out.h2 + pl + o) = make_uint2(h2[0] | ((unsigned)h2[1] << 16), h2[2] | ((unsigned)h2[3] << 16));
#endif
            }
        }
    }
}

template <int H>
__device__ __forceinline__ void encode_tile_mfma(const EncW w, const EncPrefetch<H>& pf, float (*xs)[8],
                                                 float (*s1)[H / 2 + 1], float (*smax)[NODE_TILE], int node0, int N, const EncOut out) {
    using PFT = EncPrefetch<H>;
    constexpr int TPW = PFT::TPW, KS = PFT::KS, PF = PFT::PF;
    const int tid = threadIdx.x;
    if constexpr (256 % (H / 2) == 0) {
        const int j = tid % (H / 2);
#pragma unroll
        for (int i = 0; i < H / 32; ++i) {
            const int n = tid / (H / 2) + i * (512 / H);
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < 8; ++d) acc = fmaf(xs[n][d], pf.w0[d], acc);     // xs columns >= in_dim are 0
            s1[n][j] = silu_fast(acc + pf.b0);
        }
    } else {                                     // hidden widths whose half does not divide 256: (node, unit) items in a plain loop
        for (int idx = tid; idx < NODE_TILE * (H / 2); idx += 256) {
            const int n = idx / (H / 2), j = idx % (H / 2);
            float acc = 0.0f;
            for (int d = 0; d < w.in_dim; ++d) acc = fmaf(xs[n][d], w.W0[j * w.in_dim + d], acc);
            s1[n][j] = silu_fast(acc + w.b0[j]);
        }
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    floatx4 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[j] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    const float* wf = w.W2F + ((size_t)wave * KS * 64 + lane) * TPW;
    const float* ap = &s1[lane & 15][lane >> 4];
    // the remaining fragments are requested before the first MFMA is issued
    float rest[KS - PF][TPW];
#pragma unroll
    for (int ks = PF; ks < KS; ++ks)
#pragma unroll
        for (int q = 0; q < TPW; ++q) rest[ks - PF][q] = wf[(size_t)ks * 64 * TPW + q];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const float a = ap[ks * 4];
#pragma unroll
        for (int j = 0; j < TPW; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ks < PF ? pf.wf[ks][j] : rest[ks - PF][j], a, acc[j], 0, 0, 0);
    }
    float v[TPW][4];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[j][r] = silu_fast(acc[j][r] + pf.b2[j][r]);
    enc_store_tile<H>(v, smax, node0, N, out);
}

// ---- the pose encoder's second layer on the f16 matrix pipe (hidden_dim 256, f16x2 mode; scheme of ccsp_f16x2.h) ----
// 16 nodes x 256 x 128 is 4096 cycles of v_mfma_f32_16x16x4_f32 per wave and 768 of v_mfma_f32_16x16x32_f16 with three
// products -- on a kernel that is one latency chain.  Operands: the weight * 2^w2_exp as two fp16 planes in A-fragment order
// (k_pack_enc_frag_h2); the layer-1 activations s1 = SiLU(y1) of a node scaled by 2^e with e from the BOUND
// |s1| <= |y1| <= c1 max|x| + c2 (no reduction over the row needed; a loose bound costs nothing, see ccsp_f16x2.h).
//   W2H[plane][(((w * 4 + ks) * 4 + j) * 64 + l) * 8 + e8] = term of W2[w*64 + j*16 + (l & 15)][ks*32 + 8 (l >> 4) + e8] * 2^e
__global__ void k_pack_enc_frag_h2(const float* __restrict__ W2 /*[256,128]*/, int e, unsigned short* __restrict__ W2H) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 256 * 128) return;
    const int e8 = idx & 7, l = (idx >> 3) & 63, j = (idx >> 9) & 3, ks = (idx >> 11) & 3, w = idx >> 13;
    const int col = w * 64 + j * 16 + (l & 15), k = ks * 32 + 8 * (l >> 4) + e8;
    unsigned short a, b;
    split2h(ldexpf(W2[col * 128 + k], e), a, b);
    W2H[idx] = a;
    W2H[256 * 128 + idx] = b;
}

constexpr int ENC_H2_LD = 136;      // fp16 row stride of the s1 planes: 272 bytes, 16-byte fragment reads of 16 rows hit all banks once

struct EncPrefetchH {
    float w0[8], b0;
    half8 wa[4][2][4];              // the layer-2 fragments of the wave: [k-step][plane][tile], 128 VGPRs
    float b2[4][4];
};

__device__ __forceinline__ void enc_prefetch_h2(const EncW w, EncPrefetchH& pf) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = tid % 128;
#pragma unroll
    for (int d = 0; d < 8; ++d) pf.w0[d] = d < w.in_dim ? w.W0[j * w.in_dim + d] : 0.0f;
    pf.b0 = w.b0[j];
    const half8* wh = reinterpret_cast<const half8*>(w.W2H) + (size_t)wave * 16 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) pf.wa[ks][p][q] = wh[(size_t)p * 4096 + (ks * 4 + q) * 64];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) pf.b2[q][r] = w.b2[wave * 64 + q * 16 + 4 * (lane >> 4) + r];
    __builtin_amdgcn_sched_barrier(0);      // keep these loads at kernel entry (hipcc would sink them to first use)
}

// s1h: [2][NODE_TILE][ENC_H2_LD] fp16 bits in LDS; sexp: [NODE_TILE] row exponents (enc_row_exp, written with xs).  All 256
// threads participate.
__device__ __forceinline__ void encode_tile_h2(const EncW w, const EncPrefetchH& pf, float (*xs)[8], unsigned short* s1h, int* sexp,
                                               float (*smax)[NODE_TILE], int node0, int N, const EncOut out) {
    constexpr int H = 256, LD = ENC_H2_LD;
    const int tid = threadIdx.x;
    {
        const int j = tid % 128;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = tid / 128 + 2 * i;
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < 8; ++d) acc = fmaf(xs[n][d], pf.w0[d], acc);        // columns >= in_dim are 0
            unsigned short h1, h2;
            split2h(ldexpf(silu_fast(acc + pf.b0), sexp[n]), h1, h2);
            s1h[n * LD + j] = h1;
            s1h[(NODE_TILE + n) * LD + j] = h2;
        }
    }
    __syncthreads();
    CCSP_TRK(2, 2);
    const int wave = tid >> 6, lane = tid & 63;
    floatx4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    const unsigned short* bp = s1h + (lane & 15) * LD + 8 * (lane >> 4);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const half8 b1 = *reinterpret_cast<const half8*>(bp + ks * 32);
        const half8 b2 = *reinterpret_cast<const half8*>(bp + NODE_TILE * LD + ks * 32);
        // smallest terms first; consecutive MFMAs go to different accumulators
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf.wa[ks][1][j], b1, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf.wa[ks][0][j], b2, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf.wa[ks][0][j], b1, acc[j], 0, 0, 0);
    }
    CCSP_TRK(2, 3);
    const int eu = -(sexp[lane & 15] + w.w2_exp);
    float v[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[j][r] = silu_fast(ldexpf(acc[j][r], eu) + pf.b2[j][r]);
    enc_store_tile<H>(v, smax, node0, N, out);
}

// xs: [NODE_TILE][8] in LDS; s1: [NODE_TILE][H/2] in LDS.  All 256 threads participate.
template <int H>
__device__ __forceinline__ void encode_tile(const EncW w, float (*xs)[8], float (*s1)[H / 2], int node0, int N,
                                            float* __restrict__ out /*[N,H]*/) {
    const int tid = threadIdx.x;
    for (int idx = tid; idx < NODE_TILE * (H / 2); idx += 256) {
        const int n = idx / (H / 2), j = idx % (H / 2);
        float acc = 0.0f;
        for (int d = 0; d < w.in_dim; ++d) acc = fmaf(xs[n][d], w.W0[j * w.in_dim + d], acc);
        s1[n][j] = silu_fast(acc + w.b0[j]);
    }
    __syncthreads();
    if constexpr (256 % H == 0) {
        constexpr int NG = 256 / H;             // node groups per workgroup (H=256: 1, H=64: 4)
        constexpr int NPT = NODE_TILE / NG;     // nodes per thread
        const int j = tid % H, g = tid / H;
        float acc[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) acc[i] = 0.0f;
        for (int k = 0; k < H / 2; ++k) {
            const float wv = w.W2T[(size_t)k * H + j];
#pragma unroll
            for (int i = 0; i < NPT; ++i) acc[i] = fmaf(s1[g * NPT + i][k], wv, acc[i]);
        }
        const float bj = w.b2[j];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int n = node0 + g * NPT + i;
            if (n < N) out[(size_t)n * H + j] = silu_fast(acc[i] + bj);
        }
    } else {                                    // widths that do not divide 256: every thread walks columns tid, tid + 256, ... for all nodes
        for (int j = tid; j < H; j += 256) {
            float acc[NODE_TILE];
#pragma unroll
            for (int i = 0; i < NODE_TILE; ++i) acc[i] = 0.0f;
            for (int k = 0; k < H / 2; ++k) {
                const float wv = w.W2T[(size_t)k * H + j];
#pragma unroll
                for (int i = 0; i < NODE_TILE; ++i) acc[i] = fmaf(s1[i][k], wv, acc[i]);
            }
            const float bj = w.b2[j];
#pragma unroll
            for (int i = 0; i < NODE_TILE; ++i) {
                const int n = node0 + i;
                if (n < N) out[(size_t)n * H + j] = silu_fast(acc[i] + bj);
            }
        }
    }
}

template <int H>
__global__ __launch_bounds__(256) void k_encode(int N, const float* __restrict__ in, int ld, int off, EncW w,
                                                float* __restrict__ out) {
    __shared__ float xs[NODE_TILE][8];
    __shared__ float s1[NODE_TILE][H / 2];
    const int node0 = blockIdx.x * NODE_TILE;
    const int tid = threadIdx.x;
    if (tid < NODE_TILE * 8) {
        const int n = tid / 8, d = tid % 8;
        const int node = node0 + n;
        xs[n][d] = (node < N && d < w.in_dim) ? in[(size_t)node * ld + off + d] : 0.0f;
    }
    __syncthreads();
    encode_tile<H>(w, xs, s1, node0, N, out);
}

