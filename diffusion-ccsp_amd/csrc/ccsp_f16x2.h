// fp32-class GEMMs on the f16 matrix cores ("f16x2"): every fp32 operand is scaled by an exact power of two
// into the fp16 range and split into two fp16 terms  x 2^e = x1 + x2  (11 significand bits each, 22 together),
// and a product a*b is accumulated in fp32 as the three cross terms of weight >= 2^-11:
//        a2 b1 + a1 b2 + a1 b1
// Every fp16 x fp16 product is exact in fp32; the dropped a2 b2 term and the split residuals are <= 2^-22
// relative -- the accuracy class of a chained fp32 MFMA accumulation (measured 2.4e-7 .. 4.5e-7 of sum|ab|,
// ccsp_bf16x3.h), so the parity bars are the ones of the fp32 path.  Against the six-product bf16 scheme of
// ccsp_bf16x3.h this halves the matrix-pipe time and cuts the operand bytes (global, LDS write, LDS read)
// by a third.
//
// Range.  fp16 holds 2^-24 .. 65504 and the reference sampler passes through 1e6 .. 1e19 transients in its
// first timesteps (tests/golden: chain_q256_T1000_B4), so operands are scaled, always exactly:
//   * weights: one exponent per weight tensor, chosen at model creation from its largest element;
//   * pose embeddings (A of the row GEMM): one exponent per node row, from the row's largest element,
//     computed by the producer (the encoder epilogue of k_node) and stored next to the planes;
//   * decoder input h = SiLU(U[u0] + U[u1]) (A of the edge kernel): one exponent per (edge, half) row from
//     the BOUND  max|U[u0]| + max|U[u1]| >= |h|  over that half, where the row maxima are a by-product of
//     the row GEMM's epilogue (umax).  A bound instead of the exact maximum costs nothing: elements keep all
//     22 bits down to 2^-18 of the scaled maximum (fp16 normal range), and smaller ones an absolute
//     precision of 2^-39 of it.
// The scaled row maximum lies in [2^13, 2^14), K-long sums of products stay below 2^38, and the result is
// unscaled by the exact inverse power of two in the epilogue (v_ldexp_f32).  Non-finite rows (NaN is data)
// get exponent 0 and propagate as NaN / Inf.
//
// Kernels (hidden_dim 256 only; the small fixtures at hidden_dim 64 stay on ccsp_bf16x3.h):
//   k_rowgemm_h2<KD, ND, MODE>   128 x 128 tiles, 4 waves as 2(M) x 2(N), 64 x 64 per wave (every LDS fragment
//                          feeds two MFMA tiles: 8 ds_read_b128 per 12 MFMAs, against 9 per 12 at half the
//                          flops each in k_rowgemm_bf2); epilogue per wave through a wave-private LDS tile:
//                          16-byte row-contiguous base loads / U stores and the row maxima of U for the edge
//                          kernel.  MODE picks the staging (rowgemm_h2_mode: 4 for tile lists of at most one workgroup per CU,
//                          else 6 up to 2.25 workgroups of 64-row tiles per CU, else 0; 1 / 2 / 3 / 5 through CCSP_ROW_MODE -- measured
//                          equal or behind since round 3):
//                          0 = one 16 KB stage and one register set, 3 workgroups per CU (lists longer than 2 per CU);
//                          1 = two stages, two register sets (chunk c+2 in flight while c is multiplied);
//                          2 = two stages filled by global_load_lds_dwordx4, swizzle applied to the source address;
//                          3 = ring of four such stages, counted s_waitcnt vmcnt, bare s_barrier (one workgroup per CU);
//                          5 = weight planes only in a four-stage ring, A fragments straight from global memory (experiment);
//                          6 = MODE 0's register staging on 64 x 128 tiles, four workgroups per CU (mid-size tile lists);
//                          4 = the ring on 64 x 128 tiles (32 x 64 per wave), three stages: tile lists of at most one
//                              workgroup per CU, where the kernel is a latency chain and not a throughput problem.
//                          <256, 512> is the forward GEMM, <512, 256> the energy mode's transpose GEMM
//   k_edge_h2<ENERGY, MT, L2>  MT = 2: 128 rows = 64 sorted edges x both output halves per workgroup (the decoder
//                          weight chunk is staged once for both halves); MT = 1: 32 edges, 3 workgroups per CU
//                          (default whenever all tiles then fit in one round).  SiLU + scale + split of chunk
//                          c+1 issued between the MFMA groups of chunk c.  L2 = 1 (grids of at most one workgroup
//                          per CU): the second decoder layer split over the waves (h2_decoder_l2)
//   k_edge_h2s<ENERGY>     the edge kernel of small batches (<= 4096 active edges): 16-edge tiles, four waves sharing the rows
//   k_edge_bwd_h2          energy mode: g_h = g_o Wd2 (VALU), g_q = g_h SiLU'(q), g_z = (g_q Wd1) SiLU'(z) with the
//                          GEMM on the same three products; the row exponent comes from the bound
//                          1.1 * max|Wd2| * sum|g_o| >= |g_h|
// The pose encoder's second layer (k_node, k_node_energy_h2) runs on the same scheme with v_mfma_f32_16x16x32_f16:
// encode_tile_h2 in ccsp_hip.hip.
// Included inside the anonymous namespace of ccsp_hip.hip.
#pragma once

// (half8 is declared in ccsp_hip.hip: the node kernel's encoder uses it too)

// (h2_scale_exp / split2h live in ccsp_hip.hip: the node kernel's encoder writes planes too)

// dst[plane][i] = plane-th fp16 term of src[i] * 2^e   (weights, once per model)
__global__ void k_split2h(long n, const float* __restrict__ src, int e, unsigned short* __restrict__ dst) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned short a, b;
    split2h(ldexpf(src[i], e), a, b);
    dst[i] = a; dst[n + i] = b;
}

// largest |src[i]| as fp32 bits (non-negative floats order like their bit patterns; NaN ranks above Inf, both
// give exponent 0).  atomicMax is order-independent, so the result is deterministic.
__global__ void k_absmax_bits(long n, const float* __restrict__ src, unsigned int* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int v = i < n ? (__float_as_uint(src[i]) & 0x7fffffffu) : 0u;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const unsigned int o = (unsigned int)__shfl_xor((int)v, s);
        v = v > o ? v : o;
    }
    if ((threadIdx.x & 63) == 0 && v) atomicMax(out, v);
}

// planar fp16 planes [2][rows][K] -> the forward row GEMM's weight layout [rows][K / 32][2][32] (both planes of a row's 32-element K chunk in one
// 128-byte line: k_rowgemm_h2, ILW)
__global__ void k_interleave_planes(long n /*rows * K*/, int K, const unsigned short* __restrict__ src, unsigned short* __restrict__ dst) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long row = i / K;
    const int k = (int)(i - row * K);
    const long o = row * 2 * K + (long)(k >> 5) * 64 + (k & 31);
    dst[o] = src[i];
    dst[o + 32] = src[n + i];
}

constexpr int H2_BK = 32;                    // K chunk: two MFMA k-steps of 16
constexpr int H2_BPL = 128 * H2_BK;          // fp16 elements per plane of the 128-row B operand stage
// rows are unpadded 64-byte K chunks, the 16-byte piece index XOR-swizzled by (row >> 2) & 3 (rb2_off of
// ccsp_bf16x3.h: conflict-free ds_read_b128 fragment reads and 16-byte staging writes)
__device__ __forceinline__ int h2_off(int row, int piece) { return row * H2_BK + ((piece ^ ((row >> 2) & 3)) << 3); }

// The decoder's A operand, four elements of a row: SiLU(a + b) scaled by 2^e and split into two fp16 terms (packed hi / lo pairs ready for the LDS
// planes).  The adds and multiplies are written on two-element vectors so that hipcc issues v_pk_add_f32 / v_pk_mul_f32 (two fp32 per lane and
// instruction, same IEEE results per element: bitwise the scalar form silu_fast(a + b) -> ldexpf -> split2h) -- the activation is the VALU half of
// the edge kernels' chunk (profiles/r02_findings.md: ~420 VALU cycles per chunk and wave against 384 of MFMA).
typedef float h2_f2 __attribute__((ext_vector_type(2)));
// (inline asm: left to itself hipcc scalarises most two-element vector operations again.  The compiler's hazard recogniser does not see inside an asm
// statement: on gfx940+ a non-transcendental VALU instruction that reads the result of v_exp_f32 / v_rcp_f32 needs one wait state after it (trans
// forwarding hazard; hipcc inserts the s_nop for its own instructions) -- the _t forms carry it themselves and are the ones to use on such operands.
// Without it the consumer reads the register's previous contents: tools/pk_probe.hip reproduces that.)
__device__ __forceinline__ h2_f2 h2_pk_add(h2_f2 x, h2_f2 y) { h2_f2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ h2_f2 h2_pk_mul(h2_f2 x, h2_f2 y) { h2_f2 r; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ h2_f2 h2_pk_add_t(h2_f2 x, h2_f2 y) { h2_f2 r; asm("s_nop 0\n\tv_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ h2_f2 h2_pk_mul_t(h2_f2 x, h2_f2 y) { h2_f2 r; asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ h2_f2 h2_pk_sub(h2_f2 x, h2_f2 y) { h2_f2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ void h2_act4(const float4& a, const float4& b, int e, uint2& hi, uint2& lo) {
#ifdef CCSP_ACT_SCALAR                       // the scalar form the packed one is measured against (tools/mkvariant.py act_scalar); same bits
    const float h[4] = {silu_fast(a.x + b.x), silu_fast(a.y + b.y), silu_fast(a.z + b.z), silu_fast(a.w + b.w)};
    unsigned short p1[4], p2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split2h(ldexpf(h[i], e), p1[i], p2[i]);
    hi = make_uint2(p1[0] | (unsigned int)p1[1] << 16, p1[2] | (unsigned int)p1[3] << 16);
    lo = make_uint2(p2[0] | (unsigned int)p2[1] << 16, p2[2] | (unsigned int)p2[3] << 16);
    return;
#endif
    const h2_f2 z0 = h2_pk_add(h2_f2{a.x, a.y}, h2_f2{b.x, b.y}), z1 = h2_pk_add(h2_f2{a.z, a.w}, h2_f2{b.z, b.w});
    const h2_f2 k = {-1.4426950408889634f, -1.4426950408889634f}, one = {1.0f, 1.0f};
    const h2_f2 t0 = h2_pk_mul(z0, k), t1 = h2_pk_mul(z1, k);
    const h2_f2 d0 = h2_pk_add_t(h2_f2{__builtin_amdgcn_exp2f(t0.x), __builtin_amdgcn_exp2f(t0.y)}, one);
    const h2_f2 d1 = h2_pk_add_t(h2_f2{__builtin_amdgcn_exp2f(t1.x), __builtin_amdgcn_exp2f(t1.y)}, one);
    const h2_f2 h0 = h2_pk_mul_t(z0, h2_f2{__builtin_amdgcn_rcpf(d0.x), __builtin_amdgcn_rcpf(d0.y)});
    const h2_f2 h1 = h2_pk_mul_t(z1, h2_f2{__builtin_amdgcn_rcpf(d1.x), __builtin_amdgcn_rcpf(d1.y)});
    const h2_f2 s0 = {ldexpf(h0.x, e), ldexpf(h0.y, e)}, s1 = {ldexpf(h1.x, e), ldexpf(h1.y, e)};
    typedef _Float16 h2_h2 __attribute__((ext_vector_type(2)));
    const h2_h2 a0 = __builtin_convertvector(s0, h2_h2), a1 = __builtin_convertvector(s1, h2_h2);          // v_cvt_pk_f16_f32 (round to nearest even)
    const h2_f2 r0 = h2_pk_sub(s0, __builtin_convertvector(a0, h2_f2)), r1 = h2_pk_sub(s1, __builtin_convertvector(a1, h2_f2));
    const h2_h2 b0 = __builtin_convertvector(r0, h2_h2), b1 = __builtin_convertvector(r1, h2_h2);
    hi = make_uint2(__builtin_bit_cast(unsigned int, a0), __builtin_bit_cast(unsigned int, a1));
    lo = make_uint2(__builtin_bit_cast(unsigned int, b0), __builtin_bit_cast(unsigned int, b1));
}

// one k-step (K = 16) of a staged chunk: acc[i][j] += A(rows am0 + 32 i + 0..31) . B(rows bn0 + 32 j + 0..31)^T, three
// products each.  As / Bs: [2 planes][rows][32] fp16 (plane strides apl / H2_BPL).
template <int MI>
__device__ __forceinline__ void h2_kstep(const unsigned short* __restrict__ As, int apl, const unsigned short* __restrict__ Bs, int ks,
                                         int am0, int bn0, floatx16 (&acc)[MI][2]) {
    const int lane = threadIdx.x & 63;
    const int piece = (lane >> 5) + 2 * ks;
    half8 a[MI][2], b[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i][p] = *reinterpret_cast<const half8*>(As + p * apl + h2_off(am0 + 32 * i + (lane & 31), piece));
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j][p] = *reinterpret_cast<const half8*>(Bs + p * H2_BPL + h2_off(bn0 + 32 * j + (lane & 31), piece));
    }
    // smallest terms first; consecutive MFMAs go to different accumulators
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][PA[q]], b[j][PB[q]], acc[i][j], 0, 0, 0);
}

// Both k-steps of a staged chunk with every LDS fragment read issued up front (16 ds_read_b128, 64 VGPRs) and the MFMAs behind
// them in the order the fragments arrive: with ONE wave per SIMD (tile lists of about one workgroup per CU: the lanes of a C2
// batch, small batches) nothing else covers the four read -> wait -> multiply round trips per chunk of h2_kstep, 0.5 k of a
// chunk's 1.9 k cycles (profiles/r03_findings.md); with three waves per SIMD (MODE 0) the other waves do, and the registers
// are not there.
template <int MI>
__device__ __forceinline__ void h2_chunk_ahead(const unsigned short* __restrict__ As, int apl, const unsigned short* __restrict__ Bs,
                                               int am0, int bn0, floatx16 (&acc)[MI][2]) {
    const int lane = threadIdx.x & 63;
    half8 a[2][MI][2], b[2][2][2];                               // [k-step][tile][plane]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int piece = (lane >> 5) + 2 * ks;
        // in the order the products below consume them: (a lo, b hi), (a hi, b lo), (a hi, b hi)
#pragma unroll
        for (int i = 0; i < MI; ++i) a[ks][i][1] = *reinterpret_cast<const half8*>(As + apl + h2_off(am0 + 32 * i + (lane & 31), piece));
#pragma unroll
        for (int j = 0; j < 2; ++j) b[ks][j][0] = *reinterpret_cast<const half8*>(Bs + h2_off(bn0 + 32 * j + (lane & 31), piece));
#pragma unroll
        for (int i = 0; i < MI; ++i) a[ks][i][0] = *reinterpret_cast<const half8*>(As + h2_off(am0 + 32 * i + (lane & 31), piece));
#pragma unroll
        for (int j = 0; j < 2; ++j) b[ks][j][1] = *reinterpret_cast<const half8*>(Bs + H2_BPL + h2_off(bn0 + 32 * j + (lane & 31), piece));
    }
    __builtin_amdgcn_sched_barrier(0);                            // (reads first; hipcc would sink each next to its use)
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};           // smallest terms first; consecutive MFMAs go to different accumulators
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][i][PA[q]], b[ks][j][PB[q]], acc[i][j], 0, 0, 0);
}

// h2_chunk_ahead with the NEXT ring stage's direct-to-LDS loads issued BETWEEN its MFMA pairs (between(k), k = 0 .. 6 MI - 1).  A wave issues in
// order: six global_load_lds pieces in front of the chunk's reads cost their whole issue time (60-185 cycles each inside a busy phase) before the
// first MFMA can go -- 0.85 k cycles per chunk for 0.38 k of matrix work in the small-batch traces (profiles/r05_findings.md section 6).  Behind an
// MFMA pair the same issue slots are free: the pair occupies the matrix pipe for 64 cycles while the wave moves on.  Same reads, same MFMA order.
template <int MI, typename F>
__device__ __forceinline__ void h2_chunk_ahead_with(const unsigned short* __restrict__ As, int apl, const unsigned short* __restrict__ Bs,
                                                    int am0, int bn0, floatx16 (&acc)[MI][2], F&& between) {
    const int lane = threadIdx.x & 63;
    half8 a[2][MI][2], b[2][2][2];                               // [k-step][tile][plane]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int piece = (lane >> 5) + 2 * ks;
#pragma unroll
        for (int i = 0; i < MI; ++i) a[ks][i][1] = *reinterpret_cast<const half8*>(As + apl + h2_off(am0 + 32 * i + (lane & 31), piece));
#pragma unroll
        for (int j = 0; j < 2; ++j) b[ks][j][0] = *reinterpret_cast<const half8*>(Bs + h2_off(bn0 + 32 * j + (lane & 31), piece));
#pragma unroll
        for (int i = 0; i < MI; ++i) a[ks][i][0] = *reinterpret_cast<const half8*>(As + h2_off(am0 + 32 * i + (lane & 31), piece));
#pragma unroll
        for (int j = 0; j < 2; ++j) b[ks][j][1] = *reinterpret_cast<const half8*>(Bs + H2_BPL + h2_off(bn0 + 32 * j + (lane & 31), piece));
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][i][PA[q]], b[ks][j][PB[q]], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                between((ks * 3 + q) * MI + i);
                __builtin_amdgcn_sched_barrier(0);
            }
}

// A chunk whose A fragments are already in registers (MODE 5 of k_rowgemm_h2: loaded straight from global memory in the MFMA
// operand layout), B from the staged planes.  Same MFMA order per accumulator as h2_kstep / h2_chunk_ahead: bitwise the same sums.
template <int MI>
__device__ __forceinline__ void h2_chunk_regA(const half8 (&a)[2][MI][2] /*[k-step][tile][plane]*/, const unsigned short* __restrict__ Bs,
                                              int bn0, floatx16 (&acc)[MI][2]) {
    const int lane = threadIdx.x & 63;
    half8 b[2][2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int piece = (lane >> 5) + 2 * ks;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            b[ks][j][0] = *reinterpret_cast<const half8*>(Bs + h2_off(bn0 + 32 * j + (lane & 31), piece));
            b[ks][j][1] = *reinterpret_cast<const half8*>(Bs + H2_BPL + h2_off(bn0 + 32 * j + (lane & 31), piece));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][i][PA[q]], b[ks][j][PB[q]], acc[i][j], 0, 0, 0);
}

// Epilogue of the row GEMMs, one WAVE at a time and without workgroup barriers: the wave's 64 x 64 accumulators (MFMA
// layout: one column, 16 rows per lane) go through a wave-private LDS tile [32][H2_CW_LD] (row tile i = 0, 1) and are
// re-read as rows -- 8 lanes x 16 bytes per row segment -- so the base loads and U stores are 128-byte row segments and
// the maximum of a row's 64 columns stays inside 8 neighbouring lanes.  LDS operations of one wave execute in order, so the
// tile needs no barrier.
//   U[row0 + r, colw + c] = 2^-(sE[r] + w_exp) acc + base + tau;   umax[(row0 + r) * umax_ld + umax_col] = max_c |U|
// Round 3 (profiles/r03_findings.md): the first form of this function spent 8 k of the kernel's 32 k cycles in its store loop
// with NO memory traffic left in it (ablation builds without base loads and without U stores: the same 8 k).  gfx950 counts
// loads AND stores on vmcnt, and hipcc's wait insertion takes the minimum over all control-flow paths: with the base loads
// under a `base != nullptr` branch and the stores under `row < nrows`, every use of a prefetched base value came out as
// s_waitcnt vmcnt(0) -- i.e. each of the eight store groups waited for the previous group's stores to be acknowledged by
// memory -- and the row maxima went through three dependent ds_bpermute round trips per group.  Now: what is present is a
// template parameter (FWD), the base + time-term sums are formed once, BEFORE the first store (the only wait for loads), the
// store loop of a full tile is straight-line code, and the row maxima use DPP lane exchanges (no LDS).
constexpr int H2_CW_LD = 68;
constexpr int H2_CW_SZ = 32 * H2_CW_LD;      // floats per wave

// maximum over the 8 lanes that share lane >> 3 (fmaxf semantics: a NaN operand is skipped), by DPP: two quad permutations
// and a half-row mirror -- no LDS crossbar round trips (__shfl_xor compiles to ds_bpermute_b32)
__device__ __forceinline__ float h2_max8(float m) {
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, true)));    // quad_perm [1,0,3,2]
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x4E, 0xF, 0xF, true)));    // quad_perm [2,3,0,1]
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x141, 0xF, 0xF, true)));   // row_half_mirror
    return m;
}

// Loads the compiler must neither move nor wait for: inline asm (cdna_hip_programming.md 5.7).  hipcc sinks an ordinary load whose
// value is first used in the epilogue out of the K loop, past every sched_barrier, to that use (measured: all eighteen prefetch
// loads of this kernel ended up behind the last barrier -- and the counted waits of the loop, which assume them in the queue,
// then released a chunk whose operands were still in flight); and beside LDS-DMA loads it waits vmcnt(0) for any load it knows.
// An asm load is invisible to its wait insertion: the wait is h2_ld_wait* below, a statement that names every destination.
typedef float h2_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void h2_ld16(h2_f4& d, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
// the load of `base` (read once per evaluation, never reused inside it): -DCCSP_BASE_LOAD=1 non-temporal, =2 sc1 (experiment switch)
__device__ __forceinline__ void h2_ld16_base(h2_f4& d, const float* p) {
#if defined(CCSP_BASE_LOAD) && CCSP_BASE_LOAD == 1
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(d) : "v"(p) : "memory");
#elif defined(CCSP_BASE_LOAD) && CCSP_BASE_LOAD == 2
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(d) : "v"(p) : "memory");
#else
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
#endif
}
__device__ __forceinline__ void h2_ld4(int& d, const int* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
// s_waitcnt vmcnt(N) for the 8 base values of one row tile (+ the 2 time-term values), N a literal
template <int N>
__device__ __forceinline__ void h2_ld_wait(h2_f4 (&b)[4][2]) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]), "+v"(b[3][0]), "+v"(b[3][1]) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void h2_ld_wait2(h2_f4 (&t)[2]) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(t[0]), "+v"(t[1]) : "n"(N) : "memory"); }

// base values of row tile i of the wave, in the epilogue's row layout (lane: rows er + 8 st, columns 4 eq + 32 k); requested
// ahead of their use: 8 loads per lane.  No branches: rows past the tile's end read the tile's last row (never stored).
template <int ND>
__device__ __forceinline__ void h2_epilogue_prefetch(h2_f4 (&bs)[4][2], int i, int wrow0, int nrows, int row0, int colw,
                                                     const float* __restrict__ base) {
    const int lane = threadIdx.x & 63;
    const int er = lane >> 3, eq = lane & 7;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int trow = wrow0 + i * 32 + er + 8 * st;
        const int tr = trow < nrows ? trow : (nrows - 1 > 0 ? nrows - 1 : 0);
#pragma unroll
        for (int k = 0; k < 2; ++k) h2_ld16_base(bs[st][k], base + (size_t)(row0 + tr) * ND + colw + 4 * eq + 32 * k);
    }
}

// FWD: the forward GEMM (base, time term on slot-0 tiles, row maxima); otherwise plain U = 2^-e acc (the transpose GEMM).
// PRE1: the caller requested row tile 1's base values (bs1) itself, under the K loop; otherwise they are requested here.
// bs0 (and bs1) and the time-term values tv (requested in the kernel's prologue; has_tau false: discarded, slot-1 tiles) may
// still be in flight (h2_ld16): the waits are here.
// the row store of U (experiment switch: -DCCSP_U_STORE=1 non-temporal, =2 write-through sc1; default plain -- see profiles/r03_findings.md)
__device__ __forceinline__ void h2_store_u(float* p, const float4& v) {
#if defined(CCSP_U_STORE) && CCSP_U_STORE == 1
    __builtin_nontemporal_store(v.x, p); __builtin_nontemporal_store(v.y, p + 1); __builtin_nontemporal_store(v.z, p + 2); __builtin_nontemporal_store(v.w, p + 3);
#elif defined(CCSP_U_STORE) && CCSP_U_STORE == 2
    h2_f4 t{v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(t) : "memory");
#elif defined(CCSP_U_STORE) && CCSP_U_STORE == 3
    h2_f4 t{v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(t) : "memory");
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}

template <int ND, int MI, bool FWD, bool PRE1>
__device__ __forceinline__ void h2_epilogue_wave(const floatx16 (&acc)[MI][2], h2_f4 (&bs0)[4][2], h2_f4 (&bs1)[4][2], h2_f4 (&tv)[2], float* __restrict__ Cw,
                                                 int wrow0 /*first tile row of the wave*/, int nrows, int row0,
                                                 int colw /*first global column of the wave*/, const int* __restrict__ sE, int w_exp,
                                                 const float* __restrict__ base, bool has_tau,
                                                 float* __restrict__ U, float* __restrict__ umax, int umax_ld, int umax_col) {
    const int lane = threadIdx.x & 63;
    const int er = lane >> 3, eq = lane & 7;                      // rows er + 8 s (s < 4) of a 32-row tile, columns 4 eq + 32 k (k < 2)
    if constexpr (FWD) {
        if constexpr (MI == 2 && !PRE1) h2_epilogue_prefetch<ND>(bs1, 1, wrow0, nrows, row0, colw, base);
        // bs0 (older than tv), tv -- and bs1 when it was requested under the K loop -- have landed; a bs1 requested just now
        // stays in flight while row tile 0 is processed
        if constexpr (MI == 2 && !PRE1) { h2_ld_wait<8>(bs0); h2_ld_wait2<8>(tv); }
        else { h2_ld_wait<0>(bs0); h2_ld_wait2<0>(tv); if constexpr (MI == 2) h2_ld_wait<0>(bs1); }
        if (!has_tau) { tv[0] = h2_f4{0.f, 0.f, 0.f, 0.f}; tv[1] = h2_f4{0.f, 0.f, 0.f, 0.f}; }
    }
    if (wrow0 >= nrows) {                                         // (wave-uniform) nothing of the tile in this wave's rows
        if constexpr (FWD && MI == 2 && !PRE1) h2_ld_wait<0>(bs1);        // (its registers must not be reused under the loads)
        return;
    }
    int ex[MI][4];                                                // -(row exponent + weight exponent) of this lane's rows
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int st = 0; st < 4; ++st) ex[i][st] = -(sE[wrow0 + i * 32 + er + 8 * st] + w_exp);
    const bool full = wrow0 + 32 * MI <= nrows;                   // (wave-uniform) every row of the wave's tiles exists: straight-line stores
    float* const Ub = U + (size_t)row0 * ND + colw + 4 * eq;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                Cw[rr * H2_CW_LD + j * 32 + (lane & 31)] = acc[i][j][r];
            }
        asm volatile("" ::: "memory");                            // (compiler ordering only: the LDS runs one wave's operations in order)
        CCSP_TRK(0, 13 + 2 * i);
        if constexpr (FWD && MI == 2 && !PRE1) { if (i == 1) h2_ld_wait<0>(bs1); }      // (also drains row tile 0's stores: once per wave)
        h2_f4 bt[4][2];                                           // base + time term of the tile's rows, before its first store
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if constexpr (FWD) bt[st][k] = (i == 0 ? bs0[st][k] : bs1[st][k]) + tv[k];
                else bt[st][k] = h2_f4{0.f, 0.f, 0.f, 0.f};
            }
        float4 cv[4][2];                                          // the lane's 4 x 2 row segments: all LDS reads in flight together
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int k = 0; k < 2; ++k) cv[st][k] = *reinterpret_cast<const float4*>(Cw + (er + 8 * st) * H2_CW_LD + 4 * eq + 32 * k);
        auto rows = [&](auto all_rows) {
            constexpr bool ALL = decltype(all_rows)::value;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int trow = wrow0 + i * 32 + er + 8 * st;
                const int e = ex[i][st];
                float4 o[2];
                float m = 0.0f;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float4 v = cv[st][k];
                    o[k].x = ldexpf(v.x, e) + bt[st][k][0];
                    o[k].y = ldexpf(v.y, e) + bt[st][k][1];
                    o[k].z = ldexpf(v.z, e) + bt[st][k][2];
                    o[k].w = ldexpf(v.w, e) + bt[st][k][3];
                    if constexpr (FWD) m = fmaxf(fmaxf(m, fmaxf(fabsf(o[k].x), fabsf(o[k].y))), fmaxf(fabsf(o[k].z), fabsf(o[k].w)));
                }
                if constexpr (FWD) m = h2_max8(m);                // (every lane takes part: the 8 lanes of a row are active together)
                if (ALL || trow < nrows) {
                    float* up = Ub + (size_t)trow * ND;
                    h2_store_u(up, o[0]);
                    h2_store_u(up + 32, o[1]);
                    if constexpr (FWD) { if (eq == 0) umax[(size_t)(row0 + trow) * umax_ld + umax_col] = m; }
                }
            }
        };
        if (full) rows(std::true_type{}); else rows(std::false_type{});
        asm volatile("" ::: "memory");
        CCSP_TRK(0, 14 + 2 * i);
    }
}

// s_waitcnt vmcnt(n) lgkmcnt(0) with n known after unrolling (the immediate must be a literal)
__device__ __forceinline__ void h2_wait_vm_lgkm0(int n) {
    if (n >= 28) asm volatile("s_waitcnt vmcnt(28) lgkmcnt(0)" ::: "memory");
    else if (n >= 24) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
    else if (n >= 20) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");
    else if (n >= 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    else if (n >= 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (n >= 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// s_waitcnt vmcnt(N) for the row indices requested at kernel entry (h2_ld4), N = loads issued since
template <int N, int NI>
__device__ __forceinline__ void h2_idx_wait(int (&v)[NI]) {
    static_assert(NI == 1 || NI == 2 || NI == 4, "index count");
    if constexpr (NI == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v[0]) : "n"(N) : "memory");
    else if constexpr (NI == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(v[0]), "+v"(v[1]) : "n"(N) : "memory");
    else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "n"(N) : "memory");
}

// ------------------------------------------------------------------------------------------
// k_rowgemm_h2<KD, ND, MODE>:  U[row0 + r, col0 + c] = 2^-(ea[r] + ew) * sum_k A[src(r), k] W[ts][col0 + c, k]  + base + tau
//   A planes [2][n_src][KD] fp16 bits scaled by 2^a_exp[src], rows gathered by urow_node; W planes [2][n_ts][ND][KD]
//   scaled by 2^w_exp.  umax[row, (col0 + 64 wn) / 64] = max |U| over 64 columns (forward GEMM only).
//   tile_desc[tile] = {first row, rows, 2 type + slot, -}: one scalar load.
//   The kernel is one latency chain per tile (descriptor -> row indices -> operands -> chunks -> epilogue) and a launch
//   is as long as the chains it runs one after the other on a CU slot, so the residency is chosen per launch:
//   MODE 0: one stage + one register set, 33 KB and <= 168 VGPRs, 3 workgroups per CU -- the whole tile list of a
//           C2-sized batch (640 tiles) is resident at once instead of running as a full and a 20 %-full round;
//   MODE 1: two LDS stages + two register sets (chunk c+2 in flight), 64.5 KB, 2 workgroups per CU;
//   MODE 2: two LDS stages filled by direct-to-LDS loads (global_load_lds_dwordx4: no staging registers and no
//           ds_write -- the 16-byte stores were half of the kernel's LDS-instruction cycles).  A wave-instruction
//           writes 64 lanes x 16 bytes = sixteen consecutive 64-byte rows of one plane, lane-linear, so the XOR
//           swizzle of the stage is applied on the SOURCE side: the lane in physical slot s of row r fetches logical
//           piece s ^ ((r >> 2) & 3).  Bare s_barrier + counted s_waitcnt: the base values of BOTH row tiles are requested
//           two / one chunks before the K loop ends and stay in flight across the barriers (a __syncthreads() would drain
//           them: the fabric-side latency of `base` used to sit in the last chunk's barrier and in the epilogue);
//   MODE 3 / 4: see below.
//   In every mode the weight operand of chunk 0 is requested BEFORE the row indices arrive (its address needs the tile
//   descriptor only), the A rows behind it, and the row exponents (a second dependent gather) last: they are needed by the
//   epilogue only.
// ------------------------------------------------------------------------------------------
// (the trace build's stamps push MODE 0 over its 168-register budget; a spill next to the inline-asm loads -- whose destinations the
// compiler believes defined the moment they are issued -- is not survivable, so that build runs MODE 0 at two workgroups per CU)
#ifdef CCSP_TRACE
#define CCSP_H2_MODE0_WGS 2
#else
#define CCSP_H2_MODE0_WGS 3
#endif
template <int KD, int ND, int MODE>
__global__ __launch_bounds__(256, MODE == 0 ? CCSP_H2_MODE0_WGS : (MODE == 6 ? 4 : (MODE == 3 ? 1 : 2))) void k_rowgemm_h2(const unsigned short* __restrict__ A, size_t a_plane, const int* __restrict__ a_exp,
                                                       const int* __restrict__ urow_node, const int4* __restrict__ tile_desc,
                                                       const unsigned short* __restrict__ W, size_t w_plane, size_t w_stride, int w_exp,
                                                       const float* __restrict__ base, const float* __restrict__ tau_t,
                                                       float* __restrict__ U, float* __restrict__ umax, StepRef ref, size_t tau_stride) {
    static_assert(ND % 128 == 0 && KD % H2_BK == 0 && KD / H2_BK >= 4, "shape");
    constexpr bool FWD = KD < ND;                                 // <256, 512>: the forward GEMM (base, time term, row maxima); <512, 256>: the transpose
    // Operand layout of the FORWARD GEMM's weights (round 5): the two fp16 planes of a row's 32-element K chunk side by side -- [row][chunk][plane][32],
    // 128 bytes = ONE L2 -> L1 line per (row, chunk).  In the planar layout ([plane][row][K]) a chunk of a row is HALF a line of each plane, the
    // other halves belong to the next chunk, and by then (32 KB of other lines per workgroup and chunk through a 32 KB L1) they are gone: every
    // operand line crossed the CU's L2 port twice.  WROW / WCH / WPL: element strides of a row, a chunk, a plane.
    constexpr bool ILW = true;                                    // (the transpose GEMM's weights too: WpTHI)
    constexpr int WROW = ILW ? 2 * KD : KD, WCH = ILW ? 2 * H2_BK : H2_BK;
    const size_t WPL = ILW ? (size_t)H2_BK : w_plane, WTS = ILW ? 2 * w_stride : w_stride;
    // ... and of its A operand, the pose-embedding planes the node kernels' encoder epilogue writes (enc_store_tile; CCSP_A_INTERLEAVED: the product
    // build -- the experiments build keeps the planar planes its other consumers read)
    constexpr bool ILA = CCSP_A_INTERLEAVED;                      // (transpose GEMM: the partial-row planes k_edge_bwd_h2 writes)
    constexpr int AROW = ILA ? 2 * KD : KD, ACH = ILA ? 2 * H2_BK : H2_BK;
    const size_t APLN = ILA ? (size_t)H2_BK : a_plane;
    CCSP_TRK2_DECL
    CCSP_TRK2(0);
    CCSP_TRK(0, 0);
    CCSP_TRK_RT(0, 30);
    if (ref.skip && *ref.skip == 0) return;                       // (uniform) MALA reuse: the state has not moved since this was computed
    gate_wait(ref.gate);
    const int n_work = (int)gridDim.x - ref.na.blocks;
    if ((int)blockIdx.x >= n_work) {                              // (workgroup-uniform) the evaluation's normal draws: NoiseAhead
        noise_ahead_block(ref.na, (int)blockIdx.x - n_work);
        return;
    }
    constexpr int NCT = ND / 128, NCH = KD / H2_BK;
    constexpr int MI = (MODE == 4 || MODE == 6) ? 1 : 2, TM = 64 * MI;           // 32-row MFMA tiles per wave, rows per workgroup tile (MODE 6: MODE 0's staging on 64-row tiles)
    constexpr int APL = TM * H2_BK, STAGE = MODE == 5 ? 2 * H2_BPL : 2 * APL + 2 * H2_BPL;   // 32 KB per stage (24 KB for 64-row tiles; MODE 5: the B planes only, 16 KB)
    constexpr bool DB = MODE == 1;
    constexpr bool DBP = MODE == 9;                               // two stages + two register sets like MODE 1, the staging under the MFMAs (below)
    constexpr int NST = (MODE == 0 || MODE == 6) ? 1 : ((MODE == 3 || MODE == 5) ? 4 : (MODE == 4 ? 3 : 2));      // LDS stages
    constexpr int NRS = (MODE == 1 || MODE == 9) ? 2 : 1;                        // register sets (MODE 2 and above use none)
    constexpr bool PRE1 = FWD && MI == 2 && MODE >= 2 && MODE != 6 && MODE != 9;            // row tile 1's base values requested under the K loop (VGPRs to spare)
    constexpr int SMEM_US = (NST * STAGE * 2 > 4 * H2_CW_SZ * 4 ? NST * STAGE : 4 * H2_CW_SZ * 2);
    __shared__ __attribute__((aligned(16))) unsigned short smem[SMEM_US + 256];      // stages (epilogue tiles on top) + 128 row exponents
    int* sE = reinterpret_cast<int*>(smem + SMEM_US);
    if (ref.tab) tau_t += (size_t)ref.tab[*ref.counter].t * tau_stride;      // hipGraph mode: timestep from the device table
    const int bid = xcd_remap(blockIdx.x, n_work);
    const int tile = bid / NCT, ct = bid % NCT;
    // Forward GEMM: the plane rows of this thread's tile rows come from a table padded per tile (StepRef::tile_rows): its address needs the
    // workgroup index only, so the gather is requested HERE, next to the tile descriptor, and not behind it -- descriptor -> row indices ->
    // operand rows was three dependent round trips of ~2 k cycles each in front of the first MFMA (profiles/r05_findings.md section 1), now two.
    // asm loads (the compiler neither moves nor waits for them); the wait is h2_idx_wait<N> in each mode, N = the loads issued behind them.
    constexpr bool RINGIDX = MODE >= 2 && MODE != 5 && MODE != 6 && MODE != 9;       // the direct-to-LDS forms: a lane's rows are those of its 1 KB blocks
    constexpr int TM_ = ((MODE == 4 || MODE == 6) ? 1 : 2) * 64;
    constexpr int NIDX = RINGIDX ? TM_ / 32 : TM_ / 64;
    int srcx[NIDX];
    if constexpr (KD < ND) {
        const int* trp = ref.tile_rows + (size_t)tile * TM_;
#pragma unroll
        for (int j = 0; j < NIDX; ++j) {
            int row;
            if constexpr (RINGIDX) row = (((TM_ / 32) * (int)(threadIdx.x >> 6) + j) % (TM_ / 16)) * 16 + (int)((threadIdx.x & 63) >> 2);
            else if constexpr (MODE == 5) row = (int)((threadIdx.x >> 7) * 64) + 32 * j + (int)(threadIdx.x & 31);
            else row = (int)(threadIdx.x >> 2) + 64 * j;
            h2_ld4(srcx[j], trp + row);
        }
    }
    const int4 td = tile_desc[tile];
    const int row0 = td.x, nrows = td.y, ts = td.z;
    const int col0 = ct * 128;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = tid >> 2, lq = tid & 3;                      // staging: rows lrow, lrow + 64, 16-byte piece lq, both planes
    floatx16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    h2_f4 bs0[4][2], bs1[4][2];
    const int wr0 = wm * 32 * MI;                                 // first tile row of the wave
    const int colw = col0 + wn * 64;
    // the time term of the wave's columns (slot-0 tiles of the forward GEMM; other tiles read the same bytes of `base` and
    // discard them: a select instead of a branch).  The first loads of the kernel: older than every counted wait below.
    const bool has_tau = FWD && tau_t && (ts & 1) == 0;
    h2_f4 tv[2] = {h2_f4{0.f, 0.f, 0.f, 0.f}, h2_f4{0.f, 0.f, 0.f, 0.f}};
    // The counted waits for the row indices requested at kernel entry (h2_idx_wait<N>) name how many vector loads are issued BEHIND the indices
    // before the wait: the constants below sit next to the loads they count, and the waits are written in terms of them -- a load added or removed
    // here changes the wait with it instead of silently letting it return before the indices have arrived.
    constexpr int N_TV = 2;                                       // the two time-term loads right below (FWD only; always issued: a select, not a branch, picks tau or base)
    constexpr int N_GLDS_B = 4;                                   // 16-byte pieces per lane of one weight chunk in the direct-to-LDS forms (glds_b)
    if constexpr (FWD) {
        const float* tp = (has_tau ? tau_t + (size_t)(ts >> 1) * ND : base) + colw + 4 * (lane & 7);
        h2_ld16(tv[0], tp);
        h2_ld16(tv[1], tp + 32);
        static_assert(N_TV == 2, "N_TV counts the h2_ld16(tv[...]) loads above");
    }
    if constexpr (MODE == 5) {
        // MODE 5 (tile lists of at most two workgroups per CU: the lanes of a C2 batch).  The K loop of the other modes is bound by
        // the latency of the operand loads, not by the MFMAs (phase traces: 2.3 k cycles per chunk against 0.77 k of matrix work,
        // with one chunk requested ahead) and a deeper ring of (A, B) stages does not fit two workgroups into a CU's LDS.  Here
        // only the WEIGHT planes are staged -- a ring of four 16 KB stages filled by global_load_lds, three chunks requested ahead --
        // and every lane loads its A fragments (rows gathered by node) straight from global memory in the MFMA operand layout,
        // two chunks ahead, into two register sets.  Loads retire in order, so the waits are counted.
        using gptr = const __attribute__((address_space(1))) void*;
        using lptr = __attribute__((address_space(3))) void*;
        const unsigned short* gb[4];
        int lob[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int blk = 4 * wave + j, plane = blk >> 3, rb16 = blk & 7;
            const int row = rb16 * 16 + (lane >> 2);
            const int piece = (lane & 3) ^ ((row >> 2) & 3);
            gb[j] = W + (size_t)plane * WPL + (size_t)ts * WTS + (size_t)(col0 + row) * WROW + piece * 8;
            lob[j] = __builtin_amdgcn_readfirstlane(plane * H2_BPL + rb16 * 16 * H2_BK);
        }
        auto glds_b = [&](int c) {
            unsigned short* st = smem + (c % NST) * STAGE;
#pragma unroll
            for (int j = 0; j < N_GLDS_B; ++j) __builtin_amdgcn_global_load_lds((gptr)(gb[j] + c * WCH), (lptr)(st + lob[j]), 16, 0, 0);
        };
        // the lane's two A rows (tile i = 0, 1): one dependent gather, requested first
        int src[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wr0 + 32 * i + (lane & 31);
            const int r = row < nrows ? row : nrows - 1;
            if constexpr (FWD) src[i] = srcx[i];                     // (requested at kernel entry)
            else if (urow_node) h2_ld4(src[i], urow_node + row0 + r); else src[i] = row0 + r;
        }
        __builtin_amdgcn_sched_barrier(0);
        glds_b(0);
        glds_b(1);
        glds_b(2);
        __builtin_amdgcn_sched_barrier(0);
        if (FWD || urow_node) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(src[0]), "+v"(src[1]) : "n"(3 * N_GLDS_B) : "memory");      // (the three chunks of weight loads issued since stay in flight)
        const unsigned short* ap[2][2];                               // [tile][plane]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) ap[i][pl] = A + (size_t)pl * APLN + (size_t)src[i] * AROW + 8 * (lane >> 5);
        half8 af[2][2][2][2];                                         // [register set][k-step][tile][plane]
        auto aload = [&](int c, int set) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(af[set][ks][i][pl]) : "v"(ap[i][pl] + c * ACH + 16 * ks) : "memory");
        };
        auto await_set = [&](int set) {                               // names the set's registers as read-write: the compiler keeps its uses behind the wait
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) asm volatile("" : "+v"(af[set][ks][i][pl]) :: "memory");
        };
        aload(0, 0);
        aload(1, 1);
        int ea[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) h2_ld4(ea[i], a_exp + src[i]);
        __builtin_amdgcn_sched_barrier(0);
        CCSP_TRK(0, 1);
        constexpr int CB0 = NCH - 3, CB1 = NCH - 2;                   // iterations under which the base values of row tile 0 / 1 are requested
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            // in flight behind (A(c), B(c)) at this point -- everything issued after A(c):
            //   c = 0: A(1), the two exponent loads;  c = 1: the exponents, B(3), A(2);  c >= 2: B(c + 2), the base values of iteration c - 1, A(c + 1)
            int younger;
            if (c == 0) younger = 8 + 2;
            else if (c == 1) younger = 2 + (3 < NCH ? 4 : 0) + (2 < NCH ? 8 : 0);
            else younger = (c + 2 < NCH ? 4 : 0) + ((FWD && (c - 1 == CB0 || c - 1 == CB1)) ? 8 : 0) + (c + 1 < NCH ? 8 : 0);
            __builtin_amdgcn_sched_barrier(0);
            h2_wait_vm_lgkm0(younger);
            await_set(c & 1);
            __builtin_amdgcn_s_barrier();                             // B(c) has landed for every wave; every wave is done reading stage (c - 1) % NST
            __builtin_amdgcn_sched_barrier(0);
            CCSP_TRK(0, 2 + (c < 8 ? c : 7));
            if (c + 3 < NCH) glds_b(c + 3);
            __builtin_amdgcn_sched_barrier(0);
            h2_chunk_regA<MI>(af[c & 1], smem + (c % NST) * STAGE, wn * 64, acc);
            __builtin_amdgcn_sched_barrier(0);                        // (the MFMAs have read the register set: it can be refilled)
            if constexpr (FWD) {
                if (c == CB0) h2_epilogue_prefetch<ND>(bs0, 0, wr0, nrows, row0, colw, base);
                if (c == CB1) h2_epilogue_prefetch<ND>(bs1, 1, wr0, nrows, row0, colw, base);
            }
            if (c + 2 < NCH) aload(c + 2, c & 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ea[0]), "+v"(ea[1]) :: "memory");      // (bs0 / bs1 / tv with them; the epilogue's own waits then pass)
        if (wn == 0 && lane < 32) { sE[wr0 + lane] = ea[0]; sE[wr0 + 32 + lane] = ea[1]; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                 // every wave is done reading the stages; the row exponents are visible
        __builtin_amdgcn_sched_barrier(0);
    } else if constexpr (MODE >= 2 && MODE != 6 && MODE != 9) {
        // a wave-instruction fills one 1 KB block = (plane, sixteen rows); wave w owns blocks NA w .. NA w + NA - 1 of the A
        // planes (2 x TM / 16 blocks) and 4 w .. 4 w + 3 of the B planes
        using gptr = const __attribute__((address_space(1))) void*;
        using lptr = __attribute__((address_space(3))) void*;
        constexpr int NA = TM / 32, ARB = TM / 16;
        const unsigned short* ga[NA];
        const unsigned short* gb[4];
        int loa[NA], lob[4];                                      // wave-uniform stage offsets of the blocks (fp16 elements)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int blk = 4 * wave + j, plane = blk >> 3, rb16 = blk & 7;
            const int row = rb16 * 16 + (lane >> 2);
            const int piece = (lane & 3) ^ ((row >> 2) & 3);
            gb[j] = W + (size_t)plane * WPL + (size_t)ts * WTS + (size_t)(col0 + row) * WROW + piece * 8;
            lob[j] = __builtin_amdgcn_readfirstlane(plane * H2_BPL + rb16 * 16 * H2_BK);
        }
        auto glds_b = [&](int c, int stage) {
            unsigned short* st = smem + stage * STAGE;
#pragma unroll
            for (int j = 0; j < N_GLDS_B; ++j) __builtin_amdgcn_global_load_lds((gptr)(gb[j] + c * WCH), (lptr)(st + 2 * APL + lob[j]), 16, 0, 0);
        };
        auto glds_a = [&](int c, int stage) {
            unsigned short* st = smem + stage * STAGE;
#pragma unroll
            for (int j = 0; j < NA; ++j) __builtin_amdgcn_global_load_lds((gptr)(ga[j] + c * ACH), (lptr)(st + loa[j]), 16, 0, 0);
        };
        auto glds = [&](int c, int stage) { glds_a(c, stage); glds_b(c, stage); };
        int src[NA];
#pragma unroll
        for (int j = 0; j < NA; ++j) {                            // the A rows: one dependent gather (row index -> plane row)
            const int blk = NA * wave + j, rb16 = blk % ARB;
            const int row = rb16 * 16 + (lane >> 2);
            const int r = row < nrows ? row : nrows - 1;
            if constexpr (!FWD) src[j] = urow_node ? urow_node[row0 + r] : row0 + r;
        }
        __builtin_amdgcn_sched_barrier(0);
        glds_b(0, 0);                                             // the weights of chunk 0 are on their way while the row indices arrive
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FWD) {                                      // (requested at kernel entry; behind them: the two time-term loads and the four weight pieces)
            h2_idx_wait<N_TV + N_GLDS_B>(srcx);
#pragma unroll
            for (int j = 0; j < NA; ++j) src[j] = srcx[j];
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int blk = NA * wave + j, plane = blk / ARB, rb16 = blk % ARB;
            const int row = rb16 * 16 + (lane >> 2);
            const int piece = (lane & 3) ^ ((row >> 2) & 3);
            ga[j] = A + (size_t)plane * APLN + (size_t)src[j] * AROW + piece * 8;
            loa[j] = __builtin_amdgcn_readfirstlane(plane * APL + rb16 * 16 * H2_BK);
        }
        // the row exponents (epilogue only) come from the lanes that hold a row's plane index: no second dependent gather
        int ea[NA];
        auto exps_load = [&]() {                                      // (asm loads: pinned where they are issued, see h2_ld16)
#pragma unroll
            for (int j = 0; j < NA; ++j) h2_ld4(ea[j], a_exp + src[j]);
        };
        auto exps_wait = [&]() {
#pragma unroll
            for (int j = 0; j < NA; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ea[j]) :: "memory");
        };
        auto exps_store = [&]() {
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const int blk = NA * wave + j;
                if (blk / ARB == 0 && (lane & 3) == 0) sE[(blk % ARB) * 16 + (lane >> 2)] = ea[j];
            }
        };
        CCSP_TRK(0, 1);
        constexpr int NL = NA + 4;                                // loads per chunk and thread
        if constexpr (MODE >= 3) {
            // Low-latency forms for short tile lists (small batches: the kernel is one dependent chain, not a throughput
            // problem): a ring of NST stages with NST - 1 chunks of operands in flight and COUNTED waits -- s_waitcnt vmcnt(n)
            // lets the loads of the younger chunks stay outstanding while chunk c is multiplied (vector-memory loads retire
            // in order).  The barriers are bare s_barrier: __syncthreads() carries a fence that drains every load.  `base` is
            // requested right behind the first chunk's operands (older than the rest of the ring, so it has landed early).
            //   MODE 3: 128-row tiles, four stages;  MODE 4: 64-row tiles (32 x 64 per wave), three stages, 72 KB -- a quarter of
            //   the MFMA and epilogue work per wave and twice the workgroups, for tile lists that leave most CUs empty.
            constexpr int D = NST - 1;                            // chunks requested ahead
            constexpr int NB = FWD ? 8 * MI : 0;                  // base loads per thread, issued between chunk 0 and chunk 1 of the ring
            glds_a(0, 0);
            __builtin_amdgcn_sched_barrier(0);                        // (the counted waits below rely on this issue order)
            exps_load();
            if constexpr (FWD) {
                h2_epilogue_prefetch<ND>(bs0, 0, wr0, nrows, row0, colw, base);
                if constexpr (PRE1) h2_epilogue_prefetch<ND>(bs1, 1, wr0, nrows, row0, colw, base);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 1; c < D; ++c) glds(c, c);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                // (lgkmcnt(0): this wave's LDS reads of the stage about to be refilled have returned, and its sE write)
                // younger than chunk c at this point: chunks c+1 .. c+D-1 (and, for c == 0, the base loads behind chunk 0)
                __builtin_amdgcn_sched_barrier(0);                    // (the MFMAs of chunk c - 1 stay in front of the wait: hipcc sinks register-only work past it)
                h2_wait_vm_lgkm0((NCH - 1 - c < D - 1 ? NCH - 1 - c : D - 1) * NL + (c == 0 ? NB + NA : 0));
                __builtin_amdgcn_s_barrier();                         // chunk c has landed for every wave; stage (c-1) % NST is free
                __builtin_amdgcn_sched_barrier(0);
                CCSP_TRK(0, 2 + (c < 8 ? c : 7));
                const unsigned short* st = smem + (c % NST) * STAGE;
                if (c + D < NCH) {
                    // the next ring stage's loads go out between this chunk's MFMA pairs (h2_chunk_ahead_with): A pieces first, then the weights
                    unsigned short* nst = smem + ((c + D) % NST) * STAGE;
                    h2_chunk_ahead_with<MI>(st, APL, st + 2 * APL, wr0, wn * 64, acc, [&](int k) {
                        if (k < NA) __builtin_amdgcn_global_load_lds((gptr)(ga[k] + (c + D) * ACH), (lptr)(nst + loa[k]), 16, 0, 0);
                        else if (k < NA + 4) __builtin_amdgcn_global_load_lds((gptr)(gb[k - NA] + (c + D) * WCH), (lptr)(nst + 2 * APL + lob[k - NA]), 16, 0, 0);
                    });
                } else {
                    h2_chunk_ahead<MI>(st, APL, st + 2 * APL, wr0, wn * 64, acc);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            exps_wait();
            exps_store();                                             // (written here: a use of the loaded exponents in front of the ring would drain it)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                             // every wave is done reading the stages; the row exponents are visible
            __builtin_amdgcn_sched_barrier(0);
        } else {
            glds_a(0, 0);
            exps_load();
            __builtin_amdgcn_sched_barrier(0);
            exps_wait();
            exps_store();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#ifndef CCSP_H2_CB0
#define CCSP_H2_CB0 3
#define CCSP_H2_CB1 2
#endif
            constexpr int CB0 = NCH - CCSP_H2_CB0, CB1 = NCH - CCSP_H2_CB1;       // chunks under which the base values of row tile 0 / 1 are requested
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                CCSP_TRK(0, 2 + (c < 8 ? c : 7));
                if (c + 1 < NCH) glds(c + 1, (c + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);                    // (the counted wait below relies on this issue order)
                int younger = 0;                                      // loads of this iteration that are younger than chunk c + 1's
                if constexpr (FWD) {
                    if (c == CB0) { h2_epilogue_prefetch<ND>(bs0, 0, wr0, nrows, row0, colw, base); younger += 8; }
                    if (c == CB1) { h2_epilogue_prefetch<ND>(bs1, 1, wr0, nrows, row0, colw, base); younger += 8; }
                }
                __builtin_amdgcn_sched_barrier(0);
                const unsigned short* st = smem + (c & 1) * STAGE;
                h2_chunk_ahead<MI>(st, APL, st + 2 * APL, wr0, wn * 64, acc);
                // chunk c + 1 has landed (the base loads behind it may still be in flight); this wave is done reading stage c & 1
                __builtin_amdgcn_sched_barrier(0);                    // (the MFMAs above stay in front of the wait: hipcc sinks register-only work past it)
                if (c + 1 < NCH) h2_wait_vm_lgkm0(younger);
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        const unsigned short* a_ptr[MI];
        int srcr[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            int r = lrow + 64 * i;
            r = r < nrows ? r : nrows - 1;
            if constexpr (FWD) { if (i == 0) h2_idx_wait<N_TV>(srcx); srcr[i] = srcx[i]; }      // (requested at kernel entry; behind them: the two time-term loads)
            else srcr[i] = urow_node ? urow_node[row0 + r] : row0 + r;
            a_ptr[i] = A + (size_t)srcr[i] * AROW + lq * 8;
        }
        auto row_exps = [&]() {                                   // (epilogue only) behind the first operands, by the lanes that hold the row's index
            int e[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) e[i] = a_exp[srcr[i]];
            if (lq == 0) {
#pragma unroll
                for (int i = 0; i < MI; ++i) sE[lrow + 64 * i] = e[i];
            }
        };
        const unsigned short* b_ptr = W + (size_t)ts * WTS + (size_t)(col0 + lrow) * WROW + lq * 8;
        const int st_off = h2_off(lrow, lq);                      // (row + 64 has the same swizzle: + 64 * H2_BK)
        ushort8 ra[NRS][2 * MI], rb[NRS][4];                      // [register set][row half * 2 + plane]
        auto gload = [&](int c, int set) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    rb[set][i * 2 + p] = *reinterpret_cast<const ushort8*>(b_ptr + (size_t)i * 64 * WROW + (size_t)p * WPL + c * WCH);
                    if (i < MI) ra[set][(i < MI ? i : 0) * 2 + p] = *reinterpret_cast<const ushort8*>(a_ptr[i < MI ? i : 0] + (size_t)p * APLN + c * ACH);
                }
        };
        auto lstore = [&](int stage, int set) {
            unsigned short* As = smem + stage * STAGE;
            unsigned short* Bs = As + 2 * APL;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    if (i < MI) *reinterpret_cast<ushort8*>(As + p * APL + st_off + i * 64 * H2_BK) = ra[set][(i < MI ? i : 0) * 2 + p];
                    *reinterpret_cast<ushort8*>(Bs + p * H2_BPL + st_off + i * 64 * H2_BK) = rb[set][i * 2 + p];
                }
        };
        CCSP_TRK(0, 1);
        if constexpr (DBP) {
            // MODE 9 (round 5; tile lists of at most two workgroups per CU -- the lanes of a C2 batch).  With ONE wave per SIMD nothing overlaps a
            // wave's LDS traffic with its MFMAs: per chunk a wave reads 16 KB of fragments and stores 8 KB of staging (~ 930 cycles of the CU's LDS
            // for its four waves) and multiplies for 768 cycles, one after the other -- the 2.5 k-cycle chunk period of every earlier form, however
            // its pieces were arranged (profiles/r05_findings.md sections 1, 2, 7).  Here the wave is its own second wave: two fragment sets, the
            // reads of the NEXT k-step always in flight under the twelve MFMAs of the current one; two LDS stages and two register sets, the
            // eight staging stores of chunk c + 1 (into the other stage) and the eight global loads of chunk c + 3 issued ONE per MFMA pair; one
            // bare barrier per chunk, in its MIDDLE (behind k-step 0's MFMAs: the other stage is then visible for the reads of chunk c + 1's first
            // k-step, which go out under k-step 1's MFMAs).  Same MFMA order per accumulator as h2_kstep: bitwise the same sums.  248 VGPRs.
            static_assert(MI == 2, "MODE 9 runs 128-row tiles");
            auto gload_one = [&](int c, int set, int k) {             // k: bit 0 plane, bit 1 row half, bit 2 weight / A row
                const int pl = k & 1, i = (k >> 1) & 1;
                if (k & 4) rb[set][i * 2 + pl] = *reinterpret_cast<const ushort8*>(b_ptr + (size_t)i * 64 * WROW + (size_t)pl * WPL + c * WCH);
                else ra[set][i * 2 + pl] = *reinterpret_cast<const ushort8*>(a_ptr[i] + (size_t)pl * APLN + c * ACH);
            };
            auto lstore_one = [&](int stage, int set, int k) {
                const int pl = k & 1, i = (k >> 1) & 1;
                unsigned short* As = smem + stage * STAGE;
                if (k & 4) *reinterpret_cast<ushort8*>(As + 2 * APL + pl * H2_BPL + st_off + i * 64 * H2_BK) = rb[set][i * 2 + pl];
                else *reinterpret_cast<ushort8*>(As + pl * APL + st_off + i * 64 * H2_BK) = ra[set][i * 2 + pl];
            };
            const int fr = lane & 31, fp0 = lane >> 5;
            half8 fa[2][2][2], fb[2][2][2];                           // [fragment set = k-step][tile][plane: 0 hi, 1 lo]
            auto fread = [&](const unsigned short* st, int ks) {      // in the order the products consume them: (a lo, b hi), (a hi, b lo)
                const unsigned short* Bs = st + 2 * APL;
                const int piece = fp0 + 2 * ks;
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[ks][i][1] = *reinterpret_cast<const half8*>(st + APL + h2_off(wr0 + 32 * i + fr, piece));
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[ks][j][0] = *reinterpret_cast<const half8*>(Bs + h2_off(wn * 64 + 32 * j + fr, piece));
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[ks][i][0] = *reinterpret_cast<const half8*>(st + h2_off(wr0 + 32 * i + fr, piece));
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[ks][j][1] = *reinterpret_cast<const half8*>(Bs + H2_BPL + h2_off(wn * 64 + 32 * j + fr, piece));
            };
            gload(0, 0);
            gload(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            row_exps();
            lstore(0, 0);
            gload(2, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            fread(smem, 0);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};       // smallest terms first (h2_kstep's order)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {                           // fully unrolled: register-set indices are constants
                CCSP_TRK(0, 2 + (c < 8 ? c : 7));
                CCSP_TRK2(1 + 2 * (c < 8 ? c : 7));
                if constexpr (FWD) { if (c == NCH - 1) h2_epilogue_prefetch<ND>(bs0, 0, wr0, nrows, row0, colw, base); }
                const unsigned short* st = smem + (c & 1) * STAGE;
                const int ns = (c + 1) & 1;                           // the other stage = the register set that holds chunk c + 1
                __builtin_amdgcn_sched_barrier(0);
                fread(st, 1);                                         // k-step 1's fragments: in flight under k-step 0's MFMAs
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i][PA[q]], fb[0][j][PB[q]], acc[i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        const int k = q * 2 + i;                      // behind MFMA pair k: two of chunk c + 1's eight staging stores (pairs 0 .. 3)
                        if (c + 1 < NCH && k < 4) { lstore_one(ns, ns, 2 * k); lstore_one(ns, ns, 2 * k + 1); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                CCSP_TRK2(2 + 2 * (c < 8 ? c : 7));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // k-step 1's fragments have landed, this wave's stores too
                __builtin_amdgcn_s_barrier();                         // every wave is done reading stage c & 1; stage ns is visible
                __builtin_amdgcn_sched_barrier(0);
                if (c + 1 < NCH) fread(smem + ns * STAGE, 0);         // chunk c + 1's first k-step: in flight under k-step 1's MFMAs
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][i][PA[q]], fb[1][j][PB[q]], acc[i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        const int k = q * 2 + i;                      // behind MFMA pair k: chunk c + 3's eight global loads (2, 2, 1, 1, 1, 1)
                        if (c + 3 < NCH) {
                            if (k < 2) { gload_one(c + 3, ns, 2 * k); gload_one(c + 3, ns, 2 * k + 1); }
                            else gload_one(c + 3, ns, k + 2);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                             // every wave is done reading the stages (the epilogue tiles go on top)
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (DB) {
            gload(0, 0);
            gload(1, 1);
            row_exps();
            lstore(0, 0);
            gload(2, 0);
            __syncthreads();
#pragma unroll
            for (int c = 0; c < NCH; ++c) {                       // fully unrolled: register-set indices are constants
                // register set (c+1)&1 holds chunk c+1, the other one chunk c+2 (still in flight)
                if (c + 1 < NCH) lstore((c + 1) & 1, (c + 1) & 1);
                if (c + 3 < NCH) gload(c + 3, (c + 1) & 1);
                if constexpr (FWD) { if (c == NCH - 1) h2_epilogue_prefetch<ND>(bs0, 0, wr0, nrows, row0, colw, base); }
                const unsigned short* st = smem + (c & 1) * STAGE;
                h2_kstep<MI>(st, APL, st + 2 * APL, 0, wr0, wn * 64, acc);
                h2_kstep<MI>(st, APL, st + 2 * APL, 1, wr0, wn * 64, acc);
                __syncthreads();
            }
        } else {
            gload(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            row_exps();
            lstore(0, 0);
            gload(1, 0);
            __syncthreads();
            for (int c = 0; c < NCH; ++c) {
                CCSP_TRK(0, 2 + (c < 8 ? c : 7));
                CCSP_TRK2(1 + 2 * c);                             // chunk c: the stage is visible, the MFMAs begin
                if constexpr (FWD) { if (c == NCH - 1) h2_epilogue_prefetch<ND>(bs0, 0, wr0, nrows, row0, colw, base); }
                h2_kstep<MI>(smem, APL, smem + 2 * APL, 0, wr0, wn * 64, acc);
                h2_kstep<MI>(smem, APL, smem + 2 * APL, 1, wr0, wn * 64, acc);
                CCSP_TRK2(2 + 2 * c);                             // wave 0 has issued its MFMAs and waits at the barrier
                __syncthreads();                                  // every wave is done reading the stage
                if (c + 1 < NCH) {
                    lstore(0, 0);
                    if (c + 2 < NCH) gload(c + 2, 0);
                    __syncthreads();
                }
            }
        }
    }
    // (every wave is past the last barrier: the stages are free for the wave-private epilogue tiles)
    CCSP_TRK2(25);
    CCSP_TRK(0, 10);
    h2_epilogue_wave<ND, MI, FWD, PRE1>(acc, bs0, bs1, tv, reinterpret_cast<float*>(smem) + wave * H2_CW_SZ, wr0, nrows, row0, colw, sE, w_exp, base, has_tau,
                                        U, umax, 2 * NCT, 2 * ct + wn);
#ifdef CCSP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the exit stamp: after the stores have been acknowledged)
#endif
    CCSP_TRK_RT(0, 31);
    CCSP_TRK(0, 17);
    CCSP_TRK2(26);                                                // wave 0's epilogue has issued its last store
#ifdef CCSP_TRACE2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CCSP_TRK2(27);                                                // ... and its stores have been acknowledged
    CCSP_TRK2_FLUSH();
#endif
    gate_done(ref.gate);
}

// ------------------------------------------------------------------------------------------
// k_edge_h2<ENERGY, MT>: the decoder of 32 MT sorted edges, both output halves (denoise_fn.py:341-371).  With
// ME = 32 MT, row r < ME of the tile is (edge e0 + r, half 0), row ME + r is (edge e0 + r, half 1);
// A[row, :] = SiLU(U[u0] + U[u1])[half * H : +H] scaled by the row's exponent (bound from umax, see the header) and split
// in registers; B = planes of pose_decoder.0.weight [H/2, H] (staged once for both halves).  Waves 2(M) x 2(N): wave row
// wm is the half, 32 MT x 64 per wave.  Epilogue: 2^-(e_row + wd_exp) acc + bias -> SiLU -> LDS -> pose_decoder.2 -> CSR
// slot.  ENERGY as in k_edge<H, true>: -2 (o - pose) to the CSR slot, pre-activations to Q, partial sum of squares.
//   MT = 2: 64 edges, 64.5 KB, 2 workgroups per CU;  MT = 1: 32 edges, 48.3 KB, 3 per CU -- twice the workgroups, each
//   with half the SiLU / split work per thread: the kernel is one tile's latency chain long, so shorter chains on more
//   SIMDs win until the tile list no longer fits the CUs at once.
// ------------------------------------------------------------------------------------------
// second decoder layer of a 64-row S1 tile (o[row, p] = sum_j S1[row, j] Wd2[p, j], 128 hidden units): wave w multiplies hidden
// units 32 w .. 32 w + 31 of all 64 rows (lane = row: conflict-free 16-byte LDS reads at the 132-word row stride, weights
// wave-uniform -> scalar loads) and leaves its P partial sums in RED[w][p][row]; the caller adds the four partials in a fixed order.  32 LDS
// reads per lane instead of the 128 of a whole dot product per (row, p) item: 2.3 k cycles shorter per workgroup, which is what
// counts when the grid is a single round (k_edge_h2<., ., 1>, small batches: C5 +3.5 %); with several workgroups per CU in flight
// the extra barrier costs more than the reads (C2 -1 %), so large grids keep the one-pass form.  PP > 0: pose_dim at compile time.
template <int PP, int NROWS = 64>
__device__ __forceinline__ void h2_decoder_l2(const float* __restrict__ S1, int s1_ld, const float* __restrict__ Wd2, int P, float* __restrict__ RED,
                                              int wave, int lane) {
    if (NROWS < 64 && lane >= NROWS) return;                      // (32-row tiles: the upper half-wave has no row)
    const int kq = __builtin_amdgcn_readfirstlane(wave) * 32;
    float part[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) part[p] = 0.0f;
    const float4* srow = reinterpret_cast<const float4*>(S1 + lane * s1_ld + kq);      // (s1_ld = 132: 16-byte aligned rows, conflict-free b128 reads)
    const float* w = Wd2 + kq;
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
        const float4 sv4 = srow[k4];
        const float sv[4] = {sv4.x, sv4.y, sv4.z, sv4.w};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int p = 0; p < 8; ++p)
                if (PP > 0 ? p < PP : p < P) part[p] = fmaf(sv[kk], w[p * 128 + 4 * k4 + kk], part[p]);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p)
        if (PP > 0 ? p < PP : p < P) RED[(wave * 8 + p) * NROWS + lane] = part[p];
}

// Tail of the fused edge kernels (FUSE): the node update of every 16-node block this workgroup is the LAST to deliver edge
// outputs to (FuseArgs, ccsp_hip.hip).  The outputs were stored write-through (sc1: they leave the XCD's L2 at once); every
// wave drains its stores, the workgroup synchronises, then one relaxed agent-scope fetch_add per touched block -- the
// publish form R1 of cdna_hip_programming.md Guideline 16, with the arrival counter as the flag.  Nobody waits: whoever
// completes a block runs it (sc1 loads of the edge outputs), so a launch has no barrier and cannot hang.  lds: the stages,
// free after the epilogue.
// one row of P edge outputs, write-through (sc1), from ONE lane in 16-byte pieces: a 4-byte sc1 store is one fabric write each
// (MI355X_MICROARCH.md: 6x the time per byte of a 16-byte one) -- the first fused build, one element per lane, ran the chain
// at a third of the speed.  The asm stores are invisible to hipcc's wait insertion: edge_node_tail drains them.
__device__ __forceinline__ void h2_store_row_sc1(float* dst, const float* src /*LDS, 16-byte aligned*/, int P) {
    int c = 0;
    for (; c + 4 <= P; c += 4) {
        const h2_f4 v = *reinterpret_cast<const h2_f4*>(src + c);
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst + c), "v"(v) : "memory");
    }
    for (; c < P; ++c) __hip_atomic_store(dst + c, src[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void edge_node_tail(const FuseArgs& fu, int wg, void* lds) {
    const int tid = threadIdx.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's write-through stores have been acknowledged
    __syncthreads();                                              // ... every wave's; and the epilogue's LDS tiles are dead
    int* s_list = reinterpret_cast<int*>(reinterpret_cast<char*>(lds) + ((NODE_LDS_BYTES + 15) & ~15));         // [256]
    const int b0 = fu.wg_blk_ptr[wg], nb = fu.wg_blk_ptr[wg + 1] - b0;
    for (int base = 0; base < nb; base += 256) {
        const int cnt = nb - base < 256 ? nb - base : 256;
        if (tid < cnt) {
            const int b = fu.wg_blk[b0 + base + tid];
            const unsigned int old = __hip_atomic_fetch_add(fu.blk_count + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_list[tid] = (old + 1u == (unsigned int)fu.blk_expect[b] * fu.epoch) ? b : -1;
        }
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {                           // (uniform: the list is in LDS)
            const int b = s_list[j];
            if (b >= 0) {
                node_block_direct<true>(fu.node, fu.w, fu.eo, fu.n_ent, b * NODE_TILE, node_lds(lds));
                __syncthreads();                                  // (the block's LDS is reused by the next one)
            }
        }
        __syncthreads();
    }
}

// NG (round 4, CCSP_FUSE_NODE=2): the NODE-GROUPED form.  The decoder is shared by all types and halves and every output goes to its own
// CSR slot, so a workgroup may take ANY 64 (edge, half) rows: here it takes the CSR entries of its own run of consecutive nodes (<= 16
// nodes, <= 64 entries; ng_desc / ng_off0 / ng_off1 of FuseArgs, built by fuse2_prepare), i.e. the edge work is cut by DESTINATION node.
// Every (edge, half) row is still computed exactly once, and when the tile is done the workgroup holds every input of its nodes'
// update: the node kernel runs as its tail (node_group_tail) with no hand-over between workgroups -- no counters, no write-through
// stores, no waiting for the slowest tile (what round 3's producer-side fusion, FUSE, paid for).  Same arithmetic, same order.
template <bool ENERGY, int MT, int L2, bool FUSE = false, bool NG = false>
__global__ __launch_bounds__(256, (MT == 1 && !FUSE) ? 3 : 2) void k_edge_h2(int E_act, int P, const int* __restrict__ e_u0, const int* __restrict__ e_u1,
                                                    const float* __restrict__ U, const float* __restrict__ umax /*[R][8]*/,
                                                    const unsigned short* __restrict__ Wd1H /*[128][256 / 32][2][32]: both planes of a row's K chunk in one line*/, int wd_exp,
                                                    const float* __restrict__ bd1, const float* __restrict__ Wd2,
                                                    const float* __restrict__ bd2, const int* __restrict__ ent_pos, float* __restrict__ O,
                                                    EdgeEnergyArgs en, int* __restrict__ counter_inc, FuseArgs fu) {
    static_assert(!(FUSE && ENERGY), "the fused node update is the direct-mode one");
    static_assert(!NG || (!ENERGY && !FUSE && MT == 1), "the node-grouped form: direct mode, 64-row tiles");
    if constexpr (ENERGY) { if (en.skip && *en.skip == 0) return; }                        // (uniform) MALA reuse
    if (counter_inc && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(counter_inc, 1);     // hipGraph mode: next table entry
    gate_wait(en.gate);
    CCSP_TRK(1, 0);
    CCSP_TRK_RT(1, 30);
    constexpr int H = 256, BN = 128, NCH = H / H2_BK;
    constexpr int ME = 32 * MT, ROWS = 2 * ME;                    // edges, tile rows
    constexpr int APL = ROWS * H2_BK, STAGE = 2 * APL + 2 * H2_BPL;
    constexpr int NPASS = ROWS / 32;                              // producer passes per chunk: rows lr + 32 i
    constexpr int S1_LD = BN + 4;                                 // 16-byte aligned rows; one row per lane reads conflict-free (b128: 4 rows x 132 words = all 64 banks per 16 lanes)
    static_assert((64 * S1_LD + 8 * BN + 4 * 8 * 64 + 64 * 8) * 4 <= 2 * STAGE * 2, "epilogue tile, the layer-2 weights, partials and output rows must fit the stages");
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * STAGE + 2 * ROWS];
    int* sE = reinterpret_cast<int*>(smem + 2 * STAGE);
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int e0 = wg * ME;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = tid >> 3, lq = tid & 7;                        // A producer: rows lr + 32 i, fp32 columns 4 lq .. + 3 of the chunk
    int4 ngd = make_int4(0, 0, 0, 0);                             // NG: {first node, nodes, first CSR entry, entries}
    NodeGroupPre ngp{};
    if constexpr (NG) {
        ngd = fu.ng_desc[wg];
        ngp = node_group_pre(fu.node, ngd.x, ngd.y);              // the update's own loads, at entry: nothing of the tile is needed for them
    }
    // The kernel is one latency chain per tile: edge -> row indices -> U rows -> ... .  Issue order (vector-memory loads return
    // in order): the row indices; then, the moment they are here, the U rows of the first two chunks and the weights of the
    // first; only then the row maxima (the exponents are needed when chunk 0 is written to LDS, not before) and what the
    // epilogue needs -- CSR slot, biases, second-layer weights -- so that no load sits in the epilogue's path.
    int r0v[NPASS], r1v[NPASS];                                   // (NG: element offsets of the row's two operands into U -- U row * 2H + half * H)
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        if constexpr (NG) {
            r0v[i] = fu.ng_off0[wg * ROWS + lr + 32 * i];
            r1v[i] = fu.ng_off1[wg * ROWS + lr + 32 * i];
        } else {
            int k = e0 + ((lr + 32 * i) % ME);
            k = k < E_act ? k : E_act - 1;
            r0v[i] = e_u0[k];
            r1v[i] = e_u1[k];
        }
    }
    const float* u0_ptr[NPASS];
    const float* u1_ptr[NPASS];
    int a_st[NPASS], a_exp[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int row = lr + 32 * i, s = row / ME;
        if constexpr (NG) {
            u0_ptr[i] = U + r0v[i] + lq * 4;
            u1_ptr[i] = U + r1v[i] + lq * 4;
        } else {
            u0_ptr[i] = U + (size_t)r0v[i] * (2 * H) + s * H + lq * 4;
            u1_ptr[i] = U + (size_t)r1v[i] * (2 * H) + s * H + lq * 4;
        }
        a_st[i] = h2_off(row, lq >> 1) + (lq & 1) * 4;
    }
    const int brow = tid >> 2, bq = tid & 3;                      // B copy: rows brow, brow + 64, piece bq, both planes
    const unsigned short* b_ptr = Wd1H + (size_t)brow * (2 * H) + bq * 8;       // (chunk-interleaved planes [128][H / 32][2][32]: k_interleave_planes)
    const int b_st = h2_off(brow, bq);
    float4 ua[2][NPASS], ub[2][NPASS];                            // [register set][pass]
    ushort8 rb[4];
    auto gload_a = [&](int c, int set) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            ua[set][i] = *reinterpret_cast<const float4*>(u0_ptr[i] + c * H2_BK);
            ub[set][i] = *reinterpret_cast<const float4*>(u1_ptr[i] + c * H2_BK);
        }
    };
    auto gload_b = [&](int c) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                rb[i * 2 + p] = *reinterpret_cast<const ushort8*>(b_ptr + (size_t)p * H2_BK + (size_t)i * 64 * (2 * H) + c * (2 * H2_BK));
    };
    auto store_a = [&](int stage, int set, int i) {               // SiLU + scale + split of one pass -> the A planes of the stage
        unsigned short* As = smem + stage * STAGE;
        uint2 hi, lo;
        h2_act4(ua[set][i], ub[set][i], a_exp[i], hi, lo);
        unsigned short* d = As + a_st[i];
        *reinterpret_cast<uint2*>(d) = hi;
        *reinterpret_cast<uint2*>(d + APL) = lo;
    };
    auto store_b = [&](int stage) {
        unsigned short* Bs = smem + stage * STAGE + 2 * APL;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                *reinterpret_cast<ushort8*>(Bs + p * H2_BPL + b_st + i * 64 * H2_BK) = rb[i * 2 + p];
    };
    gload_a(0, 0);
    gload_b(0);
    gload_a(1, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int row = lr + 32 * i, s = row / ME;
        // (NG: offset = U row * 512 + half * 256  ->  umax index U row * 8 + half * 4 = offset / 64)
        const float4 m0 = *reinterpret_cast<const float4*>(umax + (NG ? (size_t)(r0v[i] >> 6) : (size_t)r0v[i] * 8 + 4 * s));
        const float4 m1 = *reinterpret_cast<const float4*>(umax + (NG ? (size_t)(r1v[i] >> 6) : (size_t)r1v[i] * 8 + 4 * s));
        // |SiLU(z)| <= |z| <= max|U[u0]| + max|U[u1]| over the half's four 64-column pieces
        a_exp[i] = h2_scale_exp(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)) + fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w)));
    }
    // what the epilogue reads, requested now: the first-layer bias of the lane's two columns, the CSR slots / bias of the
    // (row, p) outputs this thread produces (one per 64-row pass when 64 P <= 256), its share of the second-layer weights
    float bj1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bj1[j] = bd1[wn * 64 + j * 32 + (lane & 31)];
    const int o_p = tid >> 6, o_row = tid & 63;                   // output item of this thread in a 64-row pass: S1 row o_row, component o_p
    int o_slot[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if constexpr (NG) {
            o_slot[i] = ngd.z + o_row;                            // the row IS the CSR entry
        } else {
            int k = e0 + i * 32 + (o_row & 31);
            k = k < E_act ? k : E_act - 1;
            o_slot[i] = ent_pos[2 * k + (o_row >> 5)];
        }
    }
    const float o_b2 = bd2[o_p < P ? o_p : 0];
    const float4 w2v = *reinterpret_cast<const float4*>(Wd2 + ((tid * 4) < P * BN ? tid * 4 : 0));      // Wd2 is [P][128] row-major: 4 P x 32 float4
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NPASS; ++i)
        if (lq == 0) sE[lr + 32 * i] = a_exp[i];
    CCSP_TRK(1, 1);
    floatx16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) store_a(0, 0, i);
    store_b(0);
    gload_b(1);
    gload_a(2, 0);
    __syncthreads();
    CCSP_TRK(1, 2);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {                               // fully unrolled: the register-set index is a constant
        const unsigned short* st = smem + (c & 1) * STAGE;
        const int nx = (c + 1) & 1;                               // next stage, and the register set holding chunk c+1
        h2_kstep<MT>(st, APL, st + 2 * APL, 0, wm * ME, wn * 64, acc);
        if (c + 1 < NCH) {                                        // in the shadow of the MFMAs just issued
#pragma unroll
            for (int i = 0; i < NPASS / 2; ++i) store_a(nx, nx, i);
        }
        h2_kstep<MT>(st, APL, st + 2 * APL, 1, wm * ME, wn * 64, acc);
        if (c + 1 < NCH) {
#pragma unroll
            for (int i = NPASS / 2; i < NPASS; ++i) store_a(nx, nx, i);
            store_b(nx);
        }
        if (c + 2 < NCH) gload_b(c + 2);
        if (c + 3 < NCH) gload_a(c + 3, nx);
        __syncthreads();
        CCSP_TRK(1, 3 + c);
    }
    // epilogue, 64 rows per pass (row tile i of every wave: 32 edges x both halves)
    float* S1 = reinterpret_cast<float*>(smem);
    float* W2s = S1 + 64 * S1_LD;                                 // pose_decoder.2.weight [P][128] (read as broadcast b128 rows)
    float* RED = W2s + 8 * BN;
    float* Os = RED + 4 * 8 * 64;                                 // FUSE: the 64 x P outputs of a pass, one row per lane for the 16-byte stores
    if (tid * 4 < P * BN) *reinterpret_cast<float4*>(W2s + tid * 4) = w2v;
    float e2 = 0.0f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int row = wm * ME + i * 32 + rr;                         // tile row: half wm, edge e0 + i * 32 + rr
                const float q = ldexpf(acc[i][j][r], -(sE[row] + wd_exp)) + bj1[j];
                S1[(wm * 32 + rr) * S1_LD + col] = silu_fast(q);
                if constexpr (ENERGY) {
                    const int k = e0 + i * 32 + rr;
                    if (en.Q && k < E_act) en.Q[((size_t)2 * k + wm) * BN + col] = q;
                }
            }
        }
        __syncthreads();
        CCSP_TRK(1, 11);
        if constexpr (L2 == 1) {                                               // (see h2_decoder_l2)
            if (P == 4) h2_decoder_l2<4>(S1, S1_LD, Wd2, P, RED, wave, lane);
            else if (P == 5) h2_decoder_l2<5>(S1, S1_LD, Wd2, P, RED, wave, lane);
            else h2_decoder_l2<0>(S1, S1_LD, Wd2, P, RED, wave, lane);
            __syncthreads();
            CCSP_TRK(1, 12);
        }
        for (int idx = tid; idx < 64 * P; idx += 256) {
            const int lrow = idx & 63;                                         // S1 row: half (lrow >> 5), edge e0 + i * 32 + (lrow & 31)
            const int p = idx >> 6;
            float o;
            if constexpr (L2 == 1) {
                const float* rp = RED + p * 64 + lrow;
                o = ((rp[0] + rp[8 * 64]) + (rp[16 * 64] + rp[24 * 64])) + bd2[p];
            } else {
                // one (row, p) dot product per thread: S1 row and weight row as 16-byte LDS reads (the weight row is the same
                // address for the whole wave: a broadcast), four independent chains summed as dot4 does
                const float4* sr = reinterpret_cast<const float4*>(S1 + lrow * S1_LD);
                const float4* wr = reinterpret_cast<const float4*>(W2s + p * BN);
                float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f, o3 = 0.0f;
#pragma unroll 8
                for (int j = 0; j < BN / 4; ++j) {
                    const float4 sv = sr[j], wv = wr[j];
                    o0 = fmaf(sv.x, wv.x, o0); o1 = fmaf(sv.y, wv.y, o1); o2 = fmaf(sv.z, wv.z, o2); o3 = fmaf(sv.w, wv.w, o3);
                }
                o = ((o0 + o1) + (o2 + o3)) + (idx == tid ? o_b2 : bd2[p]);
            }
            const int k = e0 + i * 32 + (lrow & 31), s = lrow >> 5;
            if (NG ? lrow < ngd.w : k < E_act) {
                const int slot = NG ? ngd.z + lrow : (idx == tid ? o_slot[i] : ent_pos[2 * k + s]);  // (the first 256 items' slots were requested in the prologue)
                if constexpr (ENERGY) {
                    const int node = s == 0 ? en.e_a[k] : en.e_b[k];
                    const float d = o - en.xeval[(size_t)node * P + p];
                    e2 = fmaf(d, d, e2);
                    O[(size_t)slot * P + p] = -2.0f * d;
                } else if constexpr (!FUSE) {
                    O[(size_t)slot * P + p] = o;                               // straight to the node's CSR slot
                }
            }
            if constexpr (FUSE || NG) Os[lrow * 8 + p] = o;
        }
        if constexpr (FUSE) {                                                  // rows out through LDS: one lane per row, 16-byte write-through stores
            __syncthreads();
            if (tid < 64) {
                const int k = e0 + i * 32 + (tid & 31);
                if (k < E_act) h2_store_row_sc1(O + (size_t)o_slot[i] * P, Os + tid * 8, P);
            }
        }
        if (i + 1 < MT || ENERGY) __syncthreads();                             // (the next pass / the energy reduction reuse S1)
    }
    if constexpr (NG) {
        __syncthreads();                                                       // the outputs of every row are in Os; S1 is dead (the node block's LDS)
        node_group_tail(fu.node, fu.w, fu.eo, ngd.x, ngd.y, ngd.z, Os, node_lds(smem), ngp);
        return;
    }
    if constexpr (FUSE) {
        CCSP_TRK(1, 13);
        edge_node_tail(fu, e0 / ME, smem);
        CCSP_TRK_RT(1, 31);
        CCSP_TRK(1, 14);
        return;
    }
    CCSP_TRK(1, 13);
    CCSP_TRK_RT(1, 31);
    if constexpr (ENERGY) {
        const float tot = block_sum_256(e2, reinterpret_cast<float*>(smem));
        if (tid == 0) en.partial[blockIdx.x] = tot;
    }
    gate_done(en.gate);
}

// ------------------------------------------------------------------------------------------
// k_edge_h2s: k_edge_h2 for grids that leave most of the chip empty (at most one workgroup per CU even at 16 edges per
// workgroup: small batches).  There the kernel is the lifetime of ONE workgroup with one wave per SIMD; inside a wave the LDS
// round trips, the SiLU / scale / split of the A rows on the VALU and the MFMAs do not overlap, so what counts is the work per
// wave: the tile is 16 sorted edges x both halves = 32 rows, all four waves share them and take 32 of the 128 decoder columns
// each -- half the activation work per thread and half the MFMAs per wave of the 32-edge tile, twice the workgroups.  Same
// operands, same scaling, same epilogue arithmetic as k_edge_h2<ENERGY, 1, 1>.
// ------------------------------------------------------------------------------------------
template <bool ENERGY, bool FUSE = false>
__global__ __launch_bounds__(256, 2) void k_edge_h2s(int E_act, int P, const int* __restrict__ e_u0, const int* __restrict__ e_u1,
                                                     const float* __restrict__ U, const float* __restrict__ umax /*[R][8]*/,
                                                     const unsigned short* __restrict__ Wd1H /*[128][256 / 32][2][32]: both planes of a row's K chunk in one line*/, int wd_exp,
                                                     const float* __restrict__ bd1, const float* __restrict__ Wd2,
                                                     const float* __restrict__ bd2, const int* __restrict__ ent_pos, float* __restrict__ O,
                                                     EdgeEnergyArgs en, int* __restrict__ counter_inc, FuseArgs fu) {
    static_assert(!(FUSE && ENERGY), "the fused node update is the direct-mode one");
    if constexpr (ENERGY) { if (en.skip && *en.skip == 0) return; }                        // (uniform) MALA reuse
    if (counter_inc && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(counter_inc, 1);     // hipGraph mode: next table entry
    gate_wait(en.gate);
    CCSP_TRK(1, 0);
    CCSP_TRK_RT(1, 30);
    constexpr int H = 256, BN = 128, NCH = H / H2_BK;
    constexpr int ME = 16, ROWS = 2 * ME;
    constexpr int APL = ROWS * H2_BK, STAGE = 2 * APL + 2 * H2_BPL;       // 4 KB of A planes + 16 KB of B planes
    constexpr int S1_LD = BN + 4;
    static_assert((ROWS * S1_LD + 4 * 8 * ROWS + 8 * ROWS) * 4 <= 2 * STAGE * 2, "epilogue tile, the layer-2 partials and the output rows must fit the stages");
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * STAGE + 2 * ROWS];
    int* sE = reinterpret_cast<int*>(smem + 2 * STAGE);
    const int e0 = xcd_remap(blockIdx.x, gridDim.x) * ME;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = tid >> 3, lq = tid & 7;                        // A producer: tile row lr, fp32 columns 4 lq .. + 3 of the chunk
    const int hs = lr / ME;                                       // half (slot) of the row
    int kk = e0 + (lr % ME);
    kk = kk < E_act ? kk : E_act - 1;
    const int r0 = e_u0[kk], r1 = e_u1[kk];
    const float* u0_ptr = U + (size_t)r0 * (2 * H) + hs * H + lq * 4;
    const float* u1_ptr = U + (size_t)r1 * (2 * H) + hs * H + lq * 4;
    const float4 m0 = *reinterpret_cast<const float4*>(umax + (size_t)r0 * 8 + 4 * hs);
    const float4 m1 = *reinterpret_cast<const float4*>(umax + (size_t)r1 * 8 + 4 * hs);
    const int brow = tid >> 2, bq = tid & 3;                      // B copy: rows brow, brow + 64, piece bq, both planes
    const unsigned short* b_ptr = Wd1H + (size_t)brow * (2 * H) + bq * 8;       // (chunk-interleaved planes [128][H / 32][2][32]: k_interleave_planes)
    const int b_st = h2_off(brow, bq);
    const int a_st = h2_off(lr, lq >> 1) + (lq & 1) * 4;
    float4 ua[2], ub[2];                                          // [register set]
    ushort8 rb[4];
    auto gload_a = [&](int c, int set) {
        ua[set] = *reinterpret_cast<const float4*>(u0_ptr + c * H2_BK);
        ub[set] = *reinterpret_cast<const float4*>(u1_ptr + c * H2_BK);
    };
    auto gload_b = [&](int c) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                rb[i * 2 + p] = *reinterpret_cast<const ushort8*>(b_ptr + (size_t)p * H2_BK + (size_t)i * 64 * (2 * H) + c * (2 * H2_BK));
    };
    gload_a(0, 0);
    gload_b(0);
    gload_a(1, 1);
    // |SiLU(z)| <= |z| <= max|U[u0]| + max|U[u1]| over the half's four 64-column pieces
    const int a_exp = h2_scale_exp(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)) + fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w)));
    if (lq == 0) sE[lr] = a_exp;
    auto store_a = [&](int stage, int set) {                      // SiLU + scale + split -> the A planes of the stage
        unsigned short* As = smem + stage * STAGE;
        uint2 hi, lo;
        h2_act4(ua[set], ub[set], a_exp, hi, lo);
        unsigned short* d = As + a_st;
        *reinterpret_cast<uint2*>(d) = hi;
        *reinterpret_cast<uint2*>(d + APL) = lo;
    };
    auto store_b = [&](int stage) {
        unsigned short* Bs = smem + stage * STAGE + 2 * APL;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                *reinterpret_cast<ushort8*>(Bs + p * H2_BPL + b_st + i * 64 * H2_BK) = rb[i * 2 + p];
    };
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    store_a(0, 0);
    store_b(0);
    gload_b(1);
    gload_a(2, 0);
    // (bare barriers through the K loop: __syncthreads() carries a fence that waits for EVERY outstanding load -- the operands requested two and
    // three chunks ahead were drained at each chunk's barrier, one full memory round trip per chunk in a kernel that is one workgroup's latency
    // chain.  The LDS side is all the barrier has to order: lgkmcnt(0) for this wave's stage stores; hipcc counts the loads where they are used.)
    auto stage_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    stage_barrier();
    CCSP_TRK(1, 2);
    auto kstep = [&](const unsigned short* st, int ks) {          // 32 x 32 of the wave: rows 0..31, columns 32 wave .. + 31
        const int piece = (lane >> 5) + 2 * ks;
        half8 a[2], b[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            a[p] = *reinterpret_cast<const half8*>(st + p * APL + h2_off(lane & 31, piece));
            b[p] = *reinterpret_cast<const half8*>(st + 2 * APL + p * H2_BPL + h2_off(wave * 32 + (lane & 31), piece));
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc, 0, 0, 0);      // smallest terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
    };
#pragma unroll
    for (int c = 0; c < NCH; ++c) {                               // fully unrolled: the register-set index is a constant
        const unsigned short* st = smem + (c & 1) * STAGE;
        const int nx = (c + 1) & 1;                               // next stage, and the register set holding chunk c+1
        kstep(st, 0);
        if (c + 1 < NCH) store_a(nx, nx);                         // in the shadow of the MFMAs just issued
        kstep(st, 1);
        if (c + 1 < NCH) store_b(nx);
        if (c + 2 < NCH) gload_b(c + 2);
        if (c + 3 < NCH) gload_a(c + 3, nx);
        stage_barrier();
        CCSP_TRK(1, 3 + c);
    }
    // epilogue: 32 rows (half = row / 16, edge e0 + row % 16)
    float* S1 = reinterpret_cast<float*>(smem);
    float* RED = S1 + ROWS * S1_LD;
    {
        const int col = wave * 32 + (lane & 31);
        const float bj = bd1[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float q = ldexpf(acc[r], -(sE[row] + wd_exp)) + bj;
            S1[row * S1_LD + col] = silu_fast(q);
            if constexpr (ENERGY) {
                const int k = e0 + (row % ME);
                if (en.Q && k < E_act) en.Q[((size_t)2 * k + row / ME) * BN + col] = q;
            }
        }
    }
    __syncthreads();
    CCSP_TRK(1, 11);
    if (P == 4) h2_decoder_l2<4, ROWS>(S1, S1_LD, Wd2, P, RED, wave, lane);
    else if (P == 5) h2_decoder_l2<5, ROWS>(S1, S1_LD, Wd2, P, RED, wave, lane);
    else h2_decoder_l2<0, ROWS>(S1, S1_LD, Wd2, P, RED, wave, lane);
    __syncthreads();
    CCSP_TRK(1, 12);
    float e2 = 0.0f;
    for (int idx = tid; idx < ROWS * P; idx += 256) {
        const int row = idx & (ROWS - 1), p = idx / ROWS;
        const float* rp = RED + p * ROWS + row;
        const float o = ((rp[0] + rp[8 * ROWS]) + (rp[16 * ROWS] + rp[24 * ROWS])) + bd2[p];
        const int k = e0 + (row % ME), s = row / ME;
        if (k < E_act) {
            if constexpr (ENERGY) {
                const int node = s == 0 ? en.e_a[k] : en.e_b[k];
                const float d = o - en.xeval[(size_t)node * P + p];
                e2 = fmaf(d, d, e2);
                O[(size_t)ent_pos[2 * k + s] * P + p] = -2.0f * d;
            } else if constexpr (!FUSE) {
                O[(size_t)ent_pos[2 * k + s] * P + p] = o;                 // straight to the node's CSR slot
            }
        }
        if constexpr (FUSE) (RED + 4 * 8 * ROWS)[row * 8 + p] = o;
    }
    if constexpr (FUSE) {                                             // rows out through LDS: one lane per row, 16-byte write-through stores
        __syncthreads();
        if (tid < ROWS) {
            const int k = e0 + (tid % ME), s = tid / ME;
            if (k < E_act) h2_store_row_sc1(O + (size_t)ent_pos[2 * k + s] * P, RED + 4 * 8 * ROWS + tid * 8, P);
        }
        edge_node_tail(fu, e0 / ME, smem);
    }
    if constexpr (ENERGY) {
        __syncthreads();
        const float tot = block_sum_256(e2, reinterpret_cast<float*>(smem));
        if (tid == 0) en.partial[blockIdx.x] = tot;
    }
    CCSP_TRK(1, 13);
    CCSP_TRK_RT(1, 31);
    gate_done(en.gate);
}

// ------------------------------------------------------------------------------------------
// k_edge_bwd_h2: the decoder backward of the energy mode (k_edge_bwd_bf of ccsp_bf16x3.h) on the f16 pipe.
// Rows = (sorted edge, slot s); K = H/2 = 128 decoder hidden units; N = 128 of the H columns per workgroup.
// A[row, j] = (sum_p go[p] Wd2[p, j]) * SiLU'(q[row, j]), go = 2 d = -O_csr, built on the VALU for chunk c+1 between the
// MFMA groups of chunk c.  Row exponent from the bound |A| <= 1.1 * max|Wd2| * sum_p |go[p]|  (|SiLU'| < 1.1): every thread
// of a row has its go values, so the exponent costs nothing.  B = fp16 planes of Wd1^T [H, H/2] (same exponent as Wd1).
// Epilogue: GZ[k, s H + n] = 2^-(e_row + wd_exp) acc * SiLU'(U[u0] + U[u1])[s H + n], through LDS as row segments.
// ------------------------------------------------------------------------------------------
// SUM (round 4): the ordered row sums of g_z (k_rowsum_h2) formed here, in the epilogue.  A workgroup's 64 edges touch a handful of
// distinct U rows; ccsp::build_bwdsum_plan lists them per edge block (a PARTIAL ROW per (block, U row)) with the block-local edges of
// each.  The scaled tile goes back into LDS, every partial row is added up in ascending edge order and written as the transpose row
// GEMM's operand planes -- which then runs on partial rows (linearity; the node kernel adds a node's partial-row products).  Row
// exponent from a bound every one of the block's four workgroups can form alone: |g_z[k, s H + n]| <= 1.1 |acc| <= 1.1 max_j |A[k, s, j]|
// max_n sum_j |Wd1[j, n]| <= bound_c sum_p |go[k, s, p]|, summed over both halves of an edge and over a partial row's edges.  A loose
// bound costs nothing: elements keep their 22 bits down to 2^-18 of the scaled bound, below that the absolute error is 2^-25 of a
// scaled unit (fp16 subnormals) -- far under the fp32 rounding of the K = 512 accumulation the planes feed.
struct BwdSumArgs {
    const int* blocks;            // [n_blocks][ccsp::BS_BLK]
    unsigned short* GZPH;         // [2][NP][512] fp16 planes of the partial rows
    size_t plane;                 // NP * 512
    int* gexp;                    // [NP]
    float bound_c;                // 1.21 max|Wd2| max_n sum_j |Wd1[j, n]|, rounded up
};
// PP: pose_dim as a compile-time constant (4: every world but the robot's), or 0 = the run-time P with the loops run to 8 under selects --
// which is 8 LDS reads, 32 FMAs and 32 selects per four A elements where pose_dim 4 needs 4 reads and 16 FMAs, in the loop that bounds
// this kernel (VALU: 7.1 M instructions per launch at C4 against 7.7 M MFMA-busy cycles)
template <bool SUM, int PP>
__global__ __launch_bounds__(256, 3) void k_edge_bwd_h2(int E_act, int P_rt, const int* __restrict__ e_u0, const int* __restrict__ e_u1,
                                                        const int* __restrict__ ent_pos, const float* __restrict__ U,
                                                        const float* __restrict__ Ocsr, const float* __restrict__ Q /*[2E,128]*/,
                                                        const unsigned short* __restrict__ Wd1TH /*[2][256][128]*/, int wd_exp, float wd2_absmax,
                                                        const float* __restrict__ Wd2 /*[P,128]*/, float* __restrict__ GZ,
                                                        const int* __restrict__ skip /*MALA reuse, or null*/, BwdSumArgs bs) {
    if (skip && *skip == 0) return;
    constexpr int H = 256, KD = 128, BM = 64, BN = 128, NCH = KD / H2_BK;
    constexpr int PMAX = PP ? PP : 8;
    const int P = PP ? PP : P_rt;
    constexpr int APL = BM * H2_BK, STAGE = 2 * APL + 2 * H2_BPL;             // 8 KB of A planes + 16 KB of B planes per stage
    constexpr int C_LD = BN + 4;
    static_assert(2 * STAGE * 2 >= BM * C_LD * 4, "epilogue tile must fit the stages");
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * STAGE + 2 * 8 * KD];          // stages + pose_decoder.2.weight [8][128] fp32
    float* W2s = reinterpret_cast<float*>(smem + 2 * STAGE);
    // SUM: the block's partial rows (build_bwdsum_plan) and the per-edge bounds on |g_z[k, :]| take the staged weight's bytes once the K
    // loop is over (52 KB in all: three workgroups per CU, as without SUM -- with 1.8 KB more there were two, and the kernel 50 us for 29)
    static_assert((ccsp::BS_BLK + 128) * 4 <= 8 * KD * 4 && ccsp::BS_BLK <= 3 * 256, "partial-row block and exponents must fit the staged weight");
    static_assert(ccsp::BS_CLD == C_LD && ccsp::BS_EDGES == BM && 2 * STAGE * 2 >= (BM + 1) * C_LD * 4, "tile shape of build_bwdsum_plan");
    int* bsb = reinterpret_cast<int*>(W2s);
    int* pexp = bsb + ccsp::BS_BLK;                               // [128] exponent of every partial row
    int bsr[3] = {0, 0, 0};
    float bnd_r[2] = {0.0f, 0.0f};
    // Round 3: this kernel spent most of its 30 us in serialized round trips -- the P loads of -O under `p < P` branches, and the P
    // rows of pose_decoder.2.weight re-read from global memory for EVERY row pass of EVERY chunk, each load under its own branch
    // and waited for with vmcnt(0) (hipcc's wait insertion takes the minimum over paths, and gfx950 counts loads and stores on
    // one counter).  Now: the weight is staged in LDS once, rows p >= P zeroed, and the -O loads are unconditional (clamped
    // index, select) -- no branch touches a load.
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = bid & 1, s = (bid >> 1) & 1, e0 = (bid >> 2) * BM;
    const int n0 = ct * BN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = tid >> 3, lq = tid & 7;
    float go[2][8];
    const float* q_ptr[2];
    int a_st[2], a_exp[2], o_pos[2], ku0[2], ku1[2];
    int o_pos2[2] = {0, 0};                                       // SUM: CSR slot of the edge's OTHER half (its go values enter the bound)
    // index loads first: everything below hangs off them (CSR slot -> go -> row exponent; U rows of the epilogue)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int k = e0 + lr + 32 * i;
        k = k < E_act ? k : E_act - 1;
        const size_t row = (size_t)2 * k + s;
        o_pos[i] = ent_pos[row];
        if constexpr (SUM) o_pos2[i] = ent_pos[row ^ 1];
        ku0[i] = e_u0[k];
        ku1[i] = e_u1[k];
        q_ptr[i] = Q + row * KD + lq * 4;
        a_st[i] = h2_off(lr + 32 * i, lq >> 1) + (lq & 1) * 4;
    }
    const int brow = tid >> 2, bq = tid & 3;
    const unsigned short* b_ptr = Wd1TH + (size_t)(n0 + brow) * (2 * KD) + bq * 8;       // (chunk-interleaved planes [256][128 / 32][2][32])
    const int b_st = h2_off(brow, bq);
    float4 rq[2][2];                                              // [register set][pass]: decoder pre-activations
    ushort8 rb[4];
    auto gload_a = [&](int c, int set) {
#pragma unroll
        for (int i = 0; i < 2; ++i) rq[set][i] = *reinterpret_cast<const float4*>(q_ptr[i] + c * H2_BK);
    };
    auto gload_b = [&](int c) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                rb[i * 2 + p] = *reinterpret_cast<const ushort8*>(b_ptr + (size_t)p * H2_BK + (size_t)i * 64 * (2 * KD) + c * (2 * H2_BK));
    };
    auto store_a = [&](int stage, int c, int set, int i) {
        unsigned short* As = smem + stage * STAGE;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int p = 0; p < PMAX; ++p) {
            const float4 w2 = *reinterpret_cast<const float4*>(W2s + p * KD + c * H2_BK + lq * 4);
            const float gp = go[i][p];
            // (terms p >= P: go = 0 and a zero weight row -- exactly 0, added to nothing: the select keeps a NaN / Inf out)
            g.x = (PP || p < P) ? fmaf(gp, w2.x, g.x) : g.x; g.y = (PP || p < P) ? fmaf(gp, w2.y, g.y) : g.y;
            g.z = (PP || p < P) ? fmaf(gp, w2.z, g.z) : g.z; g.w = (PP || p < P) ? fmaf(gp, w2.w, g.w) : g.w;
        }
        const float4 q = rq[set][i];
        const float h[4] = {g.x * silu_grad_fast(q.x), g.y * silu_grad_fast(q.y), g.z * silu_grad_fast(q.z), g.w * silu_grad_fast(q.w)};
        unsigned short p1[4], p2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split2h(ldexpf(h[e], a_exp[i]), p1[e], p2[e]);
        unsigned short* d = As + a_st[i];
        *reinterpret_cast<uint2*>(d) = make_uint2(p1[0] | ((unsigned)p1[1] << 16), p1[2] | ((unsigned)p1[3] << 16));
        *reinterpret_cast<uint2*>(d + APL) = make_uint2(p2[0] | ((unsigned)p2[1] << 16), p2[2] | ((unsigned)p2[3] << 16));
    };
    auto store_b = [&](int stage) {
        unsigned short* Bs = smem + stage * STAGE + 2 * APL;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                *reinterpret_cast<ushort8*>(Bs + p * H2_BPL + b_st + i * 64 * H2_BK) = rb[i * 2 + p];
    };
    gload_a(0, 0);
    gload_b(0);
    gload_a(1, 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float* o = Ocsr + (size_t)o_pos[i] * P;
        float sum = 0.0f;
#pragma unroll
        for (int p = 0; p < 8; ++p) go[i][p] = 0.0f;
#pragma unroll
        for (int p = 0; p < PMAX; ++p) {
            const float ov = o[p < P ? p : P - 1];                 // (unconditional load, clamped: no branch)
            go[i][p] = p < P ? -ov : 0.0f;                         // 2 d = -(-2 d)
            sum += fabsf(go[i][p]);
        }
        a_exp[i] = h2_scale_exp(1.1f * wd2_absmax * sum);
        if constexpr (SUM) {
            const float* o2 = Ocsr + (size_t)o_pos2[i] * P;
            float sum2 = 0.0f;
#pragma unroll
            for (int p = 0; p < PMAX; ++p) {
                const float ov = o2[p < P ? p : P - 1];
                sum2 += p < P ? fabsf(ov) : 0.0f;
            }
            bnd_r[i] = bs.bound_c * (sum + sum2);
        }
    }

    floatx16 acc[1][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.0f;
    for (int idx = threadIdx.x; idx < 8 * KD; idx += 256) W2s[idx] = idx < P * KD ? Wd2[idx < P * KD ? idx : 0] : 0.0f;
    __syncthreads();                                              // (the staged weight is read by store_a below)
    store_a(0, 0, 0, 0);
    store_a(0, 0, 0, 1);
    store_b(0);
    gload_b(1);
    gload_a(2, 0);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const unsigned short* st = smem + (c & 1) * STAGE;
        const int nx = (c + 1) & 1;
        h2_kstep<1>(st, APL, st + 2 * APL, 0, wm * 32, wn * 64, acc);
        if (c + 1 < NCH) store_a(nx, c + 1, nx, 0);
        h2_kstep<1>(st, APL, st + 2 * APL, 1, wm * 32, wn * 64, acc);
        if (c + 1 < NCH) { store_a(nx, c + 1, nx, 1); store_b(nx); }
        if (c + 2 < NCH) gload_b(c + 2);
        if (c + 3 < NCH) gload_a(c + 3, nx);
        __syncthreads();
    }
    // epilogue through LDS: the accumulators are re-read as rows of float4, so the U gathers and the GZ stores are 128-byte row
    // segments; the thread that produced a row's A values reads it back, so its exponent is in a register
    float* Cs = reinterpret_cast<float*>(smem);
    // the U rows are requested before the accumulators go through LDS
    float4 ua[2][4], ub[2][4];
    auto uload = [&](int i) {
        const float* u0 = U + (size_t)ku0[i] * (2 * H) + s * H + n0 + lq * 4;
        const float* u1 = U + (size_t)ku1[i] * (2 * H) + s * H + n0 + lq * 4;
#pragma unroll
        for (int mcol = 0; mcol < 4; ++mcol) {
            ua[i][mcol] = *reinterpret_cast<const float4*>(u0 + 32 * mcol);
            ub[i][mcol] = *reinterpret_cast<const float4*>(u1 + 32 * mcol);
        }
    };
    if constexpr (SUM) {                                          // (requested here, not at entry: two registers less across the K loop keep three workgroups per CU)
        const int* blk = bs.blocks + (size_t)(bid >> 2) * ccsp::BS_BLK;
#pragma unroll
        for (int j = 0; j < 3; ++j) bsr[j] = blk[tid + 256 * j < ccsp::BS_BLK ? tid + 256 * j : ccsp::BS_BLK - 1];
    }
    uload(0);
    uload(1);
    if constexpr (SUM) {                                          // (the K loop ended with a barrier: the staged weight is dead)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (tid + 256 * j < ccsp::BS_BLK) bsb[tid + 256 * j] = bsr[j];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            Cs[row * C_LD + wn * 64 + j * 32 + (lane & 31)] = acc[0][j][r];
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = lr + 32 * i;
        const int k = e0 + row;
        const int e = -(a_exp[i] + wd_exp);
        float4 g[4];
#pragma unroll
        for (int mcol = 0; mcol < 4; ++mcol) {
            const float4 a = ua[i][mcol], b = ub[i][mcol];
            const float4 v = *reinterpret_cast<const float4*>(Cs + row * C_LD + lq * 4 + 32 * mcol);
            g[mcol] = make_float4(ldexpf(v.x, e) * silu_grad_fast(a.x + b.x), ldexpf(v.y, e) * silu_grad_fast(a.y + b.y),
                                  ldexpf(v.z, e) * silu_grad_fast(a.z + b.z), ldexpf(v.w, e) * silu_grad_fast(a.w + b.w));
        }
        if constexpr (SUM) {                                      // back into the tile (each element is read and written by this thread only)
#pragma unroll
            for (int mcol = 0; mcol < 4; ++mcol) *reinterpret_cast<float4*>(Cs + row * C_LD + lq * 4 + 32 * mcol) = g[mcol];
            if (lq == 0) Cs[row * C_LD + BN] = bnd_r[i];          // the edge's bound rides in the row's first padding column
        } else if (k < E_act) {
            float* gz = GZ + (size_t)k * (2 * H) + s * H + n0 + lq * 4;
#pragma unroll
            for (int mcol = 0; mcol < 4; ++mcol) *reinterpret_cast<float4*>(gz + 32 * mcol) = g[mcol];
        }
    }
    if constexpr (SUM) {
        if (tid < C_LD) Cs[BM * C_LD + tid] = 0.0f;                // the all-zero row that pads odd entry counts (bound 0 with it)
        __syncthreads();
        const int np = bsb[0];
        // exponents: one thread per partial row adds the bounds of its edges (the same sum in each of the block's four workgroups)
        if (tid < np) {
            const int span = bsb[129 + tid];
            float bsum = 0.0f;
            for (int q = span >> 16; q < (span & 0xffff); ++q) {
                const int2 r = *reinterpret_cast<const int2*>(bsb + 257 + 2 * q);
                bsum += Cs[(r.x >> 2) + BN];
                bsum += Cs[(r.y >> 2) + BN];
            }
            const int e = h2_scale_exp(bsum);
            pexp[tid] = e;
            if (s == 0 && ct == 0) bs.gexp[bsb[1 + tid]] = e;
        }
        __syncthreads();
        // sums: thread = (columns 4 cg .. + 3 and 64 + 4 cg .. + 3, one of 16 row lanes); a partial row's entries two at a time in
        // ascending edge order (byte offsets straight from the plan, the zero row instead of a mask)
        const int cg = tid & 15, rl = tid >> 4;
        const char* Cb = reinterpret_cast<const char*>(Cs) + cg * 16;
        for (int p = rl; p < np; p += 16) {
            const int span = bsb[129 + p];
            const int e = pexp[p];
            const int gid = bsb[1 + p];
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
            for (int q = span >> 16; q < (span & 0xffff); ++q) {
                const int2 r = *reinterpret_cast<const int2*>(bsb + 257 + 2 * q);
                const float4 v0 = *reinterpret_cast<const float4*>(Cb + r.x), v1 = *reinterpret_cast<const float4*>(Cb + r.x + 256);
                const float4 w0 = *reinterpret_cast<const float4*>(Cb + r.y), w1 = *reinterpret_cast<const float4*>(Cb + r.y + 256);
                a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
                a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
                a0.x += w0.x; a0.y += w0.y; a0.z += w0.z; a0.w += w0.w;
                a1.x += w1.x; a1.y += w1.y; a1.z += w1.z; a1.w += w1.w;
            }
            const float h[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            unsigned short p1[8], p2[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) split2h(ldexpf(h[k], e), p1[k], p2[k]);
#if CCSP_A_INTERLEAVED                                              // [NP][512 / 32][2][32]: both planes of a row's K chunk in one line (k_rowgemm_h2, ILA)
            const int col = s * H + n0 + cg * 4;
            const size_t o = (size_t)gid * (4 * H) + (size_t)(col >> 5) * 64 + (col & 31);
            *reinterpret_cast<uint2*>(bs.GZPH + o) = make_uint2(p1[0] | ((unsigned)p1[1] << 16), p1[2] | ((unsigned)p1[3] << 16));
            *reinterpret_cast<uint2*>(bs.GZPH + o + 128) = make_uint2(p1[4] | ((unsigned)p1[5] << 16), p1[6] | ((unsigned)p1[7] << 16));      // (column + 64 = two chunks on)
            *reinterpret_cast<uint2*>(bs.GZPH + o + 32) = make_uint2(p2[0] | ((unsigned)p2[1] << 16), p2[2] | ((unsigned)p2[3] << 16));
            *reinterpret_cast<uint2*>(bs.GZPH + o + 160) = make_uint2(p2[4] | ((unsigned)p2[5] << 16), p2[6] | ((unsigned)p2[7] << 16));
#else
            const size_t o = (size_t)gid * (2 * H) + s * H + n0 + cg * 4;
            *reinterpret_cast<uint2*>(bs.GZPH + o) = make_uint2(p1[0] | ((unsigned)p1[1] << 16), p1[2] | ((unsigned)p1[3] << 16));
            *reinterpret_cast<uint2*>(bs.GZPH + o + 64) = make_uint2(p1[4] | ((unsigned)p1[5] << 16), p1[6] | ((unsigned)p1[7] << 16));
            *reinterpret_cast<uint2*>(bs.GZPH + bs.plane + o) = make_uint2(p2[0] | ((unsigned)p2[1] << 16), p2[2] | ((unsigned)p2[3] << 16));
            *reinterpret_cast<uint2*>(bs.GZPH + bs.plane + o + 64) = make_uint2(p2[4] | ((unsigned)p2[5] << 16), p2[6] | ((unsigned)p2[7] << 16));
#endif
        }
    }
}
