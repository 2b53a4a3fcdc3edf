// StructDiffusion baseline (reference networks/denoise_fn.py:267-282,391-451, networks/transformer.py:43-82):
// every graph is a sequence of 8 tokens (padded), width Wd = 2H (3H with a grasp group), 4 pre-LN blocks,
// 2 heads.  Device layout: token row = graph * 8 + position, all activations row-major [M = 8 B, width].
//
// One evaluation = k_sd_embed, then per block  k_sd_ln -> k_sd_gemm(in_proj) -> k_sd_attn ->
// k_sd_gemm(out_proj, +residual) -> k_sd_gemm(c_fc, QuickGELU) -> k_sd_gemm(c_proj) -> k_sd_ln(+residual),
// then k_sd_decode (ln_post, last H channels, pose decoder, mask fill).  24 M Wd^2 flops per block dominate.
// Round 4: the four GEMMs of a block run on the f16 matrix pipe with the f16x2 scheme of ccsp_f16x2.h (k_sd_gemm_h2: three fp16
// products per fp32 product, fp32 accumulate; weights pre-split once per model with one exponent per tensor; the activation rows
// are split on the fly with one exponent per ROW taken from the row's largest magnitude, which the producing kernel leaves behind --
// the LayerNorm / attention kernels store it, the GEMM epilogues combine their column tiles' maxima by atomicMax on the float
// bits, an order-independent and therefore deterministic reduction).  k_sd_gemm (fp32 MFMA 32x32x2, round 1) stays for
// CCSP_MMA=f32 and for widths that are not a multiple of 128.
// Included by ccsp_hip.hip, after ccsp_f16x2.h.
#pragma once

constexpr int SD_L = 8;        // max_seq_len   (denoise_fn.py:272)
constexpr int SD_HEADS = 2;    // num_heads     (:273)
constexpr int SD_LAYERS = 4;   // num_layers    (:274)

enum { SD_EPI_BIAS = 0, SD_EPI_RESID = 1, SD_EPI_QGELU = 2 };

// C[M,N] (op)= A[M,K] . W[N,K]^T + bias.  Tile 64 x (64 TNW); 4 waves as 2 x 2, each 32 x (32 TNW).
// K % 32 == 0, N % (64 TNW) == 0; M arbitrary (rows clamped).
template <int TNW, int EPI>
__global__ __launch_bounds__(256) void k_sd_gemm(int M, int K, int N, const float* __restrict__ A, const float* __restrict__ W,
                                                 const float* __restrict__ bias, float* __restrict__ Cm) {
    constexpr int TN_ = 64 * TNW, BROWS = 2 * TNW;
    __shared__ float As[2][TILE_M * LDS_LD];
    __shared__ float Bs[2][TN_ * LDS_LD];
    const int nct = N / TN_;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = (bid / nct) * TILE_M, col0 = (bid % nct) * TN_;
    const int nrows = M - row0 < TILE_M ? M - row0 : TILE_M;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = tid >> 3, lq = tid & 7;
    const float* a_ptr[2];
    const float* b_ptr[BROWS];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int r = lr + 32 * i;
        r = r < nrows ? r : nrows - 1;
        a_ptr[i] = A + (size_t)(row0 + r) * K + lq * 4;
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) b_ptr[i] = W + (size_t)(col0 + lr + 32 * i) * K + lq * 4;
    float4 ra[2], rb[BROWS];
#pragma unroll
    for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const float4*>(a_ptr[i]);
#pragma unroll
    for (int i = 0; i < BROWS; ++i) rb[i] = *reinterpret_cast<const float4*>(b_ptr[i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) lds_store4(&As[0][(lr + 32 * i) * LDS_LD + lq * 4], ra[i]);
#pragma unroll
    for (int i = 0; i < BROWS; ++i) lds_store4(&Bs[0][(lr + 32 * i) * LDS_LD + lq * 4], rb[i]);
    __syncthreads();
    floatx16 acc[TNW];
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
        const float bv = bias[col0 + wn * 32 * TNW + j * 32 + (lane & 31)];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = bv;
    }
    const int nch = K / BK;
    for (int c = 0; c < nch; ++c) {
        const int buf = c & 1;
        if (c + 1 < nch) {
#pragma unroll
            for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const float4*>(a_ptr[i] + (c + 1) * BK);
#pragma unroll
            for (int i = 0; i < BROWS; ++i) rb[i] = *reinterpret_cast<const float4*>(b_ptr[i] + (c + 1) * BK);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk<TNW>(As[buf], Bs[buf], wm * 32, wn * 32 * TNW, acc);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < nch) {
#pragma unroll
            for (int i = 0; i < 2; ++i) lds_store4(&As[buf ^ 1][(lr + 32 * i) * LDS_LD + lq * 4], ra[i]);
#pragma unroll
            for (int i = 0; i < BROWS; ++i) lds_store4(&Bs[buf ^ 1][(lr + 32 * i) * LDS_LD + lq * 4], rb[i]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int col = col0 + wn * 32 * TNW + j * 32 + (lane & 31);
            if (row < nrows) {
                float v = acc[j][r];
                float* dst = Cm + (size_t)(row0 + row) * N + col;
                if (EPI == SD_EPI_RESID) v += *dst;
                if (EPI == SD_EPI_QGELU)                                   // x * sigmoid(1.702 x), transformer.py:38-40
                    v = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.44269504088896341f * v));
                *dst = v;
            }
        }
}

// wave-wide sum (all 64 lanes get the result)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

constexpr int SD_MAXV = 12;    // width <= 768: 12 values per lane

// nn.LayerNorm(eps 1e-5) of the row held as v[i] = x[lane + 64 i]; two-pass mean / biased variance.
// NV > 0: the width is 64 NV at compile time -- no bounds tests, so no branch sits next to a load (gfx950 counts loads and stores on
// one counter and hipcc's wait insertion takes the minimum over paths: with the tests every load of these kernels was waited for on
// its own; the LayerNorm kernels of the transformer took 16-19 us for 4 MB rows).  NV = 0: any width <= 64 SD_MAXV, run-time tests.
template <int NV = 0>
__device__ __forceinline__ void ln_row(float (&v)[NV ? NV : SD_MAXV], int Wd, int lane, const float* __restrict__ gam, const float* __restrict__ bet) {
    constexpr int N = NV ? NV : SD_MAXV;
    float gv[N], bv[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const bool in = NV || lane + 64 * i < Wd;
        gv[i] = in ? gam[NV || in ? lane + 64 * i : 0] : 0.0f;
        bv[i] = in ? bet[NV || in ? lane + 64 * i : 0] : 0.0f;
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) if (NV || lane + 64 * i < Wd) s += v[i];
    const float mean = wave_sum(s) / (float)Wd;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) if (NV || lane + 64 * i < Wd) { const float d = v[i] - mean; q += d * d; }
    const float inv = 1.0f / sqrtf(wave_sum(q) / (float)Wd + 1e-5f);
#pragma unroll
    for (int i = 0; i < N; ++i) if (NV || lane + 64 * i < Wd) v[i] = (v[i] - mean) * inv * gv[i] + bv[i];
}

// Y[r] = LN(X[r])  (ACC = 0)   or   Y[r] += LN(X[r])  (ACC = 1: x = x + ln_2(mlp(x)), transformer.py:66)
// ymax (f16x2 path, or null): bits of max |Y[r]| for the GEMM that reads Y; z0..z2: per-row maxima the GEMM epilogues / the attention
// kernel of THIS block accumulate by atomicMax, cleared here (their readers of the previous block are done: same stream)
// xparts > 1: X is the sum of that many split-K partial products [xparts][M][Wd], added in order
template <int ACC, int NV = 0>
__global__ __launch_bounds__(256) void k_sd_ln(int M, int Wd, const float* __restrict__ X, const float* __restrict__ gam,
                                               const float* __restrict__ bet, float* __restrict__ Y, unsigned int* __restrict__ ymax,
                                               unsigned int* __restrict__ z0, unsigned int* __restrict__ z1, unsigned int* __restrict__ z2, int xparts) {
    constexpr int N = NV ? NV : SD_MAXV;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    if (lane == 0) { if (z0) z0[row] = 0u; if (z1) z1[row] = 0u; if (z2) z2[row] = 0u; }
    float v[N], y0[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const bool in = NV || lane + 64 * i < Wd;
        const size_t o = (size_t)row * Wd + (in ? lane + 64 * i : 0);
        v[i] = in ? X[o] : 0.0f;
        y0[i] = (ACC && in) ? Y[o] : 0.0f;
    }
    for (int k = 1; k < xparts; ++k) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const bool in = NV || lane + 64 * i < Wd;
            v[i] += in ? X[((size_t)k * M + row) * Wd + (in ? lane + 64 * i : 0)] : 0.0f;
        }
    }
    ln_row<NV>(v, Wd, lane, gam, bet);
    unsigned int b = 0u;                                          // largest |Y| as float bits (NaN ranks above Inf and reaches the exponent as NaN)
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (NV || lane + 64 * i < Wd) {
            const float o = ACC ? y0[i] + v[i] : v[i];
            Y[(size_t)row * Wd + lane + 64 * i] = o;
            const unsigned int ob = __float_as_uint(o) & 0x7fffffffu;
            b = b > ob ? b : ob;
        }
    if (ymax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned int t = (unsigned int)__shfl_xor((int)b, o); b = b > t ? b : t; }
        if (lane == 0) ymax[row] = b;
    }
}

// x = X[r] + LN_2(Y[r]) (the end of block l, transformer.py:66) and at once Y[r] = LN_1'(x) of block l + 1: one launch and one pass
// over the row instead of two (k_sd_ln<1> then k_sd_ln<0>); same arithmetic in the same order as the two kernels.
template <int NV = 0>
__global__ __launch_bounds__(256) void k_sd_ln2ln1(int M, int Wd, float* __restrict__ X, const float* __restrict__ g2, const float* __restrict__ b2,
                                                   const float* __restrict__ g1, const float* __restrict__ b1, float* __restrict__ Y,
                                                   unsigned int* __restrict__ ymax, unsigned int* __restrict__ z0, unsigned int* __restrict__ z1,
                                                   unsigned int* __restrict__ z2, int yparts) {
    constexpr int N = NV ? NV : SD_MAXV;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    if (lane == 0) { if (z0) z0[row] = 0u; if (z1) z1[row] = 0u; if (z2) z2[row] = 0u; }
    float v[N], x0[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const bool in = NV || lane + 64 * i < Wd;
        const size_t o = (size_t)row * Wd + (in ? lane + 64 * i : 0);
        v[i] = in ? Y[o] : 0.0f;
        x0[i] = in ? X[o] : 0.0f;
    }
    for (int k = 1; k < yparts; ++k) {                             // split-K partials, in order
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const bool in = NV || lane + 64 * i < Wd;
            v[i] += in ? Y[((size_t)k * M + row) * Wd + (in ? lane + 64 * i : 0)] : 0.0f;
        }
    }
    ln_row<NV>(v, Wd, lane, g2, b2);
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (NV || lane + 64 * i < Wd) {
            v[i] = x0[i] + v[i];
            X[(size_t)row * Wd + lane + 64 * i] = v[i];
        }
    ln_row<NV>(v, Wd, lane, g1, b1);
    unsigned int b = 0u;
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (NV || lane + 64 * i < Wd) {
            Y[(size_t)row * Wd + lane + 64 * i] = v[i];
            const unsigned int ob = __float_as_uint(v[i]) & 0x7fffffffu;
            b = b > ob ? b : ob;
        }
    if (ymax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned int t = (unsigned int)__shfl_xor((int)b, o); b = b > t ? b : t; }
        if (lane == 0) ymax[row] = b;
    }
}

// ------------------------------------------------------------------------------------------
// k_sd_gemm_h2<EPI, TN>: C[M,N] (op)= A[M,K] . W[N,K]^T + bias on the f16 pipe (f16x2).  64 x TN tiles (TN = 128: 32 x 64 per wave;
// TN = 64: 32 x 32 per wave -- for the N = Wd GEMMs, whose 128-wide tile list would leave half the chip empty), 4 waves as 2 x 2,
// K chunks of 32 double-buffered through LDS, operands requested PD (2 or 4) chunks ahead into PD register sets (with one chunk of
// lead every iteration waited an L2 round trip: 1.5 k cycles per chunk against 0.4 k of MFMA work, first build of this round).
// A is fp32 in memory: a row's exponent comes from amax (bits of its largest magnitude, left by the producer), the split into two
// fp16 planes happens while the chunk is staged.  W: planes [N][K / 32][2][32] scaled by 2^w_exp.  Epilogue per wave through a wave-private
// LDS tile (rows re-read as 16-byte segments): bias, residual / QuickGELU, 16-byte stores, and -- cmax non-null -- the row maxima
// of the result for the next GEMM (DPP maximum over the lanes of a row, one atomicMax on the float bits per row and wave).
// K % 64 == 0, N % TN == 0; M arbitrary (rows clamped).
// ------------------------------------------------------------------------------------------
// KS (round 6): K chunks per STEP of the loop.  A phase trace of this kernel (tools/trace_sd_run.py, profiles/r06_trace_sd_gemm_*.txt) puts 80 % of a
// workgroup's life in the K loop at 820 .. 1760 cycles per 32-wide chunk for 190 .. 380 cycles of MFMA issue: with one wave per SIMD every chunk
// pays its own chain -- wait for the set, split, ds_write, barrier, fragment reads, six dependent MFMAs -- in full.  KS = 2 stages TWO chunks per
// step (a stage is two sub-stages of the KS = 1 layout): half the barriers and waits, twice the MFMAs behind each, the fragment reads of four
// k-steps free to be issued ahead of the first MFMA.  Same products in the same order per accumulator.
template <int EPI, int TN, int PD, int KS = 1>
__global__ __launch_bounds__(256, 2) void k_sd_gemm_h2(int M, int K, int N, const float* __restrict__ A, const unsigned int* __restrict__ amax,
                                                       const unsigned short* __restrict__ WH, size_t w_plane, int w_exp,
                                                       const float* __restrict__ bias, float* __restrict__ Cm, unsigned int* __restrict__ cmax) {
    constexpr int NJ = TN / 64;                                   // 32-column MFMA tiles per wave
    constexpr int TRK_ID = EPI == SD_EPI_BIAS ? 0 : (EPI == SD_EPI_QGELU ? 1 : 2);      // (trace builds, tools/trace_sd_run.py; the direct-mode kernels that own these slots do not run here)
    (void)TRK_ID;
    CCSP_TRK(TRK_ID, 0);
    CCSP_TRK_RT(TRK_ID, 30);
    constexpr int APL = 64 * H2_BK, BPL = TN * H2_BK, SUB = 2 * APL + 2 * BPL, STAGE = KS * SUB;
    constexpr int NB = TN / 64;                                   // B pieces (16 bytes) per thread and plane
    constexpr int CW_LD = 32 * NJ + 4, CW_SZ = 32 * CW_LD;        // wave-private epilogue tile [32][CW_LD] floats
    constexpr int SMEM_US = 2 * STAGE * 2 > 4 * CW_SZ * 4 ? 2 * STAGE : 4 * CW_SZ * 2;
    __shared__ __attribute__((aligned(16))) unsigned short smem[SMEM_US + 2 * 64];
    int* sE = reinterpret_cast<int*>(smem + SMEM_US);
    const int nct = N / TN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = (bid / nct) * 64, col0 = (bid % nct) * TN;
    const int nrows = M - row0 < 64 ? M - row0 : 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    // split K (gridDim.y slices; EPI = BIAS only): slice y multiplies K columns [y Ks, (y + 1) Ks) and writes its partial product to
    // Cm + y M N (the bias rides on slice 0); the LayerNorm kernel that reads the result adds the slices in order -- deterministic,
    // and a K = 4 Wd GEMM with one 64 x 64 tile per CU becomes four times as many workgroups of a quarter of the chain
    const int Ks = K / (int)gridDim.y, k_first = (int)blockIdx.y * Ks;
    Cm += (size_t)blockIdx.y * M * N;
    const int lr = tid >> 3, lq = tid & 7;                        // A producer: rows lr, lr + 32, fp32 columns 4 lq .. + 3 of the chunk
    const float* a_ptr[2];
    const unsigned int* a_mx[2];
    int a_exp[2], a_st[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int r = lr + 32 * i;
        r = r < nrows ? r : nrows - 1;
        a_ptr[i] = A + (size_t)(row0 + r) * K + k_first + lq * 4;
        a_mx[i] = amax + row0 + r;
        a_st[i] = h2_off(lr + 32 * i, lq >> 1) + (lq & 1) * 4;
    }
    const int brow = tid >> 2, bq = tid & 3;                      // B copy: rows brow (+ 64), piece bq, both planes
    // (W: chunk-interleaved planes [N][K / 32][2][32] -- both fp16 planes of a row's K chunk in ONE 128-byte line, k_interleave_planes)
    const unsigned short* b_ptr = WH + (size_t)(col0 + brow) * (2 * K) + 2 * k_first + bq * 8;
    const int b_st = h2_off(brow, bq);
    // operand loads by inline asm, waited for by counted s_waitcnt (h2_ld16): hipcc sinks an ordinary load to its first use -- the
    // ds_write of the NEXT trip -- which puts a memory round trip into every chunk (measured: 1.8 k cycles per chunk against 0.2 k of
    // MFMA work, whatever the prefetch distance written in the source)
    h2_f4 ra[PD][2 * KS];                                         // [register set][chunk of the step][row]
    h2_f4 rb[PD][2 * NB * KS];
    constexpr int NLD = KS * (2 + 2 * NB);                        // loads per step and thread
    // Every step of the chain issues the same loads and the same wait, with NO branch around either: chunks past the end of the slice are
    // requested as dummies (every lane the slice's first 16 bytes: one request per instruction).  A wait inside `if (more chunks)` made
    // hipcc merge the two paths' register sets with v_mov copies placed BEFORE the s_waitcnt of one path -- copies of registers whose
    // loads were in flight (second build of this round: NaN in every transformer parity test).
    const float* const a_dummy = A + (size_t)row0 * K + k_first;
    const unsigned short* const b_dummy = WH + (size_t)col0 * (2 * K) + 2 * k_first;
    // (c: index of the STEP -- KS consecutive chunks)
    auto gload = [&](int c, int set) {
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const int ch = c * KS + q;
            const bool real = ch < Ks / H2_BK;
#pragma unroll
            for (int i = 0; i < 2; ++i) h2_ld16(ra[set][q * 2 + i], real ? a_ptr[i] + ch * H2_BK : a_dummy);
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    h2_ld16(rb[set][q * 2 * NB + i * 2 + p], reinterpret_cast<const float*>(real ? b_ptr + (size_t)p * H2_BK + (size_t)i * 64 * (2 * K) + ch * (2 * H2_BK) : b_dummy));
        }
    };
    // the set's loads have landed (N younger loads may stay in flight); every register of the set is named: the compiler must not touch one before
#define CCSP_SD_GWAIT(set, N)                                                                                                                        \
    do {                                                                                                                                             \
        _Pragma("unroll") for (int q_ = 0; q_ < KS; ++q_) {                                                                                          \
            if (NB == 1) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ra[set][q_ * 2]), "+v"(ra[set][q_ * 2 + 1]), "+v"(rb[set][q_ * 2]), "+v"(rb[set][q_ * 2 + 1]) : "n"(N) : "memory"); \
            else asm volatile("s_waitcnt vmcnt(%6)" : "+v"(ra[set][q_ * 2]), "+v"(ra[set][q_ * 2 + 1]), "+v"(rb[set][(q_ * 4) % (2 * NB * KS)]), "+v"(rb[set][(q_ * 4 + 1) % (2 * NB * KS)]), \
                              "+v"(rb[set][(q_ * 4 + 2) % (2 * NB * KS)]), "+v"(rb[set][(q_ * 4 + 3) % (2 * NB * KS)]) : "n"(N) : "memory");        \
        }                                                                                                                                            \
    } while (0)
    // ... with the PD - 1 sets requested after it still in flight
    auto gwait = [&](int set) { CCSP_SD_GWAIT(set, (PD - 1) * NLD); };
    static_assert(PD == 2 || PD == 4, "prefetch depth");
    static_assert(NB <= 2, "gwait names four B registers per chunk");
    auto lstore = [&](int stage, int set) {
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            unsigned short* As = smem + stage * STAGE + q * SUB;
            unsigned short* Bs = As + 2 * APL;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float h[4] = {ra[set][q * 2 + i][0], ra[set][q * 2 + i][1], ra[set][q * 2 + i][2], ra[set][q * 2 + i][3]};
                unsigned short p1[4], p2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split2h(ldexpf(h[e], a_exp[i]), p1[e], p2[e]);
                unsigned short* d = As + a_st[i];
                *reinterpret_cast<uint2*>(d) = make_uint2(p1[0] | ((unsigned)p1[1] << 16), p1[2] | ((unsigned)p1[3] << 16));
                *reinterpret_cast<uint2*>(d + APL) = make_uint2(p2[0] | ((unsigned)p2[1] << 16), p2[2] | ((unsigned)p2[3] << 16));
            }
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) *reinterpret_cast<h2_f4*>(Bs + p * BPL + b_st + i * 64 * H2_BK) = rb[set][q * 2 * NB + i * 2 + p];
        }
    };
    floatx16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    // one k-step (16) of a staged chunk: same product order per accumulator as h2_kstep
    auto kstep = [&](const unsigned short* st, int ks) {
        const int piece = (lane >> 5) + 2 * ks;
        half8 a[2], b[NJ][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            a[p] = *reinterpret_cast<const half8*>(st + p * APL + h2_off(wm * 32 + (lane & 31), piece));
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j][p] = *reinterpret_cast<const half8*>(st + 2 * APL + p * BPL + h2_off(wn * 32 * NJ + 32 * j + (lane & 31), piece));
        }
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[PA[q]], b[j][PB[q]], acc[j], 0, 0, 0);
    };
    const int nch = Ks / (H2_BK * KS);                            // steps (a multiple of PD: the host picks PD and KS)
#pragma unroll
    for (int u = 0; u < PD; ++u) gload(u, u);
    {   // the rows' maxima ride behind the first operand requests (as an ordinary load in front of them they cost every workgroup a
        // second memory round trip before its first chunk -- and hipcc's wait for them, blind to the asm loads, drained those too)
        unsigned int mx[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("global_load_dword %0, %1, off" : "=v"(mx[i]) : "v"(a_mx[i]) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(mx[0]), "+v"(mx[1]) :: "memory");
        CCSP_SD_GWAIT(0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) a_exp[i] = h2_scale_exp(__uint_as_float(mx[i]));
    }
    CCSP_TRK(TRK_ID, 1);
    lstore(0, 0);
    gload(PD, 0);
    if (lq == 0) { sE[lr] = a_exp[0]; sE[lr + 32] = a_exp[1]; }  // (here and not where a_exp is loaded: its wait would sit in front of the first operand requests)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    CCSP_TRK(TRK_ID, 2);
    for (int c0 = 0; c0 < nch; c0 += PD) {                        // PD chunks per trip: register-set and stage indices are constants
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int c = c0 + u;                                 // step multiplied now, from stage u & 1; step c + 1 is staged behind it
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                kstep(smem + (u & 1) * STAGE + q * SUB, 0);       // (behind the last step: a dummy, into the stage nobody reads again)
                kstep(smem + (u & 1) * STAGE + q * SUB, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            gwait((u + 1) % PD);
            lstore((u + 1) & 1, (u + 1) % PD);
            gload(c + 1 + PD, (u + 1) % PD);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (not __syncthreads(): its fence is a vmcnt(0) that drains the prefetch)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    CCSP_TRK(TRK_ID, 3);
    // the dummies still in flight land before their registers are handed to the epilogue
#pragma unroll
    for (int u = 0; u < PD; ++u) CCSP_SD_GWAIT(u, 0);
#undef CCSP_SD_GWAIT
    CCSP_TRK(TRK_ID, 4);
    // epilogue, one wave at a time through its private LDS tile (the stages are free: every wave is past the last barrier)
    float* Cw = reinterpret_cast<float*>(smem) + wave * CW_SZ;
    constexpr int LPR = 8 * NJ;                                   // lanes per row segment: 4 columns each
    const int er = lane / LPR, eq = lane % LPR;                   // rows er + (64 / LPR) st, columns 4 eq
    const int colw = col0 + wn * 32 * NJ;
    float4 bv = *reinterpret_cast<const float4*>(bias + colw + 4 * eq);
    if (blockIdx.y != 0) bv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            Cw[rr * CW_LD + j * 32 + (lane & 31)] = acc[j][r];
        }
    asm volatile("" ::: "memory");                                // (compiler ordering only: one wave's LDS operations execute in order)
    CCSP_TRK(TRK_ID, 5);
    constexpr int RPP = 64 / LPR;                                 // rows per pass
#pragma unroll
    for (int st = 0; st < 32 / RPP; ++st) {
        const int trow = wm * 32 + er + RPP * st;
        const int e = -(sE[trow] + w_exp);
        const bool live = trow < nrows;
        float* dst = Cm + (size_t)(row0 + (live ? trow : 0)) * N + colw + 4 * eq;
        const float4 v = *reinterpret_cast<const float4*>(Cw + (er + RPP * st) * CW_LD + 4 * eq);
        float o[4] = {ldexpf(v.x, e) + bv.x, ldexpf(v.y, e) + bv.y, ldexpf(v.z, e) + bv.z, ldexpf(v.w, e) + bv.w};
        if (EPI == SD_EPI_RESID) {
            const float4 x = *reinterpret_cast<const float4*>(dst);
            o[0] += x.x; o[1] += x.y; o[2] += x.z; o[3] += x.w;
        }
        if (EPI == SD_EPI_QGELU) {                                // x * sigmoid(1.702 x), transformer.py:38-40
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = o[q] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.44269504088896341f * o[q]));
        }
        if (live) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        if (cmax) {
            // largest |o| of the row as float bits: integer maximum (non-negative floats order like their bits; NaN ranks above Inf)
            unsigned int b = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const unsigned int ob = __float_as_uint(o[q]) & 0x7fffffffu; b = b > ob ? b : ob; }
            unsigned int t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xF, 0xF, true); b = b > t ? b : t;        // quad_perm [1,0,3,2]
            t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)b, 0x4E, 0xF, 0xF, true); b = b > t ? b : t;                     // quad_perm [2,3,0,1]
            t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)b, 0x141, 0xF, 0xF, true); b = b > t ? b : t;                    // row_half_mirror: 8 lanes
            if (LPR == 16) { t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)b, 0x140, 0xF, 0xF, true); b = b > t ? b : t; }  // row_mirror: 16 lanes
            if (eq == 0 && live) atomicMax(cmax + row0 + trow, b);
        }
    }
    CCSP_TRK(TRK_ID, 6);
    CCSP_TRK_RT(TRK_ID, 31);
}

#ifdef CCSP_EXPERIMENTS
// ------------------------------------------------------------------------------------------
// k_sd_gemm_h2w<EPI> (round 6): the same GEMM on 128 x 128 tiles -- 4 waves as 2 x 2, 64 x 64 per wave (k_rowgemm_h2's wave tile).
// Why: the 64 x 64 tile of k_sd_gemm_h2 asks the CU's address unit for 16 KB of operands and its LDS for 16 KB of stores + 32 KB of fragment
// reads per 64 x 64 x 32 chunk -- with three workgroups per CU that is ~1200 address-unit cycles (four 16-byte requests per wave at ~25 cycles
// each, profiles/r05_ta_probe.txt) and ~1080 LDS cycles per chunk and CU against 576 of MFMA, and the measured chunk is 1040 cycles
// (profiles/r06_trace_sd_gemm_1lane.txt, r06_findings.md section 3).  A 128 x 128 tile moves twice the bytes for four times the products:
// 8 requests, 8 staging stores and 16 fragment reads per wave and chunk for 24 MFMAs (768 cycles).  Same split, same scaling, same order of the
// three products per accumulator as k_sd_gemm_h2; the tile lists of a 2048-row batch are 192 .. 256 workgroups (c_proj: four K slices).
// K % 64 == 0, N % 128 == 0; M arbitrary (rows clamped).
// MEASURED SLOWER than the narrow kernel it was built to replace (same call, 256 graphs x 8 tokens, two lanes): 60.3 samples/s narrow, 51.0 wide,
// 57.0 wide + PIPE; per launch 22.8 us (21.1 with PIPE) against 23.0 for in_proj / c_proj -- i.e. a workgroup of four times the products lives four
// times as long: with ONE workgroup per CU nothing overlaps a wave's own chain, and interleaving the staging between the MFMA pairs recovers 8 % of it.
// Experiments build only (CCSP_SD_TILE=wide); tests/test_experiments.py keeps it correct.  profiles/r06_findings.md section 4.
// ------------------------------------------------------------------------------------------
// PIPE: the staging of chunk c + 1 (wait for its registers, split, LDS stores) and the requests for chunk c + 3 are issued BETWEEN the MFMA pairs of
// chunk c (h2_chunk_ahead_with: every fragment read of the chunk up front, twelve slots of two MFMAs each).  With one workgroup per CU -- 192 .. 256
// tiles on 256 CUs -- a wave has nobody to overlap with but itself: without PIPE a chunk is requests + wait + 96 VALU instructions + stores +
// barrier + fragment reads + 24 MFMAs one after the other, 3.0 k cycles for 768 of matrix work (same duration as the narrow kernel: measured).
template <int EPI, bool PIPE = false>
__global__ __launch_bounds__(256, 2) void k_sd_gemm_h2w(int M, int K, int N, const float* __restrict__ A, const unsigned int* __restrict__ amax,
                                                        const unsigned short* __restrict__ WH, size_t w_plane, int w_exp,
                                                        const float* __restrict__ bias, float* __restrict__ Cm, unsigned int* __restrict__ cmax) {
    constexpr int TM = 128, TN = 128, MI = 2, NJ = 2, PD = 2;
    constexpr int APL = TM * H2_BK, BPL = TN * H2_BK, STAGE = 2 * APL + 2 * BPL;
    constexpr int NA = TM / 32;                                   // A rows per thread and chunk
    constexpr int CW_LD = 32 * NJ + 4, CW_SZ = 32 * CW_LD;        // wave-private epilogue tile [32][CW_LD] floats
    static_assert(2 * STAGE * 2 >= 4 * CW_SZ * 4, "epilogue tiles fit the stages");
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * STAGE + 2 * TM];
    int* sE = reinterpret_cast<int*>(smem + 2 * STAGE);
    (void)w_plane;
    const int nct = N / TN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = (bid / nct) * TM, col0 = (bid % nct) * TN;
    const int nrows = M - row0 < TM ? M - row0 : TM;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int Ks = K / (int)gridDim.y, k_first = (int)blockIdx.y * Ks;      // split K as in k_sd_gemm_h2
    Cm += (size_t)blockIdx.y * M * N;
    const int lr = tid >> 3, lq = tid & 7;                        // A producer: rows lr + 32 i, fp32 columns 4 lq .. + 3 of the chunk
    const float* a_ptr[NA];
    const unsigned int* a_mx[NA];
    int a_exp[NA], a_st[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int r = lr + 32 * i;
        r = r < nrows ? r : nrows - 1;
        a_ptr[i] = A + (size_t)(row0 + r) * K + k_first + lq * 4;
        a_mx[i] = amax + row0 + r;
        a_st[i] = h2_off(lr + 32 * i, lq >> 1) + (lq & 1) * 4;
    }
    const int brow = tid >> 2, bq = tid & 3;                      // B copy: rows brow, brow + 64, piece bq, both planes
    const unsigned short* b_ptr = WH + (size_t)(col0 + brow) * (2 * K) + 2 * k_first + bq * 8;
    const int b_st = h2_off(brow, bq);
    h2_f4 ra[PD][NA];
    h2_f4 rb[PD][4];
    constexpr int NLD = NA + 4;
    const float* const a_dummy = A + (size_t)row0 * K + k_first;
    const unsigned short* const b_dummy = WH + (size_t)col0 * (2 * K) + 2 * k_first;
    auto gload = [&](int c, int set) {                            // (no branch around a load or a wait: see k_sd_gemm_h2)
        const bool real = c < Ks / H2_BK;
#pragma unroll
        for (int i = 0; i < NA; ++i) h2_ld16(ra[set][i], real ? a_ptr[i] + c * H2_BK : a_dummy);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                h2_ld16(rb[set][i * 2 + p], reinterpret_cast<const float*>(real ? b_ptr + (size_t)p * H2_BK + (size_t)i * 64 * (2 * K) + c * (2 * H2_BK) : b_dummy));
    };
#define CCSP_SDW_GWAIT(set, NN)                                                                                                                      \
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(ra[set][0]), "+v"(ra[set][1]), "+v"(ra[set][2]), "+v"(ra[set][3]), "+v"(rb[set][0]), "+v"(rb[set][1]), \
                 "+v"(rb[set][2]), "+v"(rb[set][3]) : "n"(NN) : "memory")
    auto lstore_a = [&](int stage, int set, int i) {              // one A row of the set: split, two 8-byte stores
        unsigned short* As = smem + stage * STAGE;
        const float h[4] = {ra[set][i][0], ra[set][i][1], ra[set][i][2], ra[set][i][3]};
        unsigned short p1[4], p2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split2h(ldexpf(h[e], a_exp[i]), p1[e], p2[e]);
        unsigned short* d = As + a_st[i];
        *reinterpret_cast<uint2*>(d) = make_uint2(p1[0] | ((unsigned)p1[1] << 16), p1[2] | ((unsigned)p1[3] << 16));
        *reinterpret_cast<uint2*>(d + APL) = make_uint2(p2[0] | ((unsigned)p2[1] << 16), p2[2] | ((unsigned)p2[3] << 16));
    };
    auto lstore_b = [&](int stage, int set, int i) {              // both planes of weight rows brow + 64 i
        unsigned short* Bs = smem + stage * STAGE + 2 * APL;
#pragma unroll
        for (int p = 0; p < 2; ++p) *reinterpret_cast<h2_f4*>(Bs + p * BPL + b_st + i * 64 * H2_BK) = rb[set][i * 2 + p];
    };
    auto lstore = [&](int stage, int set) {
#pragma unroll
        for (int i = 0; i < NA; ++i) lstore_a(stage, set, i);
#pragma unroll
        for (int i = 0; i < 2; ++i) lstore_b(stage, set, i);
    };
    auto gload_part = [&](int c, int set, int part) {             // a quarter of gload(c, set): A row `part`, and weight piece `part`
        const bool real = c < Ks / H2_BK;
        h2_ld16(ra[set][part], real ? a_ptr[part] + c * H2_BK : a_dummy);
        const int i = part >> 1, pl = part & 1;
        h2_ld16(rb[set][part], reinterpret_cast<const float*>(real ? b_ptr + (size_t)pl * H2_BK + (size_t)i * 64 * (2 * K) + c * (2 * H2_BK) : b_dummy));
    };
    floatx16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int nch = Ks / H2_BK;                                   // (even: K % 64 == 0)
    gload(0, 0);
    gload(1, 1);
    {
        unsigned int mx[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) asm volatile("global_load_dword %0, %1, off" : "=v"(mx[i]) : "v"(a_mx[i]) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(mx[0]), "+v"(mx[1]), "+v"(mx[2]), "+v"(mx[3]) :: "memory");
        CCSP_SDW_GWAIT(0, 0);
#pragma unroll
        for (int i = 0; i < NA; ++i) a_exp[i] = h2_scale_exp(__uint_as_float(mx[i]));
    }
    lstore(0, 0);
    gload(2, 0);
#pragma unroll
    for (int i = 0; i < NA; ++i)
        if (lq == 0) sE[lr + 32 * i] = a_exp[i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    for (int c0 = 0; c0 < nch; c0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned short* st = smem + (u & 1) * STAGE;
            if constexpr (PIPE) {
                h2_chunk_ahead_with<MI>(st, APL, st + 2 * APL, wm * 64, wn * 64, acc, [&](int k) {
                    // (k and u are constants after unrolling: every slot is straight-line code, no branch around a wait or a load)
                    if (k == 0) { if (u == 0) CCSP_SDW_GWAIT(1, NLD); else CCSP_SDW_GWAIT(0, NLD); }
                    else if (k <= 4) lstore_a((u + 1) & 1, (u + 1) & 1, k - 1);
                    else if (k <= 6) lstore_b((u + 1) & 1, (u + 1) & 1, k - 5);
                    else if (k <= 10) gload_part(c0 + u + 3, (u + 1) & 1, k - 7);
                });
            } else {
                h2_kstep<MI>(st, APL, st + 2 * APL, 0, wm * 64, wn * 64, acc);
                h2_kstep<MI>(st, APL, st + 2 * APL, 1, wm * 64, wn * 64, acc);
                __builtin_amdgcn_sched_barrier(0);
                if (u == 0) CCSP_SDW_GWAIT(1, NLD); else CCSP_SDW_GWAIT(0, NLD);
                lstore((u + 1) & 1, (u + 1) & 1);
                gload(c0 + u + 3, (u + 1) & 1);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (not __syncthreads(): its fence is a vmcnt(0) that drains the prefetch)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    CCSP_SDW_GWAIT(0, 0);
    CCSP_SDW_GWAIT(1, 0);
#undef CCSP_SDW_GWAIT
    // epilogue, one wave at a time through its private LDS tile, one 32-row tile after the other
    float* Cw = reinterpret_cast<float*>(smem) + wave * CW_SZ;
    constexpr int LPR = 8 * NJ;
    const int er = lane / LPR, eq = lane % LPR;
    const int colw = col0 + wn * 32 * NJ;
    float4 bv = *reinterpret_cast<const float4*>(bias + colw + 4 * eq);
    if (blockIdx.y != 0) bv = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int RPP = 64 / LPR;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                Cw[rr * CW_LD + j * 32 + (lane & 31)] = acc[mi][j][r];
            }
        asm volatile("" ::: "memory");                            // (compiler ordering only: one wave's LDS operations execute in order)
#pragma unroll
        for (int st = 0; st < 32 / RPP; ++st) {
            const int trow = wm * 64 + 32 * mi + er + RPP * st;
            const int e = -(sE[trow] + w_exp);
            const bool live = trow < nrows;
            float* dst = Cm + (size_t)(row0 + (live ? trow : 0)) * N + colw + 4 * eq;
            const float4 v = *reinterpret_cast<const float4*>(Cw + (er + RPP * st) * CW_LD + 4 * eq);
            float o[4] = {ldexpf(v.x, e) + bv.x, ldexpf(v.y, e) + bv.y, ldexpf(v.z, e) + bv.z, ldexpf(v.w, e) + bv.w};
            if (EPI == SD_EPI_RESID) {
                const float4 x = *reinterpret_cast<const float4*>(dst);
                o[0] += x.x; o[1] += x.y; o[2] += x.z; o[3] += x.w;
            }
            if (EPI == SD_EPI_QGELU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = o[q] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.44269504088896341f * o[q]));
            }
            if (live) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            if (cmax) {
                unsigned int b = 0u;
#pragma unroll
                for (int q = 0; q < 4; ++q) { const unsigned int ob = __float_as_uint(o[q]) & 0x7fffffffu; b = b > ob ? b : ob; }
                unsigned int t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xF, 0xF, true); b = b > t ? b : t;
                t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)b, 0x4E, 0xF, 0xF, true); b = b > t ? b : t;
                t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)b, 0x141, 0xF, 0xF, true); b = b > t ? b : t;
                t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)b, 0x140, 0xF, 0xF, true); b = b > t ? b : t;      // LPR == 16
                if (eq == 0 && live) atomicMax(cmax + row0 + trow, b);
            }
        }
        asm volatile("" ::: "memory");
    }
}

// ------------------------------------------------------------------------------------------
// k_sd_gemm_h2x<EPI> (round 6): 128 x 128 tiles on EIGHT waves (512 threads, 4 x 2, 32 x 64 per wave).  k_sd_gemm_h2w showed that a 128-wide tile with
// one wave per SIMD cannot overlap its own staging with its own MFMAs; here a workgroup brings two waves per SIMD of its own, every thread stages
// half as much (two A rows, two weight pieces per chunk: 4 requests per wave for a quarter of a 128 x 128 x 32 chunk), and the tile still moves half
// the operand bytes per product of the 64-wide kernel.  Same split, scaling and product order per accumulator as k_sd_gemm_h2 (bitwise k_sd_gemm_h2w's
// results).  MEASURED (same call, CCSP_SD_TILE=wide8): 57.0 samples/s against 61.4 with the 64-row tiles -- like k_sd_gemm_h2w with PIPE.  Three
// organisations of the 128-wide tile end at the same place: what a chunk costs is the chain request -> wait -> split -> store -> barrier -> fragment
// read -> multiply, and many small co-resident workgroups hide it better than few large ones.  Experiments build.
// K % 64 == 0, N % 128 == 0; M arbitrary (rows clamped).
// ------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512, 1) void k_sd_gemm_h2x(int M, int K, int N, const float* __restrict__ A, const unsigned int* __restrict__ amax,
                                                        const unsigned short* __restrict__ WH, size_t w_plane, int w_exp,
                                                        const float* __restrict__ bias, float* __restrict__ Cm, unsigned int* __restrict__ cmax) {
    constexpr int TM = 128, TN = 128, NJ = 2, PD = 2, NW = 8;
    constexpr int APL = TM * H2_BK, BPL = TN * H2_BK, STAGE = 2 * APL + 2 * BPL;
    constexpr int CW_LD = 32 * NJ + 4, CW_SZ = 32 * CW_LD;        // wave-private epilogue tile [32][CW_LD] floats
    constexpr int SMEM_US = 2 * STAGE * 2 > NW * CW_SZ * 4 ? 2 * STAGE : NW * CW_SZ * 2;
    __shared__ __attribute__((aligned(16))) unsigned short smem[SMEM_US + 2 * TM];
    int* sE = reinterpret_cast<int*>(smem + SMEM_US);
    (void)w_plane;
    const int nct = N / TN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = (bid / nct) * TM, col0 = (bid % nct) * TN;
    const int nrows = M - row0 < TM ? M - row0 : TM;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;                      // rows 32 wm .. + 31, columns 64 wn .. + 63
    const int Ks = K / (int)gridDim.y, k_first = (int)blockIdx.y * Ks;
    Cm += (size_t)blockIdx.y * M * N;
    const int lr = tid >> 3, lq = tid & 7;                        // A producer: rows lr, lr + 64, fp32 columns 4 lq .. + 3 of the chunk
    const float* a_ptr[2];
    const unsigned int* a_mx[2];
    int a_exp[2], a_st[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int r = lr + 64 * i;
        r = r < nrows ? r : nrows - 1;
        a_ptr[i] = A + (size_t)(row0 + r) * K + k_first + lq * 4;
        a_mx[i] = amax + row0 + r;
        a_st[i] = h2_off(lr + 64 * i, lq >> 1) + (lq & 1) * 4;
    }
    const int brow = tid >> 2, bq = tid & 3;                      // B copy: row brow, piece bq, both planes
    const unsigned short* b_ptr = WH + (size_t)(col0 + brow) * (2 * K) + 2 * k_first + bq * 8;
    const int b_st = h2_off(brow, bq);
    h2_f4 ra[PD][2];
    h2_f4 rb[PD][2];
    constexpr int NLD = 4;
    const float* const a_dummy = A + (size_t)row0 * K + k_first;
    const unsigned short* const b_dummy = WH + (size_t)col0 * (2 * K) + 2 * k_first;
    auto gload = [&](int c, int set) {                            // (no branch around a load or a wait: see k_sd_gemm_h2)
        const bool real = c < Ks / H2_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) h2_ld16(ra[set][i], real ? a_ptr[i] + c * H2_BK : a_dummy);
#pragma unroll
        for (int p = 0; p < 2; ++p)
            h2_ld16(rb[set][p], reinterpret_cast<const float*>(real ? b_ptr + (size_t)p * H2_BK + c * (2 * H2_BK) : b_dummy));
    };
#define CCSP_SDX_GWAIT(set, NN) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ra[set][0]), "+v"(ra[set][1]), "+v"(rb[set][0]), "+v"(rb[set][1]) : "n"(NN) : "memory")
    auto lstore = [&](int stage, int set) {
        unsigned short* As = smem + stage * STAGE;
        unsigned short* Bs = As + 2 * APL;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float h[4] = {ra[set][i][0], ra[set][i][1], ra[set][i][2], ra[set][i][3]};
            unsigned short p1[4], p2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split2h(ldexpf(h[e], a_exp[i]), p1[e], p2[e]);
            unsigned short* d = As + a_st[i];
            *reinterpret_cast<uint2*>(d) = make_uint2(p1[0] | ((unsigned)p1[1] << 16), p1[2] | ((unsigned)p1[3] << 16));
            *reinterpret_cast<uint2*>(d + APL) = make_uint2(p2[0] | ((unsigned)p2[1] << 16), p2[2] | ((unsigned)p2[3] << 16));
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) *reinterpret_cast<h2_f4*>(Bs + p * BPL + b_st) = rb[set][p];
    };
    floatx16 acc[1][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.0f;
    const int nch = Ks / H2_BK;                                   // (even: K % 64 == 0)
    gload(0, 0);
    gload(1, 1);
    {
        unsigned int mx[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("global_load_dword %0, %1, off" : "=v"(mx[i]) : "v"(a_mx[i]) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(mx[0]), "+v"(mx[1]) :: "memory");
        CCSP_SDX_GWAIT(0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) a_exp[i] = h2_scale_exp(__uint_as_float(mx[i]));
    }
    lstore(0, 0);
    gload(2, 0);
    if (lq == 0) { sE[lr] = a_exp[0]; sE[lr + 64] = a_exp[1]; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    for (int c0 = 0; c0 < nch; c0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned short* st = smem + (u & 1) * STAGE;
            h2_kstep<1>(st, APL, st + 2 * APL, 0, wm * 32, wn * 64, acc);
            h2_kstep<1>(st, APL, st + 2 * APL, 1, wm * 32, wn * 64, acc);
            __builtin_amdgcn_sched_barrier(0);
            if (u == 0) CCSP_SDX_GWAIT(1, NLD); else CCSP_SDX_GWAIT(0, NLD);
            lstore((u + 1) & 1, (u + 1) & 1);
            gload(c0 + u + 3, (u + 1) & 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (not __syncthreads(): its fence is a vmcnt(0) that drains the prefetch)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    CCSP_SDX_GWAIT(0, 0);
    CCSP_SDX_GWAIT(1, 0);
#undef CCSP_SDX_GWAIT
    // epilogue: the stages are free (every wave is past the last barrier); one wave-private tile per wave
    float* Cw = reinterpret_cast<float*>(smem) + wave * CW_SZ;
    constexpr int LPR = 8 * NJ;
    const int er = lane / LPR, eq = lane % LPR;
    const int colw = col0 + wn * 32 * NJ;
    float4 bv = *reinterpret_cast<const float4*>(bias + colw + 4 * eq);
    if (blockIdx.y != 0) bv = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int RPP = 64 / LPR;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            Cw[rr * CW_LD + j * 32 + (lane & 31)] = acc[0][j][r];
        }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int st = 0; st < 32 / RPP; ++st) {
        const int trow = wm * 32 + er + RPP * st;
        const int e = -(sE[trow] + w_exp);
        const bool live = trow < nrows;
        float* dst = Cm + (size_t)(row0 + (live ? trow : 0)) * N + colw + 4 * eq;
        const float4 v = *reinterpret_cast<const float4*>(Cw + (er + RPP * st) * CW_LD + 4 * eq);
        float o[4] = {ldexpf(v.x, e) + bv.x, ldexpf(v.y, e) + bv.y, ldexpf(v.z, e) + bv.z, ldexpf(v.w, e) + bv.w};
        if (EPI == SD_EPI_RESID) {
            const float4 x = *reinterpret_cast<const float4*>(dst);
            o[0] += x.x; o[1] += x.y; o[2] += x.z; o[3] += x.w;
        }
        if (EPI == SD_EPI_QGELU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = o[q] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.44269504088896341f * o[q]));
        }
        if (live) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        if (cmax) {
            unsigned int b = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const unsigned int ob = __float_as_uint(o[q]) & 0x7fffffffu; b = b > ob ? b : ob; }
            unsigned int t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xF, 0xF, true); b = b > t ? b : t;
            t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)b, 0x4E, 0xF, 0xF, true); b = b > t ? b : t;
            t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)b, 0x141, 0xF, 0xF, true); b = b > t ? b : t;
            t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)b, 0x140, 0xF, 0xF, true); b = b > t ? b : t;
            if (eq == 0 && live) atomicMax(cmax + row0 + trow, b);
        }
    }
}

#endif  // CCSP_EXPERIMENTS

// token rows: [grasp_emb] geoms_emb (poses_emb + time_emb) + pe[position] -> ln_pre; padding rows are zero
// (denoise_fn.py:397-423)
template <int NV = 0>
__global__ __launch_bounds__(256) void k_sd_embed(int M, int H, int Wd, int grasp, const int* __restrict__ tok_node,
                                                  const int* __restrict__ tok_pos, const float* __restrict__ gemb,
                                                  const float* __restrict__ remb, const float* __restrict__ pemb,
                                                  const float* __restrict__ temb_t, const float* __restrict__ pe,
                                                  const float* __restrict__ gam, const float* __restrict__ bet, float* __restrict__ X,
                                                  // round 6: ln_1 of the FIRST block in the same pass over the row (k_sd_ln<0> of round 4: one launch less per
                                                  // evaluation; same arithmetic in the same order).  Y == nullptr: ln_pre only
                                                  const float* __restrict__ g1, const float* __restrict__ b1, float* __restrict__ Y, unsigned int* __restrict__ ymax,
                                                  unsigned int* __restrict__ z0, unsigned int* __restrict__ z1, unsigned int* __restrict__ z2) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    if (Y && lane == 0) { if (z0) z0[row] = 0u; if (z1) z1[row] = 0u; if (z2) z2[row] = 0u; }
    const int n = tok_node[row];
    constexpr int N = NV ? NV : SD_MAXV;                          // (NV > 0: the width is 64 NV at compile time, as in k_sd_ln)
    float v[N];
    if (n < 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) { v[i] = 0.0f; if (NV || lane + 64 * i < Wd) X[(size_t)row * Wd + lane + 64 * i] = 0.0f; }
        if (!Y) return;
    } else {
        const int off = grasp ? H : 0;
        const float* per = pe + (size_t)tok_pos[row] * Wd;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = lane + 64 * i;
            float e = 0.0f;
            if (NV || c < Wd) {
                if (c < off) e = remb[(size_t)n * H + c];
                else if (c < off + H) e = gemb[(size_t)n * H + c - off];
                else e = pemb[(size_t)n * H + c - off - H] + temb_t[c - off - H];
                e += per[c];
            }
            v[i] = e;
        }
        ln_row<NV>(v, Wd, lane, gam, bet);
#pragma unroll
        for (int i = 0; i < N; ++i) if (NV || lane + 64 * i < Wd) X[(size_t)row * Wd + lane + 64 * i] = v[i];
        if (!Y) return;
    }
    ln_row<NV>(v, Wd, lane, g1, b1);                                   // (a padding row is all zeros: its ln_1 is the bias, as k_sd_ln computes it)
    unsigned int b = 0u;
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (NV || lane + 64 * i < Wd) {
            Y[(size_t)row * Wd + lane + 64 * i] = v[i];
            const unsigned int ob = __float_as_uint(v[i]) & 0x7fffffffu;
            b = b > ob ? b : ob;
        }
    if (ymax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned int t = (unsigned int)__shfl_xor((int)b, o); b = b > t ? b : t; }
        if (lane == 0) ymax[row] = b;
    }
}

// nn.MultiheadAttention core for one (graph, head): 8 x 8 scores, the FLOAT pad mask added to them
// (denoise_fn.py:426-434: +1.0 where a row or column is padding; with no padding `[-0:]` marks
// everything).  mask_from[b * heads + h] = first padded index of the graph whose mask this head
// sees, (b heads + h) mod B -- the reference repeats the masks graph-major while MHA reads them
// head-major.
constexpr int SD_DH_MAX = 384;
// 16-byte accesses throughout (DH is a multiple of 32): 6 DH / 4 requests of a (graph, head)'s q, k, v issued at once, scores and P.V read
// as ds_read_b128; every sum keeps the element order of the scalar form it replaces (13.6 -> see profiles/r04_findings.md section 4).
// parts / part_stride: in_proj run as `parts` K slices (k_sd_gemm_h2, gridDim.y): QKV holds that many partial products, added here in
// slice order while they are loaded (deterministic; the bias rides on slice 0)
__global__ __launch_bounds__(256) void k_sd_attn(int Wd, const float* __restrict__ QKV, const int* __restrict__ mask_from,
                                                 float* __restrict__ Aout, unsigned int* __restrict__ amax /*[8 B] or null: atomicMax of |Aout| per token row*/,
                                                 int parts, size_t part_stride) {
    constexpr int LD = SD_DH_MAX + 4;                             // row stride: 16-byte aligned, consecutive rows 4 banks apart
    __shared__ __attribute__((aligned(16))) float qkv[3][SD_L][LD];
    __shared__ float part[4][SD_L * SD_L];
    __shared__ float ps[SD_L][SD_L];
    __shared__ unsigned int srow[SD_L];
    if (threadIdx.x < SD_L) srow[threadIdx.x] = 0u;
    const int b = blockIdx.x / SD_HEADS, h = blockIdx.x % SD_HEADS;
    const int DH = Wd / SD_HEADS, tid = threadIdx.x;
    const int D4 = DH >> 2;                                       // 16-byte pieces per row segment
    const float* base = QKV + (size_t)b * SD_L * 3 * Wd + h * DH;
    constexpr int NLD = (3 * SD_L * (SD_DH_MAX / 4) + 255) / 256; // 9 pieces per thread at most
    float4 ld[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int idx = tid + 256 * k;                            // (which, row, piece): which slowest
        const int w = idx / (SD_L * D4), rem = idx - w * (SD_L * D4);
        const int r = rem / D4, c4 = rem - r * D4;
        const bool in = idx < 3 * SD_L * D4;
        ld[k] = *reinterpret_cast<const float4*>(base + (in ? (size_t)r * 3 * Wd + (size_t)w * Wd + 4 * c4 : 0));
    }
    if (parts > 1) {                                              // (uniform)
        float4 l2[NLD];
        for (int q = 1; q < parts; ++q) {
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int idx = tid + 256 * k;
                const int w = idx / (SD_L * D4), rem = idx - w * (SD_L * D4);
                const int r = rem / D4, c4 = rem - r * D4;
                const bool in = idx < 3 * SD_L * D4;
                l2[k] = *reinterpret_cast<const float4*>(base + (size_t)q * part_stride + (in ? (size_t)r * 3 * Wd + (size_t)w * Wd + 4 * c4 : 0));
            }
#pragma unroll
            for (int k = 0; k < NLD; ++k) { ld[k].x += l2[k].x; ld[k].y += l2[k].y; ld[k].z += l2[k].z; ld[k].w += l2[k].w; }
        }
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int idx = tid + 256 * k;
        const int w = idx / (SD_L * D4), rem = idx - w * (SD_L * D4);
        const int r = rem / D4, c4 = rem - r * D4;
        if (idx < 3 * SD_L * D4) *reinterpret_cast<float4*>(&qkv[w][r][4 * c4]) = ld[k];
    }
    __syncthreads();
    // 64 (query, key) pairs x 4 quarters of the head dimension; the quarters are added in order 0..3
    const int pair = tid & 63, qt = tid >> 6;
    const int i = pair >> 3, j = pair & 7;
    const float scale = 1.0f / sqrtf((float)DH);          // q is scaled before the product, like F.multi_head_attention_forward
    {
        const int c0 = qt * (DH / 4), c1 = c0 + DH / 4;   // (DH / 4 is a multiple of 4 when DH % 16 == 0; else the scalar tail below)
        float sacc = 0.0f;
        int c = c0;
        for (; c + 4 <= c1 && (c0 & 3) == 0; c += 4) {
            const float4 q = *reinterpret_cast<const float4*>(&qkv[0][i][c]);
            const float4 kk = *reinterpret_cast<const float4*>(&qkv[1][j][c]);
            sacc += (q.x * scale) * kk.x; sacc += (q.y * scale) * kk.y; sacc += (q.z * scale) * kk.z; sacc += (q.w * scale) * kk.w;
        }
        for (; c < c1; ++c) sacc += (qkv[0][i][c] * scale) * qkv[1][j][c];
        part[qt][pair] = sacc;
    }
    __syncthreads();
    if (tid < 64) {
        float sc = ((part[0][pair] + part[1][pair]) + part[2][pair]) + part[3][pair];
        const int from = mask_from[blockIdx.x];
        sc += (i >= from || j >= from) ? 1.0f : 0.0f;
        float mx = sc;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float e = expf(sc - mx);
        float den = e;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) den += __shfl_xor(den, o);
        ps[i][j] = e / den;
    }
    __syncthreads();
    for (int idx = tid; idx < SD_L * D4; idx += 256) {
        const int r = idx / D4, c4 = idx - r * D4;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int jj = 0; jj < SD_L; ++jj) {
            const float pw = ps[r][jj];
            const float4 v = *reinterpret_cast<const float4*>(&qkv[2][jj][4 * c4]);
            o.x += pw * v.x; o.y += pw * v.y; o.z += pw * v.z; o.w += pw * v.w;
        }
        *reinterpret_cast<float4*>(Aout + ((size_t)b * SD_L + r) * Wd + h * DH + 4 * c4) = o;
        if (amax) {                                               // (LDS; integer maximum of the bits: order-independent, NaN on top)
            unsigned int m0 = __float_as_uint(o.x) & 0x7fffffffu, m1 = __float_as_uint(o.y) & 0x7fffffffu;
            const unsigned int m2 = __float_as_uint(o.z) & 0x7fffffffu, m3 = __float_as_uint(o.w) & 0x7fffffffu;
            m0 = m0 > m2 ? m0 : m2; m1 = m1 > m3 ? m1 : m3;
            atomicMax(&srow[r], m0 > m1 ? m0 : m1);
        }
    }
    if (amax) {
        __syncthreads();
        if (tid < SD_L) atomicMax(amax + (size_t)b * SD_L + tid, srow[tid]);      // (the two heads of a row, in any order)
    }
}

// per node: ln_post of its token row, last H channels -> pose_decoder -> eps; masked nodes take
// batch.x[:, -P:] (denoise_fn.py:437-449).
// Round 6: one wave per node in the row phases as before, but the first decoder layer is shared by the workgroup's four nodes: a thread owns one
// hidden unit for all four, so one weight load feeds four accumulators (round 1: every wave streamed the whole [H, H/2] weight for its own node
// through 2 H dependent load-multiply steps; 18.6 us per evaluation, as long as a GEMM -- profiles/r06_kernel_stats_sd_before.csv).  NV > 0: the
// width at compile time (no bounds tests next to the row loads, as in k_sd_ln).  Every sum keeps its order (c ascending per unit; the second layer's
// lane-strided partials and butterfly): results are bitwise those of the old kernel.  (Eight nodes per workgroup -- two per wave in the row
// phases -- was slower, 24.5 us: the row phase is the latency chain of this kernel.)
template <int H, int NV = 0>
__global__ __launch_bounds__(256) void k_sd_decode(int N, int Wd, int P, int F, const int* __restrict__ node_tok, const float* __restrict__ X,
                                                   const float* __restrict__ gam, const float* __restrict__ bet,
                                                   const float* __restrict__ pd0_wT /*[H][H/2]*/, const float* __restrict__ pd0_b,
                                                   const float* __restrict__ pd2_w /*[P][H/2]*/, const float* __restrict__ pd2_b,
                                                   const float* __restrict__ xfeat, const signed char* __restrict__ mask,
                                                   float* __restrict__ eps,
                                                   // the last block's  x = x + ln_2(mlp(x))  (k_sd_ln<1>, transformer.py:66) for the rows decoded here -- nobody else reads
                                                   // that x.  Ymlp: the c_proj output as `yparts` split-K partial products [yparts][M][Wd], added in order; null: X is final
                                                   const float* __restrict__ Ymlp, int yparts, int M, const float* __restrict__ g2, const float* __restrict__ b2) {
    constexpr int NPW = 4, HH = H / 2, NR = NV ? NV : SD_MAXV;
    __shared__ float ys[NPW][H];
    __shared__ float hs[NPW][HH];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, tid = threadIdx.x;
    const int n = blockIdx.x * NPW + w;
    const bool live = n < N;
    {
        float v[NR];
        const int row = node_tok[live ? n : 0];
#pragma unroll
        for (int i = 0; i < NR; ++i) v[i] = (NV || lane + 64 * i < Wd) ? X[(size_t)row * Wd + ((NV || lane + 64 * i < Wd) ? lane + 64 * i : 0)] : 0.0f;
        if (Ymlp) {
            float y[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) y[i] = (NV || lane + 64 * i < Wd) ? Ymlp[(size_t)row * Wd + ((NV || lane + 64 * i < Wd) ? lane + 64 * i : 0)] : 0.0f;
            for (int k = 1; k < yparts; ++k) {
#pragma unroll
                for (int i = 0; i < NR; ++i)
                    y[i] += (NV || lane + 64 * i < Wd) ? Ymlp[((size_t)k * M + row) * Wd + ((NV || lane + 64 * i < Wd) ? lane + 64 * i : 0)] : 0.0f;
            }
            ln_row<NV>(y, Wd, lane, g2, b2);
#pragma unroll
            for (int i = 0; i < NR; ++i) v[i] = v[i] + y[i];
        }
        ln_row<NV>(v, Wd, lane, gam, bet);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int c = lane + 64 * i;
            if ((NV || c < Wd) && c >= Wd - H) ys[w][c - (Wd - H)] = live ? v[i] : 0.0f;
        }
    }
    __syncthreads();
    for (int u = tid; u < HH; u += 256) {                          // one hidden unit, all four nodes
        const float b0 = pd0_b[u];
        float q[NPW] = {b0, b0, b0, b0};
        const float* wp = pd0_wT + u;
#pragma unroll 8
        for (int c = 0; c < H; ++c) {
            const float wv = wp[(size_t)c * HH];
#pragma unroll
            for (int j = 0; j < NPW; ++j) q[j] += ys[j][c] * wv;
        }
#pragma unroll
        for (int j = 0; j < NPW; ++j) hs[j][u] = silu_f(q[j]);
    }
    __syncthreads();
    if (!live) return;
    for (int p = 0; p < P; ++p) {
        float part = 0.0f;
        for (int k = lane; k < HH; k += 64) part += hs[w][k] * pd2_w[(size_t)p * HH + k];
        const float o = wave_sum(part) + pd2_b[p];
        if (lane == 0) eps[(size_t)n * P + p] = mask[n] ? xfeat[(size_t)n * F + F - P + p] : o;
    }
}
