// ccsp_edge_fb.h -- energy mode, round 6: the decoder's forward AND backward in one kernel (k_edge_fb_h2), and the backward alone on the same
// tile geometry (k_edge_bwd2_h2).  Included inside the anonymous namespace of ccsp_hip.hip, behind ccsp_f16x2.h.
//
// Everything the decoder backward needs is LOCAL to an (edge, half) row of the forward tile (reference: the autograd of denoise_fn.py:341-375,
// SURVEY Appendix A.4): q (the decoder's hidden pre-activations), go = 2 d = 2 (o - pose), and z = U[u0] + U[u1].  Rounds 1-5 ran it as a kernel of
// its own, k_edge_bwd_h2: workgroup = (64 edges, half, 128 of the 256 g_z columns), reading q back from a [2E, 128] array the forward kernel wrote
// (20 MB out, 20 MB in at C4), the edge outputs from the CSR, and forming the A operand  A[row, j] = (sum_p go[p] Wd2[p, j]) SiLU'(q[row, j])  TWICE --
// once per column tile -- on the VALU, in the loop that bounds it.  Here the tile is the forward kernel's: 32 sorted edges x both halves = 64 rows,
// all 256 columns of g_z per row:
//   * A is formed ONCE, straight from the accumulator registers that hold q at the end of the forward GEMM (go from LDS, the two decoder-weight
//     columns of the lane in registers), scaled, split and written as fp16 planes of all four K chunks (32 KB of LDS);
//   * the two 128-column passes of  g_h = A . Wd1  run back to back on those planes (weights through one 16 KB stage), both passes' accumulators
//     stay in registers (64 VGPRs), then the two epilogues -- x SiLU'(z), the ordered row sums of ccsp::build_bwdsum_plan on 32-edge blocks, the
//     transpose GEMM's operand planes -- go through one 34 KB tile that takes the planes' place;
//   * no Q array, no second launch, the U rows of the tile are gathered a second time from the L2 they were just read through instead of the fabric.
// 50.3 KB of LDS and <= 168 VGPRs: three workgroups per CU, C4's 624 tiles in one round (the separate backward was 1248 workgroups = 1.6 rounds).
// Same products, same scaling rule and the same order of every sum as k_edge_h2<true, 1, 0> followed by k_edge_bwd_h2<true, PP> except for the
// partial rows, which now cover 32 edges instead of 64 (a U row's gradient is added up from more, shorter partial sums: fp32 rounding only --
// test_energy_mode_* / test_mala_h256_every_timestep_vs_reference hold with unchanged bars).
#pragma once

// Partial-row block of 32 sorted edges (ccsp::build_bwdsum_plan(p, tile_m, plan, 32, 64)): [0] = partial rows, [1 + p] = global partial row,
// [65 + p] = first << 16 | end of p's PAIRS, [129 + 2 q], [130 + 2 q] = the two entries of pair q -- byte offsets of a block-local edge's half-0 row
// in the epilogue tile (edge * C_LD * 4; the half-1 row is 32 rows further on), an odd count padded with the all-zero row 64.
constexpr int FB_EDGES = 32, FB_MAXP = 64, FB_BLK = 1 + 2 * FB_MAXP + 2 * FB_MAXP;

struct EdgeFbArgs {
    const unsigned short* Wd1TH;  // [256][128 / 32][2][32] chunk-interleaved planes of pose_decoder.0.weight^T (k_edge_bwd_h2's B operand)
    float wd2_absmax;             // max |pose_decoder.2.weight|
    BwdSumArgs bs;                // blocks of FB_BLK ints per 32 edges
};

// LDS map of the backward phase (bytes).  The forward phase (k_edge_h2<true, 1, 0>'s two 24 KB stages, its epilogue tiles on top of them) lies
// inside [0, FB_GO); go / the row exponents are written while the forward epilogue's tiles are still live and read until the kernel's end.
constexpr int FB_APL = 64 * H2_BK;                           // elements per A plane of one K chunk
constexpr int FB_A2 = 0;                                     // [4 chunks][2 planes][64 rows][32] fp16: 32 KB
constexpr int FB_BST = 4 * 2 * FB_APL * 2;                   // one weight stage [2 planes][128 rows][32] fp16: 16 KB
constexpr int FB_GO = FB_BST + 2 * H2_BPL * 2;               // go[64][8] fp32: 2 KB
constexpr int FB_EXP = FB_GO + 64 * 8 * 4;                   // a_exp2[64] int
constexpr int FB_SMEM = FB_EXP + 64 * 4;                     // 51 456 bytes
constexpr int FB_CLD = 132;                                  // row stride of the epilogue tile (ccsp::BS_CLD)
constexpr int FB_CS = 0;                                     // epilogue tile [65][132] fp32 = 34 320 bytes (rows 0..31 half 0, 32..63 half 1, 64 zero)
constexpr int FB_BSB = 36 * 1024;                            // the block's plan (FB_BLK ints) + exponents of its partial rows [64]
static_assert(65 * FB_CLD * 4 <= FB_BSB && FB_BSB + (FB_BLK + FB_MAXP) * 4 <= FB_GO, "epilogue overlays");
static_assert(FB_CLD == ccsp::BS_CLD, "tile stride of the partial-row plan");
static_assert(3 * FB_SMEM <= 160 * 1024, "three workgroups per CU");

// The backward phase.  On entry: qv[j][r] = q of tile row  wm * 32 + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), decoder unit  wn * 64 + 32 j + (lane & 31)
// (the forward GEMM's accumulator layout); go[row][0 .. PMAX) and W2s (pose_decoder.2.weight [8][128], rows >= P zero) in LDS, visible to every
// thread; nobody reads [0, FB_GO) any more EXCEPT W2s, which this function reads before its first barrier.
//   u0 / u1: U rows of the edge of tile rows tid >> 3 and 32 + (tid >> 3) (one edge: both halves); k_lr: that edge's index (>= E_act: a padding row)
template <int PP>
__device__ __forceinline__ void edge_bwd_phase(unsigned char* __restrict__ smem, float (&qv)[2][16], const float* __restrict__ W2s, int P_rt, int blk_id,
                                               int u0, int u1, const float* __restrict__ U, int wd_exp, const EdgeFbArgs& fa) {
    constexpr int H = 256, KD = 128, NCH = KD / H2_BK;
    constexpr int PMAX = PP ? PP : 8;
    const int P = PP ? PP : P_rt;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = tid >> 3, lq = tid & 7;
    unsigned short* A2 = reinterpret_cast<unsigned short*>(smem + FB_A2);
    unsigned short* Bst = reinterpret_cast<unsigned short*>(smem + FB_BST);
    const float* GOs = reinterpret_cast<const float*>(smem + FB_GO);
    int* sE2 = reinterpret_cast<int*>(smem + FB_EXP);
    // weight staging of k_edge_bwd_h2: 128 n-rows x one K chunk, both planes; iteration it = pass * 4 + chunk
    const int brow = tid >> 2, bq = tid & 3;
    const unsigned short* b_ptr = fa.Wd1TH + (size_t)brow * (2 * KD) + bq * 8;
    const int b_st = h2_off(brow, bq);
    ushort8 rb[4];
    auto gload_b = [&](int it) {
        const int n0 = (it >> 2) * 128, c = it & 3;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                rb[i * 2 + p] = *reinterpret_cast<const ushort8*>(b_ptr + (size_t)(n0 + i * 64) * (2 * KD) + (size_t)p * H2_BK + c * (2 * H2_BK));
    };
    auto store_b = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                *reinterpret_cast<ushort8*>(Bst + p * H2_BPL + b_st + i * 64 * H2_BK) = rb[i * 2 + p];
    };
    gload_b(0);
    // ---- the A operand, once: 32 elements per thread from the registers that hold q ----
    float w2[2][PMAX];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < PMAX; ++p) w2[j][p] = W2s[p * KD + wn * 64 + j * 32 + (lane & 31)];
    // (pairs of lanes hold neighbouring columns of a row: the even lane writes both columns' first fp16 terms as one dword, the odd lane both second
    // terms -- 32 ds_write_b32 per thread instead of 64 ds_write_b16)
    const int kk = lane & 31, odd = lane & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float gv[PMAX];
        float sum = 0.0f;
        if constexpr (PMAX == 4) {
            const float4 g4 = *reinterpret_cast<const float4*>(GOs + row * 8);
            gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
        } else {
#pragma unroll
            for (int p4 = 0; p4 < PMAX; p4 += 4) {
                const float4 g4 = *reinterpret_cast<const float4*>(GOs + row * 8 + p4);
                gv[p4] = g4.x; gv[p4 + 1] = g4.y; gv[p4 + 2] = g4.z; gv[p4 + 3] = g4.w;
            }
        }
#pragma unroll
        for (int p = 0; p < PMAX; ++p) sum += (PP || p < P) ? fabsf(gv[p]) : 0.0f;
        const int ae = h2_scale_exp(1.1f * fa.wd2_absmax * sum);            // |A| <= 1.1 max|Wd2| sum_p |go|: k_edge_bwd_h2's row exponent
        if (wn == 0 && kk == 0) sE2[row] = ae;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float g = 0.0f;
#pragma unroll
            for (int p = 0; p < PMAX; ++p) g = (PP || p < P) ? fmaf(gv[p], w2[j][p], g) : g;       // (same order as k_edge_bwd_h2's store_a)
            const float h = g * silu_grad_fast(qv[j][r]);
            unsigned short p1, p2;
            split2h(ldexpf(h, ae), p1, p2);
            const unsigned int mine = (unsigned int)p1 | ((unsigned int)p2 << 16);
            const unsigned int other = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1 /*quad_perm [1, 0, 3, 2]*/, 0xf, 0xf, true);
            const unsigned int word = odd ? ((other >> 16) | (mine & 0xffff0000u)) : ((mine & 0xffffu) | (other << 16));
            const int c = wn * 2 + j;                                       // K chunk of the lane's column
            const int ke = kk & ~1;                                         // first column of the pair
            unsigned short* d = A2 + c * (2 * FB_APL) + odd * FB_APL + h2_off(row, ke >> 3) + (ke & 7);
            *reinterpret_cast<unsigned int*>(d) = word;
        }
    }
    floatx16 acc2[2][1][2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[n][0][j][r] = 0.0f;
    __syncthreads();                                              // the planes are written, W2s has been read: the stage may take its place
    CCSP_TRK(1, 4);
#pragma unroll
    for (int it = 0; it < 2 * NCH; ++it) {
        store_b();
        if (it + 1 < 2 * NCH) gload_b(it + 1);
        __syncthreads();
        const unsigned short* As = A2 + (it & 3) * (2 * FB_APL);
        h2_kstep<1>(As, FB_APL, Bst, 0, wm * 32, wn * 64, acc2[it >> 2]);
        h2_kstep<1>(As, FB_APL, Bst, 1, wm * 32, wn * 64, acc2[it >> 2]);
        __syncthreads();                                          // (the stage is rewritten next; after the last chunk: the planes are dead)
        if (it == NCH - 1) CCSP_TRK(1, 5);
    }
    CCSP_TRK(1, 6);
    // ---- epilogues: g_z = 2^-(e_row + wd_exp) acc x SiLU'(U[u0] + U[u1]), ordered partial-row sums, operand planes of the transpose GEMM ----
    float* Cs = reinterpret_cast<float*>(smem + FB_CS);
    int* bsb = reinterpret_cast<int*>(smem + FB_BSB);
    int* pexp = bsb + FB_BLK;
    {
        const int* blk = fa.bs.blocks + (size_t)blk_id * FB_BLK;
        const int i0 = tid, i1 = tid + 256;
        const int v0 = blk[i0], v1 = blk[i1 < FB_BLK ? i1 : FB_BLK - 1];
        bsb[i0] = v0;
        if (i1 < FB_BLK) bsb[i1] = v1;
    }
    const int ae_r[2] = {sE2[lr], sE2[lr + 32]};
    float bnd = 0.0f;                                             // the edge's bound on |g_z[k, :]| (both halves): rides in row lr's padding column
    {
        float s2 = 0.0f;
#pragma unroll
        for (int p = 0; p < PMAX; ++p) s2 += (PP || p < P) ? (fabsf(GOs[lr * 8 + p]) + fabsf(GOs[(lr + 32) * 8 + p])) : 0.0f;
        bnd = fa.bs.bound_c * s2;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int n0 = n * 128;
        float4 ua[2][4], ub[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {                             // tile row lr + 32 i: half i of the edge
            const float* p0 = U + (size_t)u0 * (2 * H) + i * H + n0 + lq * 4;
            const float* p1 = U + (size_t)u1 * (2 * H) + i * H + n0 + lq * 4;
#pragma unroll
            for (int mcol = 0; mcol < 4; ++mcol) {
                ua[i][mcol] = *reinterpret_cast<const float4*>(p0 + 32 * mcol);
                ub[i][mcol] = *reinterpret_cast<const float4*>(p1 + 32 * mcol);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                Cs[row * FB_CLD + wn * 64 + j * 32 + (lane & 31)] = acc2[n][0][j][r];
            }
        if (tid < FB_CLD) Cs[64 * FB_CLD + tid] = 0.0f;            // the all-zero row that pads odd entry counts
        __syncthreads();
        CCSP_TRK(1, 10 + 4 * n);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = lr + 32 * i;
            const int e = -(ae_r[i] + wd_exp);
#pragma unroll
            for (int mcol = 0; mcol < 4; ++mcol) {
                const float4 a = ua[i][mcol], b = ub[i][mcol];
                float4* cp = reinterpret_cast<float4*>(Cs + row * FB_CLD + lq * 4 + 32 * mcol);
                const float4 v = *cp;
                *cp = make_float4(ldexpf(v.x, e) * silu_grad_fast(a.x + b.x), ldexpf(v.y, e) * silu_grad_fast(a.y + b.y),
                                  ldexpf(v.z, e) * silu_grad_fast(a.z + b.z), ldexpf(v.w, e) * silu_grad_fast(a.w + b.w));
            }
        }
        if (lq == 0) Cs[lr * FB_CLD + 128] = bnd;
        __syncthreads();
        CCSP_TRK(1, 11 + 4 * n);
        const int np = bsb[0];
        if (n == 0) {
            // exponents: one thread per partial row adds the bounds of its edges (ascending; the same for both passes)
            if (tid < np) {
                const int span = bsb[1 + FB_MAXP + tid];
                float bsum = 0.0f;
                for (int q = span >> 16; q < (span & 0xffff); ++q) {
                    const int2 rr = *reinterpret_cast<const int2*>(bsb + 1 + 2 * FB_MAXP + 2 * q);
                    bsum += Cs[(rr.x >> 2) + 128];
                    bsum += Cs[(rr.y >> 2) + 128];
                }
                const int e = h2_scale_exp(bsum);
                pexp[tid] = e;
                fa.bs.gexp[bsb[1 + tid]] = e;
            }
            __syncthreads();
        }
        CCSP_TRK(1, 12 + 4 * n);
        // (round 6, measured and dropped: two partial rows per trip with interleaved loads + four threads per exponent -- C4 recomputing 164.6 -> 161.4
        //  same call: the phase is 12 k of the kernel's 70 k cycles in the trace, but the longer loop body costs more than the second chain hides)
        // sums: thread = (columns 4 cg .. + 3 and 64 + 4 cg .. + 3 of the pass, half hs, one of 8 row lanes); entries two at a time, ascending
        const int cg = tid & 15, hs = (tid >> 4) & 1, rl = tid >> 5;
        const char* Cb = reinterpret_cast<const char*>(Cs) + cg * 16;
        const int hoff = hs * 32 * FB_CLD * 4, zrow = 64 * FB_CLD * 4;
        for (int p = rl; p < np; p += 8) {
            const int span = bsb[1 + FB_MAXP + p];
            const int e = pexp[p];
            const int gid = bsb[1 + p];
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
            for (int q = span >> 16; q < (span & 0xffff); ++q) {
                const int2 rr = *reinterpret_cast<const int2*>(bsb + 1 + 2 * FB_MAXP + 2 * q);
                const int ox = rr.x + hoff, oy = rr.y == zrow ? zrow : rr.y + hoff;      // (only a pair's second entry can be the padding row)
                const float4 v0 = *reinterpret_cast<const float4*>(Cb + ox), v1 = *reinterpret_cast<const float4*>(Cb + ox + 256);
                const float4 w0 = *reinterpret_cast<const float4*>(Cb + oy), w1 = *reinterpret_cast<const float4*>(Cb + oy + 256);
                a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
                a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
                a0.x += w0.x; a0.y += w0.y; a0.z += w0.z; a0.w += w0.w;
                a1.x += w1.x; a1.y += w1.y; a1.z += w1.z; a1.w += w1.w;
            }
            const float hh[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            unsigned short p1[8], p2[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) split2h(ldexpf(hh[k], e), p1[k], p2[k]);
            const int col = hs * H + n0 + cg * 4;
#if CCSP_A_INTERLEAVED                                              // [NP][512 / 32][2][32]: both planes of a row's K chunk in one line (k_rowgemm_h2, ILA)
            const size_t o = (size_t)gid * (4 * H) + (size_t)(col >> 5) * 64 + (col & 31);
            *reinterpret_cast<uint2*>(fa.bs.GZPH + o) = make_uint2(p1[0] | ((unsigned)p1[1] << 16), p1[2] | ((unsigned)p1[3] << 16));
            *reinterpret_cast<uint2*>(fa.bs.GZPH + o + 128) = make_uint2(p1[4] | ((unsigned)p1[5] << 16), p1[6] | ((unsigned)p1[7] << 16));      // (column + 64 = two chunks on)
            *reinterpret_cast<uint2*>(fa.bs.GZPH + o + 32) = make_uint2(p2[0] | ((unsigned)p2[1] << 16), p2[2] | ((unsigned)p2[3] << 16));
            *reinterpret_cast<uint2*>(fa.bs.GZPH + o + 160) = make_uint2(p2[4] | ((unsigned)p2[5] << 16), p2[6] | ((unsigned)p2[7] << 16));
#else
            const size_t o = (size_t)gid * (2 * H) + col;
            *reinterpret_cast<uint2*>(fa.bs.GZPH + o) = make_uint2(p1[0] | ((unsigned)p1[1] << 16), p1[2] | ((unsigned)p1[3] << 16));
            *reinterpret_cast<uint2*>(fa.bs.GZPH + o + 64) = make_uint2(p1[4] | ((unsigned)p1[5] << 16), p1[6] | ((unsigned)p1[7] << 16));
            *reinterpret_cast<uint2*>(fa.bs.GZPH + fa.bs.plane + o) = make_uint2(p2[0] | ((unsigned)p2[1] << 16), p2[2] | ((unsigned)p2[3] << 16));
            *reinterpret_cast<uint2*>(fa.bs.GZPH + fa.bs.plane + o + 64) = make_uint2(p2[4] | ((unsigned)p2[5] << 16), p2[6] | ((unsigned)p2[7] << 16));
#endif
        }
        if (n == 0) { __syncthreads(); CCSP_TRK(1, 7); }          // (the second pass rewrites the tile)
    }
}

// The backward alone on the fused kernel's tile geometry: q from the [2E, 128] array and go from the CSR slots the forward kernel wrote
// (k_edge_h2<true, ...> with en.Q set).  The stepping stone the fused kernel was validated on, kept as CCSP_EDGE_FB=1 for A/B runs.
template <int PP>
__global__ __launch_bounds__(256, 3) void k_edge_bwd2_h2(int E_act, int P_rt, const int* __restrict__ e_u0, const int* __restrict__ e_u1,
                                                         const int* __restrict__ ent_pos, const float* __restrict__ U, const float* __restrict__ Ocsr,
                                                         const float* __restrict__ Q /*[2E,128]*/, int wd_exp, const float* __restrict__ Wd2 /*[P,128]*/,
                                                         const int* __restrict__ skip, EdgeFbArgs fa) {
    if (skip && *skip == 0) return;
    constexpr int KD = 128;
    constexpr int PMAX = PP ? PP : 8;
    const int P = PP ? PP : P_rt;
    __shared__ __attribute__((aligned(16))) unsigned char smem[FB_SMEM];
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int e0 = wg * FB_EDGES;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = tid >> 3;
    int k_lr = e0 + lr;
    k_lr = k_lr < E_act ? k_lr : E_act - 1;
    const int u0 = e_u0[k_lr], u1 = e_u1[k_lr];
    // q in the forward accumulator layout
    float qv[2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int le = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int k = e0 + le;
        k = k < E_act ? k : E_act - 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) qv[j][r] = Q[((size_t)2 * k + wm) * KD + wn * 64 + j * 32 + (lane & 31)];
    }
    float* W2s = reinterpret_cast<float*>(smem + FB_BST);        // (where the forward epilogue leaves it: inside the stage, read before the stage is written)
    float* GOs = reinterpret_cast<float*>(smem + FB_GO);
    for (int idx = tid; idx < 8 * KD; idx += 256) W2s[idx] = idx < P * KD ? Wd2[idx < P * KD ? idx : 0] : 0.0f;
    for (int idx = tid; idx < 64 * 8; idx += 256) {
        const int row = idx >> 3, p = idx & 7;
        const int k = e0 + (row & 31);
        const int kc = k < E_act ? k : E_act - 1;
        const int slot = ent_pos[2 * kc + (row >> 5)];
        const float ov = Ocsr[(size_t)slot * P + (p < P ? p : P - 1)];
        GOs[idx] = (p < P && p < PMAX && k < E_act) ? -ov : 0.0f;                         // 2 d = -(-2 d); padding rows contribute nothing
    }
    __syncthreads();
    edge_bwd_phase<PP>(smem, qv, W2s, P_rt, wg, u0, u1, U, wd_exp, fa);
}

// Forward + backward: k_edge_h2<true, 1, 0>'s body up to its epilogue, then edge_bwd_phase on the registers that hold q.
template <int PP>
__global__ __launch_bounds__(256, 3) void k_edge_fb_h2(int E_act, int P_rt, const int* __restrict__ e_u0, const int* __restrict__ e_u1,
                                                       const float* __restrict__ U, const float* __restrict__ umax /*[R][8]*/,
                                                       const unsigned short* __restrict__ Wd1H /*[128][256 / 32][2][32]*/, int wd_exp,
                                                       const float* __restrict__ bd1, const float* __restrict__ Wd2, const float* __restrict__ bd2,
                                                       const int* __restrict__ ent_pos, float* __restrict__ O, EdgeEnergyArgs en, EdgeFbArgs fa) {
    if (en.skip && *en.skip == 0) return;                          // (uniform) MALA reuse
    CCSP_TRK(1, 0);
    CCSP_TRK_RT(1, 30);
    constexpr int H = 256, BN = 128, NCH = H / H2_BK;
    constexpr int ME = FB_EDGES, ROWS = 2 * ME, NPASS = ROWS / 32;
    constexpr int APL = ROWS * H2_BK, STAGE = 2 * APL + 2 * H2_BPL;
    constexpr int S1_LD = BN + 4;
    constexpr int PMAX = PP ? PP : 8;
    const int P = PP ? PP : P_rt;
    static_assert(2 * STAGE * 2 <= FB_GO && (64 * S1_LD + 8 * BN) * 4 <= FB_GO, "forward stages and epilogue tiles lie below go / the exponents");
    static_assert(64 * S1_LD * 4 >= FB_BST, "W2s behind the S1 tile is inside the weight stage (edge_bwd_phase reads it before the stage is written)");
    __shared__ __attribute__((aligned(16))) unsigned char smem_b[FB_SMEM];
    unsigned short* smem = reinterpret_cast<unsigned short*>(smem_b);
    int* sE = reinterpret_cast<int*>(smem_b + FB_EXP);            // forward row exponents (the backward phase rewrites the array with its own)
    float* GOs = reinterpret_cast<float*>(smem_b + FB_GO);
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int e0 = wg * ME;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = tid >> 3, lq = tid & 7;
    int r0v[NPASS], r1v[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        int k = e0 + ((lr + 32 * i) % ME);
        k = k < E_act ? k : E_act - 1;
        r0v[i] = e_u0[k];
        r1v[i] = e_u1[k];
    }
    const float* u0_ptr[NPASS];
    const float* u1_ptr[NPASS];
    int a_st[NPASS], a_exp[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int row = lr + 32 * i, s = row / ME;
        u0_ptr[i] = U + (size_t)r0v[i] * (2 * H) + s * H + lq * 4;
        u1_ptr[i] = U + (size_t)r1v[i] * (2 * H) + s * H + lq * 4;
        a_st[i] = h2_off(row, lq >> 1) + (lq & 1) * 4;
    }
    const int brow = tid >> 2, bq = tid & 3;
    const unsigned short* b_ptr = Wd1H + (size_t)brow * (2 * H) + bq * 8;
    const int b_st = h2_off(brow, bq);
    float4 ua[2][NPASS], ub[2][NPASS];
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
    auto store_a = [&](int stage, int set, int i) {
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
        const float4 m0 = *reinterpret_cast<const float4*>(umax + (size_t)r0v[i] * 8 + 4 * s);
        const float4 m1 = *reinterpret_cast<const float4*>(umax + (size_t)r1v[i] * 8 + 4 * s);
        a_exp[i] = h2_scale_exp(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)) + fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w)));
    }
    float bj1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bj1[j] = bd1[wn * 64 + j * 32 + (lane & 31)];
    const int o_p = tid >> 6, o_row = tid & 63;
    int o_slot;
    {
        int k = e0 + (o_row & 31);
        k = k < E_act ? k : E_act - 1;
        o_slot = ent_pos[2 * k + (o_row >> 5)];
    }
    const float o_b2 = bd2[o_p < P ? o_p : 0];
    const float4 w2v = *reinterpret_cast<const float4*>(Wd2 + ((tid * 4) < P * BN ? tid * 4 : 0));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NPASS; ++i)
        if (lq == 0) sE[lr + 32 * i] = a_exp[i];
    floatx16 acc[1][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.0f;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) store_a(0, 0, i);
    store_b(0);
    gload_b(1);
    gload_a(2, 0);
    __syncthreads();
    CCSP_TRK(1, 1);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const unsigned short* st = smem + (c & 1) * STAGE;
        const int nx = (c + 1) & 1;
        h2_kstep<1>(st, APL, st + 2 * APL, 0, wm * ME, wn * 64, acc);
        if (c + 1 < NCH) store_a(nx, nx, 0);
        h2_kstep<1>(st, APL, st + 2 * APL, 1, wm * ME, wn * 64, acc);
        if (c + 1 < NCH) { store_a(nx, nx, 1); store_b(nx); }
        if (c + 2 < NCH) gload_b(c + 2);
        if (c + 3 < NCH) gload_a(c + 3, nx);
        __syncthreads();
    }
    CCSP_TRK(1, 2);
    // forward epilogue (k_edge_h2<true, 1, 0>): q stays in registers, SiLU(q) -> S1, layer 2, d = o - pose, energy partial, -2 d to the CSR slot
    float* S1 = reinterpret_cast<float*>(smem);
    float* W2s = S1 + 64 * S1_LD;
    for (int idx = tid; idx < 8 * BN; idx += 256) W2s[idx] = 0.0f;
    float qv[2][16];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float q = ldexpf(acc[0][j][r], -(sE[wm * ME + rr] + wd_exp)) + bj1[j];
            qv[j][r] = q;
            S1[(wm * 32 + rr) * S1_LD + col] = silu_fast(q);
        }
    }
    __syncthreads();                                               // (the zeros of W2s are in place before the rows < P are written over them)
    if (tid * 4 < P * BN) *reinterpret_cast<float4*>(W2s + tid * 4) = w2v;
    for (int idx = tid; idx < 64 * 8; idx += 256) GOs[idx] = 0.0f;   // (components >= P and padding rows stay 0)
    __syncthreads();
    float e2 = 0.0f;
    for (int idx = tid; idx < 64 * P; idx += 256) {
        const int lrow = idx & 63;
        const int p = idx >> 6;
        const float4* sr = reinterpret_cast<const float4*>(S1 + lrow * S1_LD);
        const float4* wr = reinterpret_cast<const float4*>(W2s + p * BN);
        float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f, o3 = 0.0f;
#pragma unroll 8
        for (int j = 0; j < BN / 4; ++j) {
            const float4 sv = sr[j], wv = wr[j];
            o0 = fmaf(sv.x, wv.x, o0); o1 = fmaf(sv.y, wv.y, o1); o2 = fmaf(sv.z, wv.z, o2); o3 = fmaf(sv.w, wv.w, o3);
        }
        const float o = ((o0 + o1) + (o2 + o3)) + (idx == tid ? o_b2 : bd2[p]);
        const int k = e0 + (lrow & 31), s = lrow >> 5;
        if (k < E_act) {
            const int slot = idx == tid ? o_slot : ent_pos[2 * k + s];
            const int node = s == 0 ? en.e_a[k] : en.e_b[k];
            const float d = o - en.xeval[(size_t)node * P + p];
            e2 = fmaf(d, d, e2);
            O[(size_t)slot * P + p] = -2.0f * d;
            if (p < PMAX) GOs[lrow * 8 + p] = 2.0f * d;
        }
    }
    // the energy partial (two barriers inside: they also make go visible and end every read of S1)
    {
        const float tot = block_sum_256(e2, reinterpret_cast<float*>(smem_b + FB_EXP));       // (the forward exponents are dead: q is unscaled)
        if (tid == 0) en.partial[blockIdx.x] = tot;
    }
    __syncthreads();
    CCSP_TRK(1, 3);
    edge_bwd_phase<PP>(smem_b, qv, W2s, P_rt, wg, r0v[0], r1v[0], U, wd_exp, fa);
    CCSP_TRK(1, 9);
    CCSP_TRK_RT(1, 31);
}
