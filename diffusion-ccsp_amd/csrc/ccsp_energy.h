// Energy mode of the denoiser (reference networks/denoise_fn.py:373-375,518-519,527-529,539-548 and
// ComposedEBMDenoiseFn :57-83): E = sum over (edge, slot) |o - pose|^2 for the whole batch, and the
// "epsilon" the sampler consumes is dE/dposes.  The reference gets the gradient from autograd; here it
// is the hand-derived backward of SURVEY.md Appendix A.4, evaluated with the same row factorisation
// as the forward pass:
//
//   k_edge<H, true>   forward decoder; writes -2 d = -2 (o - pose) to the node CSR slots (the direct
//                     term), the decoder pre-activations q, and a per-workgroup partial of sum d^2
//   k_edge_bwd        g_h = (Wd2^T 2d  (.) SiLU'(q)) Wd1  per (edge, slot) on the MFMA; epilogue
//                     g_z = g_h (.) SiLU'(z), z recomputed from the two U rows of the edge
//   k_rowsum          g_z summed over the edges that share a U row (ordered, no atomics)
//   k_rowgemm<2H, H>  g_p[row] = g_z[row] . Wp[type, slot]   (the transpose GEMM of k_ugemm)
//   k_node_energy     per node: direct term (CSR) + pose-encoder backward of sum_rows g_p -> dE/dpose
//
// This file is included inside the anonymous namespace of ccsp_hip.hip.
#pragma once

// sum of n partials in a fixed order -> out[0]   (one workgroup of 256)
__global__ __launch_bounds__(256) void k_energy_sum(const float* __restrict__ partial, int n, float* __restrict__ out) {
    __shared__ float red[256];
    float v = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) v += partial[i];
    const float s = block_sum_256(v, red);
    if (threadIdx.x == 0) out[0] = s;
}

// ------------------------------------------------------------------------------------------
// k_edge_bwd: rows = (sorted edge k, slot s); K = H/2 (decoder hidden), N = H (one half of the
// type-MLP output).  A[row, j] = (sum_p go[p] Wd2[p, j]) * SiLU'(q[row, j]) with go = 2 d = -O_csr.
// B rows = Wd1^T [H, H/2].  Epilogue: GZ[k, s*H + n] = acc * SiLU'(U[u0(k)] + U[u1(k)])[s*H + n].
// ------------------------------------------------------------------------------------------
template <int H> struct BwdCfg { static constexpr int WM = 2, WN = 2, TN = 1; };      // (any other multiple of 64: 64 x 64 tiles, H / 64 column tiles)
template <> struct BwdCfg<256> { static constexpr int WM = 2, WN = 2, TN = 2; };   // 64 x 128 tile, 2 column tiles
template <> struct BwdCfg<128> { static constexpr int WM = 2, WN = 2, TN = 1; };   // 64 x 64 tile, 2 column tiles
template <> struct BwdCfg<64> { static constexpr int WM = 2, WN = 2, TN = 1; };    // 64 x 64 tile, 1 column tile

template <int H>
__global__ __launch_bounds__(256) void k_edge_bwd(int E_act, int P, const int* __restrict__ e_u0,
                                                  const int* __restrict__ e_u1, const int* __restrict__ ent_pos,
                                                  const float* __restrict__ U, const float* __restrict__ Ocsr,
                                                  const float* __restrict__ Q /*[2E,H/2]*/,
                                                  const float* __restrict__ Wd1T /*[H,H/2]*/,
                                                  const float* __restrict__ Wd2 /*[P,H/2]*/, float* __restrict__ GZ) {
    using Cfg = BwdCfg<H>;
    constexpr int BM = 32 * Cfg::WM, BN = 32 * Cfg::TN * Cfg::WN, TN = Cfg::TN, KD = H / 2;
    constexpr int NCT = H / BN;
    constexpr int A_ROWS_PT = BM / 32, B_ROWS_PT = BN / 32;
    constexpr int STAGE = (BM + BN) * LDS_LD;
    __shared__ float smem[2 * STAGE];
    auto As = [&](int buf) -> float* { return smem + buf * STAGE; };
    auto Bs = [&](int buf) -> float* { return smem + buf * STAGE + BM * LDS_LD; };
    const int bid = blockIdx.x;
    const int ct = bid % NCT, s = (bid / NCT) & 1, e0 = (bid / (2 * NCT)) * BM;
    const int n0 = ct * BN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int lr = tid >> 3, lq = tid & 7;
    float go[A_ROWS_PT][8];
    const float* q_ptr[A_ROWS_PT];
    const float* b_ptr[B_ROWS_PT];
#pragma unroll
    for (int i = 0; i < A_ROWS_PT; ++i) {
        int k = e0 + lr + 32 * i;
        k = k < E_act ? k : E_act - 1;
        const size_t row = (size_t)2 * k + s;
        const float* o = Ocsr + (size_t)ent_pos[row] * P;
#pragma unroll
        for (int p = 0; p < 8; ++p) go[i][p] = p < P ? -o[p] : 0.0f;       // 2 d = -(-2 d)
        q_ptr[i] = Q + row * KD + lq * 4;
    }
#pragma unroll
    for (int i = 0; i < B_ROWS_PT; ++i) b_ptr[i] = Wd1T + (size_t)(n0 + lr + 32 * i) * KD + lq * 4;
    float4 ra[A_ROWS_PT], rb[B_ROWS_PT];
    auto load_chunk = [&](int c) {
        float4 w2[8];
#pragma unroll
        for (int p = 0; p < 8; ++p)
            w2[p] = p < P ? *reinterpret_cast<const float4*>(Wd2 + (size_t)p * KD + c * BK + lq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < A_ROWS_PT; ++i) {
            const float4 q = *reinterpret_cast<const float4*>(q_ptr[i] + c * BK);
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                g.x = fmaf(go[i][p], w2[p].x, g.x); g.y = fmaf(go[i][p], w2[p].y, g.y);
                g.z = fmaf(go[i][p], w2[p].z, g.z); g.w = fmaf(go[i][p], w2[p].w, g.w);
            }
            ra[i].x = g.x * silu_grad_fast(q.x); ra[i].y = g.y * silu_grad_fast(q.y);
            ra[i].z = g.z * silu_grad_fast(q.z); ra[i].w = g.w * silu_grad_fast(q.w);
        }
#pragma unroll
        for (int i = 0; i < B_ROWS_PT; ++i) rb[i] = *reinterpret_cast<const float4*>(b_ptr[i] + c * BK);
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_ROWS_PT; ++i) lds_store4(As(buf) + (lr + 32 * i) * LDS_LD + lq * 4, ra[i]);
#pragma unroll
        for (int i = 0; i < B_ROWS_PT; ++i) lds_store4(Bs(buf) + (lr + 32 * i) * LDS_LD + lq * 4, rb[i]);
    };
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    floatx16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    constexpr int NCH = KD / BK;
    for (int c = 0; c < NCH; ++c) {
        const int buf = c & 1;
        if (c + 1 < NCH) load_chunk(c + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk<TN>(As(buf), Bs(buf), wm * 32, wn * 32 * TN, acc);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < NCH) store_chunk(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int k = e0 + row;
        if (k >= E_act) continue;
        const float* u0 = U + (size_t)e_u0[k] * (2 * H) + s * H;
        const float* u1 = U + (size_t)e_u1[k] * (2 * H) + s * H;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * 32 * TN + j * 32 + (lane & 31);
            const float z = u0[n] + u1[n];
            GZ[(size_t)k * (2 * H) + s * H + n] = acc[j][r] * silu_grad_fast(z);
        }
    }
}

// GZR[r, :] = sum over the sorted edges sharing U row r (ascending) of GZ[k, :], written in the form the transpose row GEMM
// consumes: fp32 (GZR), or three bf16 planes [3][R][W2] (GZRS, k_rowgemm_bf2<2H, H>); the f16x2 form is k_rowsum_h2 below
__global__ __launch_bounds__(256) void k_rowsum(int R, int W2, const int* __restrict__ row_ptr, const int* __restrict__ row_edge,
                                                const float* __restrict__ GZ, float* __restrict__ GZR, unsigned short* __restrict__ GZRS) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int W4 = W2 / 4;
    if (idx >= (long)R * W4) return;
    const int r = (int)(idx / W4), c = (int)(idx % W4) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = row_ptr[r]; q < row_ptr[r + 1]; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(GZ + (size_t)row_edge[q] * W2 + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const size_t o = (size_t)r * W2 + c, pl = (size_t)R * W2;
    if (GZRS) {
        const float h[4] = {acc.x, acc.y, acc.z, acc.w};
        unsigned short p1[4], p2[4], p3[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split3(h[e], p1[e], p2[e], p3[e]);
        *reinterpret_cast<uint2*>(GZRS + o) = make_uint2(p1[0] | ((unsigned)p1[1] << 16), p1[2] | ((unsigned)p1[3] << 16));
        *reinterpret_cast<uint2*>(GZRS + pl + o) = make_uint2(p2[0] | ((unsigned)p2[1] << 16), p2[2] | ((unsigned)p2[3] << 16));
        *reinterpret_cast<uint2*>(GZRS + 2 * pl + o) = make_uint2(p3[0] | ((unsigned)p3[1] << 16), p3[2] | ((unsigned)p3[3] << 16));
    } else {
        *reinterpret_cast<float4*>(GZR + o) = acc;
    }
}

#ifdef CCSP_EXPERIMENTS      // (round 3's separate row-sum launch, CCSP_ENERGY_ROWSUM=kernel: the product path forms the sums in k_edge_bwd_h2's epilogue)
// the f16x2 form of k_rowsum at hidden_dim 256 (W2 = 512): one wavefront per U row, two float4 column groups per lane, so the
// row maximum is one shuffle butterfly (no LDS, no barrier) and a row's edge indices are fetched four at a time ahead of the
// GZ rows they address (the sum keeps the ascending edge order)
__global__ __launch_bounds__(256) void k_rowsum_h2(int R, const int* __restrict__ row_ptr, const int* __restrict__ row_edge,
                                                   const float* __restrict__ GZ, unsigned short* __restrict__ GZRH, int* __restrict__ gexp,
                                                   const int* __restrict__ skip /*MALA reuse, or null*/) {
    if (skip && *skip == 0) return;
    constexpr int W2 = 512;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;                                           // (wave-uniform)
    const int q0 = row_ptr[r], q1 = row_ptr[r + 1];
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    for (int q = q0; q < q1; q += 4) {
        int ek[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ek[i] = row_edge[q + i < q1 ? q + i : q1 - 1];
        float4 v0[4], v1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* g = GZ + (size_t)ek[i] * W2 + lane * 4;
            v0[i] = *reinterpret_cast<const float4*>(g);
            v1[i] = *reinterpret_cast<const float4*>(g + 256);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (q + i < q1) {
                a0.x += v0[i].x; a0.y += v0[i].y; a0.z += v0[i].z; a0.w += v0[i].w;
                a1.x += v1[i].x; a1.y += v1[i].y; a1.z += v1[i].z; a1.w += v1[i].w;
            }
    }
    float m = fmaxf(fmaxf(fmaxf(fabsf(a0.x), fabsf(a0.y)), fmaxf(fabsf(a0.z), fabsf(a0.w))),
                    fmaxf(fmaxf(fabsf(a1.x), fabsf(a1.y)), fmaxf(fabsf(a1.z), fabsf(a1.w))));
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft));
    const int e = h2_scale_exp(m);
    if (lane == 0) gexp[r] = e;
    const size_t o = (size_t)r * W2 + lane * 4, pl = (size_t)R * W2;
    const float h[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    unsigned short p1[8], p2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) split2h(ldexpf(h[k], e), p1[k], p2[k]);
    *reinterpret_cast<uint2*>(GZRH + o) = make_uint2(p1[0] | ((unsigned)p1[1] << 16), p1[2] | ((unsigned)p1[3] << 16));
    *reinterpret_cast<uint2*>(GZRH + o + 256) = make_uint2(p1[4] | ((unsigned)p1[5] << 16), p1[6] | ((unsigned)p1[7] << 16));
    *reinterpret_cast<uint2*>(GZRH + pl + o) = make_uint2(p2[0] | ((unsigned)p2[1] << 16), p2[2] | ((unsigned)p2[3] << 16));
    *reinterpret_cast<uint2*>(GZRH + pl + o + 256) = make_uint2(p2[4] | ((unsigned)p2[5] << 16), p2[6] | ((unsigned)p2[7] << 16));
}
#endif  // CCSP_EXPERIMENTS

// ------------------------------------------------------------------------------------------
// k_node_energy: dE/dpose for a tile of 16 nodes
//   grad[n] = sum_CSR (-2 d)  +  W0^T ( SiLU'(y1) (.) W2^T ( SiLU'(y2) (.) sum_rows GP[r] ) )
// with y1 = W0 x + b0, y2 = W2 SiLU(y1) + b2 the pose-encoder pre-activations (recomputed).
// Block 0 also folds the per-workgroup energy partials of k_edge into E_out[0].
// ------------------------------------------------------------------------------------------
struct EnergyNodeArgs {
    int N, P;
    const int* node_ptr;     // node -> CSR range of (edge, slot) entries
    const float* Ocsr;       // [-2 d] per entry
    const int* nrow_ptr;     // node -> U rows
    const int* nrow_idx;
    const float* GP;         // [R, H]
    const float* x;          // evaluation point [N, P]
    float* grad;             // [N, P]
    const float* partial;    // energy partials
    int n_partial;
    float* E_out;
    const float* W0;         // pose_encoder.0.weight [H/2, P]
    const float* b0;
    const float* W2;         // pose_encoder.2.weight [H, H/2]
    const float* W2T;        // [H/2, H]
    const float* b2;
    const int* skip;         // MALA reuse: if non-null and *skip == 0 the launch returns at once (k_node_energy_h2)
    // composed domains (ccsp_compose_energy_grad): the encoder saw x_enc, not the comparison target x, and only its first
    // enc_cols inputs are variables (the rest are constants of the batch); defaults: x_enc = x, every column
    const float* x_enc;      // or null
    int enc_cols;            // 0 = all P columns
    int* skip_count;         // MALA reuse: += 1 per skipped gradient evaluation (or null)
};

template <int H>
__global__ __launch_bounds__(256) void k_node_energy(EnergyNodeArgs a) {
    constexpr int KC = H / 2;
    __shared__ float xs[NODE_TILE][8];
    __shared__ float dir[NODE_TILE][8];
    __shared__ float y1[NODE_TILE][KC + 1];     // pre-activation, then g_y1
    __shared__ float s1[NODE_TILE][KC + 1];
    __shared__ float gy2[NODE_TILE][H + 1];
    __shared__ float red[256];
    const int node0 = blockIdx.x * NODE_TILE;
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && a.E_out) {            // uniform branch: whole block participates
        float v = 0.0f;
        for (int i = tid; i < a.n_partial; i += 256) v += a.partial[i];
        const float s = block_sum_256(v, red);
        if (tid == 0) a.E_out[0] = s;
        __syncthreads();
    }
    if (tid < NODE_TILE * 8) {
        const int nl = tid / 8, p = tid % 8, n = node0 + nl;
        float xv = 0.0f, dv = 0.0f;
        if (n < a.N && p < a.P) {
            xv = (a.x_enc ? a.x_enc : a.x)[(size_t)n * a.P + p];
            const int beg = a.node_ptr[n], end = a.node_ptr[n + 1];
            const float* op = a.Ocsr + (size_t)beg * a.P + p;
            for (int q = 0; q < end - beg; ++q) dv += op[(size_t)q * a.P];
        }
        xs[nl][p] = xv;
        dir[nl][p] = dv;
    }
    __syncthreads();
    for (int idx = tid; idx < NODE_TILE * KC; idx += 256) {
        const int n = idx / KC, j = idx % KC;
        float acc = 0.0f;
        for (int d = 0; d < a.P; ++d) acc = fmaf(xs[n][d], a.W0[j * a.P + d], acc);
        acc += a.b0[j];
        y1[n][j] = acc;
        s1[n][j] = silu_fast(acc);
    }
    __syncthreads();
    {   // y2 and g_y2 = (sum_rows GP) * SiLU'(y2)
        static_assert(256 % H == 0, "k_node_energy (the VALU form, CCSP_NODE_ENERGY_VALU) is written for widths that divide 256");
        constexpr int NG = 256 / H, NPT = NODE_TILE / NG;
        const int j = tid % H, g = tid / H;
        float acc[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) acc[i] = 0.0f;
        for (int k = 0; k < KC; ++k) {
            const float wv = a.W2T[(size_t)k * H + j];
#pragma unroll
            for (int i = 0; i < NPT; ++i) acc[i] = fmaf(s1[g * NPT + i][k], wv, acc[i]);
        }
        const float bj = a.b2[j];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int nl = g * NPT + i, n = node0 + nl;
            float gp = 0.0f;
            if (n < a.N)
                for (int q = a.nrow_ptr[n]; q < a.nrow_ptr[n + 1]; ++q) gp += a.GP[(size_t)a.nrow_idx[q] * H + j];
            gy2[nl][j] = gp * silu_grad_fast(acc[i] + bj);
        }
    }
    __syncthreads();
    {   // g_s1[k] = sum_j g_y2[j] W2[j, k];  g_y1 = g_s1 * SiLU'(y1)
        constexpr int NG = 256 / KC, NPT = NODE_TILE / NG;
        const int k = tid % KC, g = tid / KC;
        float acc[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) acc[i] = 0.0f;
        for (int j = 0; j < H; ++j) {
            const float wv = a.W2[(size_t)j * KC + k];
#pragma unroll
            for (int i = 0; i < NPT; ++i) acc[i] = fmaf(gy2[g * NPT + i][j], wv, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int nl = g * NPT + i;
            acc[i] *= silu_grad_fast(y1[nl][k]);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NPT; ++i) y1[g * NPT + i][k] = acc[i];
    }
    __syncthreads();
    if (tid < NODE_TILE * 8) {
        const int nl = tid / 8, p = tid % 8, n = node0 + nl;
        if (n < a.N && p < a.P) {
            float gx = 0.0f;
            for (int k = 0; k < KC; ++k) gx = fmaf(y1[nl][k], a.W0[k * a.P + p], gx);
            a.grad[(size_t)n * a.P + p] = (a.enc_cols && p >= a.enc_cols) ? dir[nl][p] : dir[nl][p] + gx;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_node_energy_mfma: the same computation with both H x H/2 products of the pose-encoder forward /
// backward on v_mfma_f32_16x16x4_f32 (M = the 16 nodes of the tile) instead of 256 threads walking the
// K loops with one global weight load per step: 55 us -> see DESIGN.md (it was 17 % of a MALA chain).
// Products are computed transposed (weights as the A operand) so that a lane owns four consecutive
// columns of one node: the GP row sums are float4 loads and the LDS tiles are written 16 bytes at a time.
// ------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(256) void k_node_energy_mfma(EnergyNodeArgs a, const float* __restrict__ W2F /*pose_encoder.2 in fragment order*/) {
    constexpr int KC = H / 2, TPW = H / 64, KS = H / 8;          // forward: K = H/2 in steps of 4, TPW column tiles per wave
    constexpr int BT = KC / 16;                                   // backward: 16-wide tiles of the H/2 hidden units
    constexpr int BTW = (BT + 3) / 4;                             // ... per wave (the last wave's share may be short: `live` below)
    constexpr int PF = KS >= 16 ? 8 : KS / 2;                     // forward fragments requested at kernel entry
    __shared__ float xs[NODE_TILE][8];
    __shared__ float dir[NODE_TILE][8];
    __shared__ float y1[NODE_TILE][KC + 4];                       // pre-activation, later g_y1
    __shared__ float s1[NODE_TILE][KC + 1];
    __shared__ float gy2[NODE_TILE][H + 4];
    __shared__ float w0s[KC][9];                                  // pose_encoder.0.weight, columns >= P are 0 (stride 9: conflict-free)
    __shared__ float red[256];
    const int node0 = blockIdx.x * NODE_TILE;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nl = lane & 15, n = node0 + nl;
    // The kernel is a latency chain on 200-odd workgroups (CSR ranges -> rows -> two dependent small GEMMs), so every load
    // that does not depend on the chain is requested here, in the order it will be consumed: first-layer weights, the
    // node's U-row range, the first forward fragments.
    constexpr int NW0 = (KC * 8 + 255) / 256;                      // first-layer weight elements per thread (4 at H = 256)
    float w0r[NW0];
#pragma unroll
    for (int i = 0; i < NW0; ++i) {
        const int idx = tid + 256 * i;
        const int j = idx >> 3, d = idx & 7;
        w0r[i] = (idx < KC * 8 && d < a.P) ? a.W0[j * a.P + d] : 0.0f;
    }
    const int rb = n < a.N ? a.nrow_ptr[n] : 0, re = n < a.N ? a.nrow_ptr[n + 1] : 0;
    const float* wf = W2F + ((size_t)wave * KS * 64 + lane) * TPW;
    float wpf[PF][TPW];
#pragma unroll
    for (int ks = 0; ks < PF; ++ks)
#pragma unroll
        for (int q = 0; q < TPW; ++q) wpf[ks][q] = wf[(size_t)ks * 64 * TPW + q];
    float4 bj[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) bj[j] = *reinterpret_cast<const float4*>(a.b2 + wave * 16 * TPW + j * 16 + 4 * (lane >> 4));
    const float b0j = a.b0[tid % KC];
    __builtin_amdgcn_sched_barrier(0);
    if (blockIdx.x == 0 && a.E_out) {            // uniform branch: whole block participates
        float v = 0.0f;
        for (int i = tid; i < a.n_partial; i += 256) v += a.partial[i];
        const float s = block_sum_256(v, red);
        if (tid == 0) a.E_out[0] = s;
        __syncthreads();
    }
    if (tid < NODE_TILE * 8) {
        const int nl1 = tid / 8, p = tid % 8, n1 = node0 + nl1;
        float xv = 0.0f, dv = 0.0f;
        if (n1 < a.N && p < a.P) {
            xv = (a.x_enc ? a.x_enc : a.x)[(size_t)n1 * a.P + p];
            const int beg = a.node_ptr[n1], end = a.node_ptr[n1 + 1];
            const float* op = a.Ocsr + (size_t)beg * a.P + p;
#pragma unroll 8
            for (int q = 0; q < end - beg; ++q) dv += op[(size_t)q * a.P];
        }
        xs[nl1][p] = xv;
        dir[nl1][p] = dv;
    }
#pragma unroll
    for (int i = 0; i < NW0; ++i) {
        const int idx = tid + 256 * i;
        if (idx < KC * 8) w0s[idx >> 3][idx & 7] = w0r[i];
    }
    // sum of the node's GP rows (ascending row order), independent of the pose: in flight under the first layer
    float4 gp[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) gp[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const int c00 = wave * 16 * TPW + 4 * (lane >> 4);
        for (int q = rb; q < re; ++q) {
            const float* row = a.GP + (size_t)a.nrow_idx[q] * H + c00;
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const float4 v = *reinterpret_cast<const float4*>(row + j * 16);
                gp[j].x += v.x; gp[j].y += v.y; gp[j].z += v.z; gp[j].w += v.w;
            }
        }
    }
    __syncthreads();
    if constexpr (256 % KC == 0) {
        const int j = tid % KC;
#pragma unroll
        for (int i = 0; i < NODE_TILE * KC / 256; ++i) {
            const int nn = tid / KC + i * (256 / KC);
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < 8; ++d) acc = fmaf(xs[nn][d], w0s[j][d], acc);      // xs / w0s columns >= P are 0
            acc += b0j;
            y1[nn][j] = acc;
            s1[nn][j] = silu_fast(acc);
        }
    } else {                                     // hidden widths whose half does not divide 256
        for (int idx = tid; idx < NODE_TILE * KC; idx += 256) {
            const int nn = idx / KC, j = idx % KC;
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < 8; ++d) acc = fmaf(xs[nn][d], w0s[j][d], acc);
            acc += a.b0[j];
            y1[nn][j] = acc;
            s1[nn][j] = silu_fast(acc);
        }
    }
    __syncthreads();
    {   // y2^T tiles = W2 . s1^T; g_y2 = (sum of the node's GP rows) * SiLU'(y2)
        floatx4 acc[TPW];
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[j] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        const float* bp = &s1[nl][lane >> 4];
        float rest[KS - PF][TPW];                                 // the remaining fragments, requested before the first MFMA
#pragma unroll
        for (int ks = PF; ks < KS; ++ks)
#pragma unroll
            for (int q = 0; q < TPW; ++q) rest[ks - PF][q] = wf[(size_t)ks * 64 * TPW + q];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float b = bp[ks * 4];
#pragma unroll
            for (int j = 0; j < TPW; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ks < PF ? wpf[ks][j] : rest[ks - PF][j], b, acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            const int c0 = wave * 16 * TPW + j * 16 + 4 * (lane >> 4);
            *reinterpret_cast<float4*>(&gy2[nl][c0]) =
                make_float4(gp[j].x * silu_grad_fast(acc[j][0] + bj[j].x), gp[j].y * silu_grad_fast(acc[j][1] + bj[j].y),
                            gp[j].z * silu_grad_fast(acc[j][2] + bj[j].z), gp[j].w * silu_grad_fast(acc[j][3] + bj[j].w));
        }
    }
    // backward weights: A[i = hidden unit][c] = W2[c][i]; half of them requested ahead of the barrier
    constexpr int KB = H / 4, KBH = KB / 2;
    const float* ap = a.W2 + (size_t)(lane >> 4) * KC + wave * 16 * BTW + (lane & 15);
    float wb0[KBH][BTW];
    auto live = [&](int t) { return wave * BTW + t < BT; };       // (wave-uniform) tile t of this wave exists; others read tile 0's weights and store nothing
    if (live(0)) {
#pragma unroll
        for (int ks = 0; ks < KBH; ++ks)
#pragma unroll
            for (int t = 0; t < BTW; ++t) wb0[ks][t] = ap[(size_t)ks * 4 * KC + (live(t) ? t : 0) * 16];
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (live(0)) {   // g_s1^T tiles = W2^T . g_y2^T  (contraction over the H outputs);  g_y1 = g_s1 * SiLU'(y1)
        floatx4 acc[BTW];
#pragma unroll
        for (int t = 0; t < BTW; ++t) acc[t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        const float* bp = &gy2[nl][lane >> 4];
        float wb1[KB - KBH][BTW];
#pragma unroll
        for (int ks = KBH; ks < KB; ++ks)
#pragma unroll
            for (int t = 0; t < BTW; ++t) wb1[ks - KBH][t] = ap[(size_t)ks * 4 * KC + (live(t) ? t : 0) * 16];
#pragma unroll
        for (int ks = 0; ks < KB; ++ks) {
            const float b = bp[ks * 4];
#pragma unroll
            for (int t = 0; t < BTW; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ks < KBH ? wb0[ks][t] : wb1[ks - KBH][t], b, acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < BTW; ++t) {
            if (!live(t)) continue;
            const int k0 = wave * 16 * BTW + t * 16 + 4 * (lane >> 4);
            float4 g;
            g.x = acc[t][0] * silu_grad_fast(y1[nl][k0]);     g.y = acc[t][1] * silu_grad_fast(y1[nl][k0 + 1]);
            g.z = acc[t][2] * silu_grad_fast(y1[nl][k0 + 2]); g.w = acc[t][3] * silu_grad_fast(y1[nl][k0 + 3]);
            *reinterpret_cast<float4*>(&y1[nl][k0]) = g;      // each (node, unit) is read and written by this lane only
        }
    }
    __syncthreads();
    {   // grad = direct term + W0^T g_y1: two threads per (node, component), each half of the hidden units in ascending order
        const int half = tid >> 7, t7 = tid & 127;
        const int nl2 = t7 / 8, p = t7 % 8;
        float gx = 0.0f;
#pragma unroll 16
        for (int k = half * (KC / 2); k < (half + 1) * (KC / 2); ++k) gx = fmaf(y1[nl2][k], w0s[k][p], gx);
        if (half == 1) red[t7] = gx;
        __syncthreads();
        const int n2 = node0 + nl2;
        if (half == 0 && n2 < a.N && p < a.P) a.grad[(size_t)n2 * a.P + p] = (a.enc_cols && p >= a.enc_cols) ? dir[nl2][p] : dir[nl2][p] + (gx + red[t7]);
    }
}

// ------------------------------------------------------------------------------------------
// k_node_energy_h2: k_node_energy_mfma at hidden_dim 256 with both 16 x 256 x 128 products on the f16 matrix pipe (three products
// of two-term fp16 operands, ccsp_f16x2.h; 2 x 768 cycles of v_mfma_f32_16x16x32_f16 per wave instead of 2 x 4096 of
// v_mfma_f32_16x16x4_f32).  Operand scaling: the weight by one exponent (planes packed by k_pack_enc_frag_h2 for the forward
// product, k_pack_enc_frag_h2t for W2^T); the layer-1 activations per node by the bound c1 max|x| + c2 (as in encode_tile_h2);
// g_y2 per node by its exact row maximum (shuffles + one LDS exchange).  Loads are issued in the order of the dependent chains
// (vector-memory loads return in order): U-row ranges and CSR ranges, row indices and edge outputs, GP rows, then the weights.
//   W2TH[plane][(((w * 8 + ks) * 2 + t) * 64 + l) * 8 + e8] = term of W2[ks*32 + 8 (l >> 4) + e8][w*32 + t*16 + (l & 15)] * 2^e
// ------------------------------------------------------------------------------------------
__global__ void k_pack_enc_frag_h2t(const float* __restrict__ W2 /*[256,128]*/, int e, unsigned short* __restrict__ W2TH) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 256 * 128) return;
    const int e8 = idx & 7, l = (idx >> 3) & 63, t = (idx >> 9) & 1, ks = (idx >> 10) & 7, w = idx >> 13;
    const int cc = ks * 32 + 8 * (l >> 4) + e8, i = w * 32 + t * 16 + (l & 15);
    unsigned short a, b;
    split2h(ldexpf(W2[cc * 128 + i], e), a, b);
    W2TH[idx] = a;
    W2TH[256 * 128 + idx] = b;
}

__device__ __forceinline__ void node_energy_h2_body(const EnergyNodeArgs& a, const EncW& w, const unsigned short* __restrict__ W2TH) {
    constexpr int H = 256, KC = 128, LD1 = ENC_H2_LD, LD2 = H + 8;       // fp16 row strides: 16-byte fragment reads hit all banks once
    __shared__ float xs[NODE_TILE][8];
    __shared__ float dir[NODE_TILE][8];
    __shared__ float y1[NODE_TILE][KC + 4];                       // pre-activation, later g_y1
    __shared__ __attribute__((aligned(16))) unsigned short s1h[2 * NODE_TILE * LD1];
    __shared__ __attribute__((aligned(16))) unsigned short g2h[2 * NODE_TILE * LD2];
    __shared__ float w0s[KC][9];                                  // pose_encoder.0.weight, columns >= P are 0 (stride 9: conflict-free)
    __shared__ float red[256];
    __shared__ float smax[4][NODE_TILE];
    __shared__ int sexp[NODE_TILE];
    const int node0 = blockIdx.x * NODE_TILE;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nl = lane & 15, n = node0 + nl;
    // ---- chain heads
    const int rb = n < a.N ? a.nrow_ptr[n] : 0, re = n < a.N ? a.nrow_ptr[n + 1] : 0;
    int beg = 0, cnt = 0;
    float xv = 0.0f;
    const int nl1 = tid / 8, p1 = tid % 8, n1 = node0 + nl1;
    const bool upd = tid < NODE_TILE * 8 && n1 < a.N && p1 < a.P;
    if (upd) {
        beg = a.node_ptr[n1];
        cnt = a.node_ptr[n1 + 1] - beg;
        xv = (a.x_enc ? a.x_enc : a.x)[(size_t)n1 * a.P + p1];
    }
    // ---- second links: the first four U rows of the lane's node, the first sixteen CSR entries
    const int c00 = wave * 64 + 4 * (lane >> 4);
    int id0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) id0[i] = rb < re ? a.nrow_idx[rb + i < re ? rb + i : re - 1] : 0;
    float ov[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) ov[j] = 0.0f;
    if (upd) {
        const float* op = a.Ocsr + (size_t)beg * a.P + p1;
#pragma unroll
        for (int j = 0; j < 16; ++j) ov[j] = j < cnt ? op[(size_t)j * a.P] : 0.0f;
    }
    float4 gv[4][4];
    if (rb < re) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[i][j] = *reinterpret_cast<const float4*>(a.GP + (size_t)id0[i] * H + c00 + j * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- the weights, behind the chains
    float w0r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;                            // KC * 8 = 4 * 256
        const int j = idx >> 3, d = idx & 7;
        w0r[i] = d < a.P ? a.W0[j * a.P + d] : 0.0f;
    }
    float w0t[8];                                                 // layer-1 row of this thread's hidden unit
    {
        const int j = tid % KC;
#pragma unroll
        for (int d = 0; d < 8; ++d) w0t[d] = d < a.P ? a.W0[j * a.P + d] : 0.0f;
    }
    const float b0j = a.b0[tid % KC];
    const half8* wh = reinterpret_cast<const half8*>(w.W2H) + (size_t)wave * 16 * 64 + lane;
    half8 wa[4][2][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int q = 0; q < 4; ++q) wa[ks][pp][q] = wh[(size_t)pp * 4096 + (ks * 4 + q) * 64];
    float4 bj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bj[j] = *reinterpret_cast<const float4*>(a.b2 + c00 + j * 16);
    __builtin_amdgcn_sched_barrier(0);
    if (blockIdx.x == 0 && a.E_out) {            // uniform branch: whole block participates
        float v = 0.0f;
        for (int i = tid; i < a.n_partial; i += 256) v += a.partial[i];
        const float s = block_sum_256(v, red);
        if (tid == 0) a.E_out[0] = s;
        __syncthreads();
    }
    // ---- direct term, poses and row exponents
    if (tid < NODE_TILE * 8) {
        float dv = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) dv += ov[j];
        if (upd) {
            const float* op = a.Ocsr + (size_t)beg * a.P + p1;
            for (int q0 = 16; q0 < cnt; q0 += 16) {
                float v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = q0 + j < cnt ? op[(size_t)(q0 + j) * a.P] : 0.0f;
#pragma unroll
                for (int j = 0; j < 16; ++j) dv += v[j];
            }
        }
        xs[nl1][p1] = xv;
        dir[nl1][p1] = dv;
        float amax = fabsf(xv);
        amax = fmaxf(amax, __shfl_xor(amax, 1));
        amax = fmaxf(amax, __shfl_xor(amax, 2));
        amax = fmaxf(amax, __shfl_xor(amax, 4));
        if (p1 == 0) sexp[nl1] = h2_scale_exp(fmaf(w.c1, amax, w.c2));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        w0s[idx >> 3][idx & 7] = w0r[i];
    }
    // ---- sum of the node's GP rows in ascending row order (first four rows already in flight)
    float4 gp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) gp[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = rb; q < re; q += 4) {
        if (q > rb) {
            int id[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) id[i] = a.nrow_idx[q + i < re ? q + i : re - 1];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) gv[i][j] = *reinterpret_cast<const float4*>(a.GP + (size_t)id[i] * H + c00 + j * 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (q + i < re) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { gp[j].x += gv[i][j].x; gp[j].y += gv[i][j].y; gp[j].z += gv[i][j].z; gp[j].w += gv[i][j].w; }
            }
    }
    __syncthreads();
    {   // layer 1: pre-activations kept in fp32 (SiLU' in the backward), activations as fp16 planes
        const int j = tid % KC;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int nn = tid / KC + 2 * i;
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < 8; ++d) acc = fmaf(xs[nn][d], w0t[d], acc);
            acc += b0j;
            y1[nn][j] = acc;
            unsigned short h1, h2;
            split2h(ldexpf(silu_fast(acc), sexp[nn]), h1, h2);
            s1h[nn * LD1 + j] = h1;
            s1h[(NODE_TILE + nn) * LD1 + j] = h2;
        }
    }
    __syncthreads();
    const half8* wt = reinterpret_cast<const half8*>(W2TH) + (size_t)wave * 16 * 64 + lane;
    half8 wb[8][2][2];                                            // backward fragments [k-step][plane][tile]
    {   // y2^T tiles = W2 . s1^T; g_y2 = (sum of the node's GP rows) * SiLU'(y2)
        floatx4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        const unsigned short* bp = s1h + nl * LD1 + 8 * (lane >> 4);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const half8 b1 = *reinterpret_cast<const half8*>(bp + ks * 32);
            const half8 b2 = *reinterpret_cast<const half8*>(bp + NODE_TILE * LD1 + ks * 32);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ks][1][j], b1, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ks][0][j], b2, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ks][0][j], b1, acc[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)                            // (the forward fragments are dead: their registers take these)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int t = 0; t < 2; ++t) wb[ks][pp][t] = wt[(size_t)pp * 4096 + (ks * 2 + t) * 64];
        __builtin_amdgcn_sched_barrier(0);
        const int eu = -(sexp[nl] + w.w2_exp);
        float g[4][4];
        float m = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gpj[4] = {gp[j].x, gp[j].y, gp[j].z, gp[j].w};
            const float bjj[4] = {bj[j].x, bj[j].y, bj[j].z, bj[j].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                g[j][r] = gpj[r] * silu_grad_fast(ldexpf(acc[j][r], eu) + bjj[r]);
                m = fmaxf(m, fabsf(g[j][r]));
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        if (lane < NODE_TILE) smax[wave][lane] = m;
        __syncthreads();
        m = fmaxf(fmaxf(smax[0][nl], smax[1][nl]), fmaxf(smax[2][nl], smax[3][nl]));
        const int eg = h2_scale_exp(m);
        if (wave == 0 && lane < NODE_TILE) sexp[lane] = eg;      // (layer 1's exponents are dead: every wave read its own above)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned short h1[4], h2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) split2h(ldexpf(g[j][r], eg), h1[r], h2[r]);
            const int c0 = c00 + j * 16;
            *reinterpret_cast<uint2*>(g2h + nl * LD2 + c0) = make_uint2(h1[0] | ((unsigned)h1[1] << 16), h1[2] | ((unsigned)h1[3] << 16));
            *reinterpret_cast<uint2*>(g2h + (NODE_TILE + nl) * LD2 + c0) = make_uint2(h2[0] | ((unsigned)h2[1] << 16), h2[2] | ((unsigned)h2[3] << 16));
        }
    }
    __syncthreads();
    {   // g_s1^T tiles = W2^T . g_y2^T  (contraction over the H outputs);  g_y1 = g_s1 * SiLU'(y1)
        floatx4 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        const unsigned short* bp = g2h + nl * LD2 + 8 * (lane >> 4);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const half8 b1 = *reinterpret_cast<const half8*>(bp + ks * 32);
            const half8 b2 = *reinterpret_cast<const half8*>(bp + NODE_TILE * LD2 + ks * 32);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ks][1][t], b1, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ks][0][t], b2, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ks][0][t], b1, acc[t], 0, 0, 0);
        }
        const int eu = -(sexp[nl] + w.w2_exp);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int k0 = wave * 32 + t * 16 + 4 * (lane >> 4);
            float4 gq;
            gq.x = ldexpf(acc[t][0], eu) * silu_grad_fast(y1[nl][k0]);     gq.y = ldexpf(acc[t][1], eu) * silu_grad_fast(y1[nl][k0 + 1]);
            gq.z = ldexpf(acc[t][2], eu) * silu_grad_fast(y1[nl][k0 + 2]); gq.w = ldexpf(acc[t][3], eu) * silu_grad_fast(y1[nl][k0 + 3]);
            *reinterpret_cast<float4*>(&y1[nl][k0]) = gq;     // each (node, unit) is read and written by this lane only
        }
    }
    __syncthreads();
    {   // grad = direct term + W0^T g_y1: two threads per (node, component), each half of the hidden units in ascending order
        const int half = tid >> 7, t7 = tid & 127;
        const int nl2 = t7 / 8, p = t7 % 8;
        float gx = 0.0f;
#pragma unroll 16
        for (int k = half * (KC / 2); k < (half + 1) * (KC / 2); ++k) gx = fmaf(y1[nl2][k], w0s[k][p], gx);
        if (half == 1) red[t7] = gx;
        __syncthreads();
        const int n2 = node0 + nl2;
        if (half == 0 && n2 < a.N && p < a.P) a.grad[(size_t)n2 * a.P + p] = (a.enc_cols && p >= a.enc_cols) ? dir[nl2][p] : dir[nl2][p] + (gx + red[t7]);
    }
}

__global__ __launch_bounds__(256) void k_node_energy_h2(EnergyNodeArgs a, EncW w, const unsigned short* __restrict__ W2TH) {
    if (a.skip && *a.skip == 0) {                                 // (uniform) MALA reuse: gradient and E(x) of the unmoved state stand
        if (blockIdx.x == 0 && threadIdx.x == 0 && a.skip_count) atomicAdd(a.skip_count, 1);
        return;
    }
    node_energy_h2_body(a, w, W2TH);
}

// The gradient evaluation's last kernel and the update that consumes it in ONE launch (MALA's propose step, the Langevin step of an
// energy-mode ULA chain): both work on the same 16-node blocks and a thread of the update reads exactly the gradient element the
// same thread stored a moment ago, so nothing but the launch boundary separated them (5 launches per gradient evaluation and
// update instead of 6).  The MALA-reuse flag the evaluation reads and the one the update resets are different words (the inner
// step's parity picks them: chain_run_impl), so a block that runs ahead cannot change what a later block reads.
__global__ __launch_bounds__(256) void k_node_energy_h2_update(EnergyNodeArgs a, EncW w, const unsigned short* __restrict__ W2TH, NodeArgs b, EncW wb, EncOut eo) {
    if (a.skip && *a.skip == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && a.skip_count) atomicAdd(a.skip_count, 1);
    } else {
        node_energy_h2_body(a, w, W2TH);
    }
    __syncthreads();
    node_body<256, true>(b, wb, eo);
}

// acceptance counts -> mean acceptance rate per timestep
// ... of a batch that ran as two coupled lanes: decisions accepted / decisions made over both
__global__ void k_accept_rates2(int T, const int* __restrict__ c0, const int* __restrict__ d0, const int* __restrict__ c1, const int* __restrict__ d1,
                                float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) out[t] = d0[t] + d1[t] > 0 ? (float)(c0[t] + c1[t]) / (float)(d0[t] + d1[t]) : 0.0f;
}
__global__ void k_accept_rates(int T, const int* __restrict__ count, const int* __restrict__ denom, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) out[t] = denom[t] > 0 ? (float)count[t] / (float)denom[t] : 0.0f;
}
