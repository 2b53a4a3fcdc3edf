// StructDiffusion baseline (reference networks/denoise_fn.py:267-282,391-451, networks/transformer.py:43-82):
// every graph is a sequence of 8 tokens (padded), width Wd = 2H (3H with a grasp group), 4 pre-LN blocks,
// 2 heads.  Device layout: token row = graph * 8 + position, all activations row-major [M = 8 B, width].
//
// One evaluation = k_sd_embed, then per block  k_sd_ln -> k_sd_gemm(in_proj) -> k_sd_attn ->
// k_sd_gemm(out_proj, +residual) -> k_sd_gemm(c_fc, QuickGELU) -> k_sd_gemm(c_proj) -> k_sd_ln(+residual),
// then k_sd_decode (ln_post, last H channels, pose decoder, mask fill).  The GEMMs are fp32 MFMA 32x32x2
// tiles through LDS (the same core as k_rowgemm); 24 M Wd^2 flops per block dominate (MFMA bound).
// Included by ccsp_hip.hip.
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

// nn.LayerNorm(eps 1e-5) of the row held as v[i] = x[lane + 64 i]; two-pass mean / biased variance
__device__ __forceinline__ void ln_row(float (&v)[SD_MAXV], int Wd, int lane, const float* __restrict__ gam, const float* __restrict__ bet) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < SD_MAXV; ++i) if (lane + 64 * i < Wd) s += v[i];
    const float mean = wave_sum(s) / (float)Wd;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < SD_MAXV; ++i) if (lane + 64 * i < Wd) { const float d = v[i] - mean; q += d * d; }
    const float inv = 1.0f / sqrtf(wave_sum(q) / (float)Wd + 1e-5f);
#pragma unroll
    for (int i = 0; i < SD_MAXV; ++i) if (lane + 64 * i < Wd) v[i] = (v[i] - mean) * inv * gam[lane + 64 * i] + bet[lane + 64 * i];
}

// Y[r] = LN(X[r])  (ACC = 0)   or   Y[r] += LN(X[r])  (ACC = 1: x = x + ln_2(mlp(x)), transformer.py:66)
template <int ACC>
__global__ __launch_bounds__(256) void k_sd_ln(int M, int Wd, const float* __restrict__ X, const float* __restrict__ gam,
                                               const float* __restrict__ bet, float* __restrict__ Y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    float v[SD_MAXV];
#pragma unroll
    for (int i = 0; i < SD_MAXV; ++i) v[i] = lane + 64 * i < Wd ? X[(size_t)row * Wd + lane + 64 * i] : 0.0f;
    ln_row(v, Wd, lane, gam, bet);
#pragma unroll
    for (int i = 0; i < SD_MAXV; ++i)
        if (lane + 64 * i < Wd) {
            float* dst = Y + (size_t)row * Wd + lane + 64 * i;
            *dst = ACC ? *dst + v[i] : v[i];
        }
}

// token rows: [grasp_emb] geoms_emb (poses_emb + time_emb) + pe[position] -> ln_pre; padding rows are zero
// (denoise_fn.py:397-423)
__global__ __launch_bounds__(256) void k_sd_embed(int M, int H, int Wd, int grasp, const int* __restrict__ tok_node,
                                                  const int* __restrict__ tok_pos, const float* __restrict__ gemb,
                                                  const float* __restrict__ remb, const float* __restrict__ pemb,
                                                  const float* __restrict__ temb_t, const float* __restrict__ pe,
                                                  const float* __restrict__ gam, const float* __restrict__ bet, float* __restrict__ X) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int n = tok_node[row];
    float v[SD_MAXV];
    if (n < 0) {
#pragma unroll
        for (int i = 0; i < SD_MAXV; ++i) if (lane + 64 * i < Wd) X[(size_t)row * Wd + lane + 64 * i] = 0.0f;
        return;
    }
    const int off = grasp ? H : 0;
    const float* per = pe + (size_t)tok_pos[row] * Wd;
#pragma unroll
    for (int i = 0; i < SD_MAXV; ++i) {
        const int c = lane + 64 * i;
        float e = 0.0f;
        if (c < Wd) {
            if (c < off) e = remb[(size_t)n * H + c];
            else if (c < off + H) e = gemb[(size_t)n * H + c - off];
            else e = pemb[(size_t)n * H + c - off - H] + temb_t[c - off - H];
            e += per[c];
        }
        v[i] = e;
    }
    ln_row(v, Wd, lane, gam, bet);
#pragma unroll
    for (int i = 0; i < SD_MAXV; ++i) if (lane + 64 * i < Wd) X[(size_t)row * Wd + lane + 64 * i] = v[i];
}

// nn.MultiheadAttention core for one (graph, head): 8 x 8 scores, the FLOAT pad mask added to them
// (denoise_fn.py:426-434: +1.0 where a row or column is padding; with no padding `[-0:]` marks
// everything).  mask_from[b * heads + h] = first padded index of the graph whose mask this head
// sees, (b heads + h) mod B -- the reference repeats the masks graph-major while MHA reads them
// head-major.
constexpr int SD_DH_MAX = 384;
__global__ __launch_bounds__(256) void k_sd_attn(int Wd, const float* __restrict__ QKV, const int* __restrict__ mask_from,
                                                 float* __restrict__ Aout) {
    __shared__ float qkv[3][SD_L][SD_DH_MAX + 1];
    __shared__ float part[4][SD_L * SD_L];
    __shared__ float ps[SD_L][SD_L];
    const int b = blockIdx.x / SD_HEADS, h = blockIdx.x % SD_HEADS;
    const int DH = Wd / SD_HEADS, tid = threadIdx.x;
    const float* base = QKV + (size_t)b * SD_L * 3 * Wd + h * DH;
    for (int idx = tid; idx < SD_L * DH; idx += 256) {
        const int r = idx / DH, c = idx % DH;
        const float* src = base + (size_t)r * 3 * Wd + c;
        qkv[0][r][c] = src[0];
        qkv[1][r][c] = src[Wd];
        qkv[2][r][c] = src[2 * Wd];
    }
    __syncthreads();
    // 64 (query, key) pairs x 4 quarters of the head dimension; the quarters are added in order 0..3
    const int pair = tid & 63, qt = tid >> 6;
    const int i = pair >> 3, j = pair & 7;
    const float scale = 1.0f / sqrtf((float)DH);          // q is scaled before the product, like F.multi_head_attention_forward
    {
        const int c0 = qt * (DH / 4), c1 = c0 + DH / 4;
        float s = 0.0f;
        for (int c = c0; c < c1; ++c) s += (qkv[0][i][c] * scale) * qkv[1][j][c];
        part[qt][pair] = s;
    }
    __syncthreads();
    if (tid < 64) {
        float s = ((part[0][pair] + part[1][pair]) + part[2][pair]) + part[3][pair];
        const int from = mask_from[blockIdx.x];
        s += (i >= from || j >= from) ? 1.0f : 0.0f;
        float mx = s;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float e = expf(s - mx);
        float den = e;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) den += __shfl_xor(den, o);
        ps[i][j] = e / den;
    }
    __syncthreads();
    for (int idx = tid; idx < SD_L * DH; idx += 256) {
        const int r = idx / DH, c = idx % DH;
        float o = 0.0f;
#pragma unroll
        for (int jj = 0; jj < SD_L; ++jj) o += ps[r][jj] * qkv[2][jj][c];
        Aout[((size_t)b * SD_L + r) * Wd + h * DH + c] = o;
    }
}

// per node: ln_post of its token row, last H channels -> pose_decoder -> eps; masked nodes take
// batch.x[:, -P:] (denoise_fn.py:437-449).  One wave per node.
template <int H>
__global__ __launch_bounds__(256) void k_sd_decode(int N, int Wd, int P, int F, const int* __restrict__ node_tok, const float* __restrict__ X,
                                                   const float* __restrict__ gam, const float* __restrict__ bet,
                                                   const float* __restrict__ pd0_wT /*[H][H/2]*/, const float* __restrict__ pd0_b,
                                                   const float* __restrict__ pd2_w /*[P][H/2]*/, const float* __restrict__ pd2_b,
                                                   const float* __restrict__ xfeat, const signed char* __restrict__ mask,
                                                   float* __restrict__ eps) {
    __shared__ float ys[4][H];
    __shared__ float hs[4][H / 2];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + w;
    const bool live = n < N;
    float v[SD_MAXV];
    if (live) {
        const int row = node_tok[n];
#pragma unroll
        for (int i = 0; i < SD_MAXV; ++i) v[i] = lane + 64 * i < Wd ? X[(size_t)row * Wd + lane + 64 * i] : 0.0f;
        ln_row(v, Wd, lane, gam, bet);
#pragma unroll
        for (int i = 0; i < SD_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < Wd && c >= Wd - H) ys[w][c - (Wd - H)] = v[i];
        }
    }
    __syncthreads();
    if (live) {
        for (int k = lane; k < H / 2; k += 64) {
            float q = pd0_b[k];
            for (int c = 0; c < H; ++c) q += ys[w][c] * pd0_wT[(size_t)c * (H / 2) + k];
            hs[w][k] = silu_f(q);
        }
    }
    __syncthreads();
    if (!live) return;
    for (int p = 0; p < P; ++p) {
        float part = 0.0f;
        for (int k = lane; k < H / 2; k += 64) part += hs[w][k] * pd2_w[(size_t)p * (H / 2) + k];
        const float o = wave_sum(part) + pd2_b[p];
        if (lane == 0) eps[(size_t)n * P + p] = mask[n] ? xfeat[(size_t)n * F + F - P + p] : o;
    }
}
