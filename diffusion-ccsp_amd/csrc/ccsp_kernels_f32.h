// ccsp_kernels_f32.h -- the fp32-MFMA evaluation kernels: MFMA core, k_rowgemm, k_edge (CCSP_MMA=f32 and the widths besides 256).
// A fragment of the ONE translation unit csrc/ccsp_hip.hip (included there, at this position, inside its namespaces): not a standalone header.
// ------------------------------------------------------------------------------------------
// MFMA core shared by k_ugemm and k_edge: one K chunk (BK) of a [32*WM] x [32*TN*WN] tile.
// LDS tiles are row-major [row][k] with stride LDS_LD; A rows = output rows, B rows = output cols.
// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5], B[k = l>>5][j = l&31].
// ------------------------------------------------------------------------------------------
template <int TN>
__device__ __forceinline__ void mfma_chunk(const float* __restrict__ As, const float* __restrict__ Bs, int a_row0,
                                           int b_row0, floatx16 (&acc)[TN]) {
    const int lane = threadIdx.x & 63;
    const float* ap = As + (a_row0 + (lane & 31)) * LDS_LD + (lane >> 5);
    const float* bp = Bs + (b_row0 + (lane & 31)) * LDS_LD + (lane >> 5);
    // fetch every fragment of the chunk first (BK/2 * (1 + TN) registers), then issue the MFMAs back
    // to back: the matrix pipe is not held up by LDS round trips between dependent k-steps
    float a[BK / 2], b[TN][BK / 2];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
        a[kk] = ap[2 * kk];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j][kk] = bp[j * 32 * LDS_LD + 2 * kk];
    }
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[j][kk], acc[j], 0, 0, 0);
}

// XCD-aware workgroup order (cdna_hip_programming.md T1): the dispatcher places block b on XCD b % 8
// and every XCD has a private 4 MiB L2.  Remapping block ids so that each XCD owns a contiguous range
// of tiles makes neighbouring tiles (same weight slice, same gathered rows) hit the same L2.
// Bijective for any grid size; a different placement would only change speed.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ void lds_store4(float* dst, const float4 v) {
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
}

// ------------------------------------------------------------------------------------------
// k_rowgemm<KD, ND>: out[row0+r, col0+c] = sum_k A[src(row0+r), k] * W[ts][col0+c, k]  (+ base + tau)
//   forward  (k_ugemm): KD = H,  ND = 2H, A = pose embeddings gathered by node, W = Wp[type, slot]
//   backward          : KD = 2H, ND = H,  A = row-summed g_z (identity rows),     W = Wp^T
//   work list = n_tiles * (ND / TILE_N) tiles, XCD-remapped, walked by a persistent grid; 4 waves as
//   2(M) x 2(N), each 32 x (TILE_N / 2).
//   `base` [R, ND] (chain-constant geometry/grasp term of the row) and `tau_t` [C, ND] (time term + bias,
//   slot-0 rows only) seed the accumulators, so an edge's pre-activation downstream is U[u0] + U[u1].
// ------------------------------------------------------------------------------------------
template <int ND> struct RowGemmCfg {
    static constexpr int TN_ = ND % 128 == 0 ? TILE_N : 64;   // column tile (64 wide when the width is not a multiple of 128: hidden_dim 64, 192, 320, 448)
    static constexpr int TNW = TN_ / 64;                  // 32-column MFMA tiles per wave
    static constexpr int BROWS = TN_ / 32;                // B staging rows per thread
    static constexpr int NCT = ND / TN_;                  // column tiles per row tile
};

// one 64 x TN_ output tile; `bid` = (row tile, column tile) work index
template <int KD, int ND>
__device__ __forceinline__ void rowgemm_tile(int bid, float (*As)[TILE_M * LDS_LD], float (*Bs)[RowGemmCfg<ND>::TN_ * LDS_LD],
                                             const float* __restrict__ A, const int* __restrict__ urow_node,
                                             const int* __restrict__ tile_row0, const int* __restrict__ tile_nrows,
                                             const int* __restrict__ tile_ts, const float* __restrict__ W,
                                             size_t w_stride, const float* __restrict__ base,
                                             const float* __restrict__ tau_t, float* __restrict__ U) {
    using Cfg = RowGemmCfg<ND>;
    constexpr int TN_ = Cfg::TN_, TNW = Cfg::TNW, BROWS = Cfg::BROWS, NCT = Cfg::NCT;
    const int tile = bid / NCT;
    const int row0 = tile_row0[tile], nrows = tile_nrows[tile], ts = tile_ts[tile];
    const int col0 = (bid % NCT) * TN_;
    const float* Wt = W + (size_t)ts * w_stride + (size_t)col0 * KD;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = tid >> 3, lq = tid & 7;
    const float* a_ptr[2];
    const float* b_ptr[BROWS];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int r = lr + 32 * i;
        r = r < nrows ? r : nrows - 1;
        const int src = urow_node ? urow_node[row0 + r] : row0 + r;
        a_ptr[i] = A + (size_t)src * KD + lq * 4;
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) b_ptr[i] = Wt + (size_t)(lr + 32 * i) * KD + lq * 4;
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
    // accumulators start from base (+ tau on slot-0 rows); these loads are in flight while the first K
    // chunk is staged.  C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    floatx16 acc[TNW];
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
        const int col = col0 + wn * 32 * TNW + j * 32 + (lane & 31);
        const float tv = (tau_t && (ts & 1) == 0) ? tau_t[(size_t)(ts >> 1) * ND + col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            row = row < nrows ? row : nrows - 1;
            acc[j][r] = (base ? base[(size_t)(row0 + row) * ND + col] : 0.0f) + tv;
        }
    }
    constexpr int NCH = KD / BK;
    for (int c = 0; c < NCH; ++c) {
        const int buf = c & 1;
        if (c + 1 < NCH) {
#pragma unroll
            for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const float4*>(a_ptr[i] + (c + 1) * BK);
#pragma unroll
            for (int i = 0; i < BROWS; ++i) rb[i] = *reinterpret_cast<const float4*>(b_ptr[i] + (c + 1) * BK);
        }
        // keep the prefetch ahead of the MFMA block: without this fence hipcc sinks the global loads to
        // just before their first use (the LDS stores below)
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk<TNW>(As[buf], Bs[buf], wm * 32, wn * 32 * TNW, acc);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < NCH) {
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
            if (row < nrows) U[(size_t)(row0 + row) * ND + col] = acc[j][r];
        }
}

// The grid may be smaller than the work list (persistent launch, stride gridDim.x): a workgroup's
// epilogue stores then drain under its own next tile.  In an isolated benchmark of this kernel that is
// worth -12 % (one-tile launches run the resident workgroups in lockstep into the epilogue); inside the
// chain it measured neutral, and a dynamic atomic tile queue was 2x slower, so the default launch is
// one tile per workgroup (see ccsp_model_create, CCSP_MAX_WGS).
template <int KD, int ND>
__global__ __launch_bounds__(256) void k_rowgemm(int n_work, const float* __restrict__ A, const int* __restrict__ urow_node,
                                                 const int* __restrict__ tile_row0, const int* __restrict__ tile_nrows,
                                                 const int* __restrict__ tile_ts, const float* __restrict__ W,
                                                 size_t w_stride, const float* __restrict__ base /*[R,ND] or null*/,
                                                 const float* __restrict__ tau_t /*[C,ND] or null*/, float* __restrict__ U) {
    __shared__ float As[2][TILE_M * LDS_LD];
    __shared__ float Bs[2][RowGemmCfg<ND>::TN_ * LDS_LD];
    for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
        rowgemm_tile<KD, ND>(xcd_remap(w, n_work), As, Bs, A, urow_node, tile_row0, tile_nrows, tile_ts, W, w_stride, base, tau_t, U);
        __syncthreads();
    }
}

template <int KD, int ND>
constexpr int rowgemm_col_tiles() { return ND / (ND % 128 == 0 ? TILE_N : 64); }

// base[r, :] = UG[r, :] (+ UR[r, :] on slot-0 rows: grasp_emb[args_1], denoise_fn.py:337) -- the
// chain-constant geometry/grasp part of row r's contribution to an edge pre-activation
__global__ void k_rowbase(int R, int W2, const int* __restrict__ urow_ts, const float* __restrict__ UR, float* __restrict__ UG) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)R * W2) return;
    const int r = (int)(idx / W2);
    if (UR && (urow_ts[r] & 1) == 0) UG[idx] += UR[idx];
}

// ------------------------------------------------------------------------------------------
// k_edge: rows = (sorted edge k, half s).  h = SiLU(U[u0(k)] + U[u1(k)])[s*H : (s+1)*H] is built
// chunk by chunk straight into the LDS A tile; B = pose_decoder.0 weight [H/2, H]; epilogue
// bias + SiLU -> LDS -> pose_decoder.2 (H/2 -> P) -> O[(2k+s)*P ..]   (denoise_fn.py:341-371)
//   H=256: 64 rows x 128 cols per workgroup (waves 2x2, 32x64 each)
//   H=64 : 128 rows x 32 cols per workgroup (waves 4x1, 32x32 each)
// grid = 2 * ceil(E_act / BM), XCD-remapped (a persistent-loop form of this kernel measured 1.4x slower)
// ------------------------------------------------------------------------------------------
#include "ccsp_energy_pre.h"

// sum_j a[j] w[j] over N (multiple of 4) as four independent chains: the serial fma chain of the naive loop,
// one exposed LDS/scalar-load round trip per element, was 8 us of k_edge's 39 (tools/abl_run.sh)
template <int N>
__device__ __forceinline__ float dot4(const float* __restrict__ a, const float* __restrict__ w) {
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f, o3 = 0.0f;
#pragma unroll 8
    for (int j = 0; j < N; j += 4) {
        o0 = fmaf(a[j], w[j], o0);
        o1 = fmaf(a[j + 1], w[j + 1], o1);
        o2 = fmaf(a[j + 2], w[j + 2], o2);
        o3 = fmaf(a[j + 3], w[j + 3], o3);
    }
    return (o0 + o1) + (o2 + o3);
}

// (any other multiple of 64: four row tiles, one wave column, H / 64 column tiles per wave)
template <int H> struct EdgeCfg { static constexpr int WM = 4, WN = 1, TN = H / 64; };
template <> struct EdgeCfg<256> { static constexpr int WM = 1, WN = 4, TN = 1; };
template <> struct EdgeCfg<128> { static constexpr int WM = 2, WN = 2, TN = 1; };
template <> struct EdgeCfg<64> { static constexpr int WM = 4, WN = 1, TN = 1; };

// Relay mode (EXPERIMENTS build, CCSP_RELAY=1; profiles/r05_findings.md section 5 -- slower than stream order): the three kernels of an evaluation are enqueued on three streams of their own and
// handed over through device counters instead of stream order, so that the launch boundary, the start-up of a kernel and everything it can
// load without its producer's results run UNDER the producer.  A workgroup polls `wait` until it has reached `target` (one lane, agent-scope
// acquire; then the workgroup's L1 / this XCD's L2 are invalidated like at a kernel start) and adds 1 to `done` once its own stores have been
// written back (agent-scope release, like a kernel end).  A wait that outlasts GATE_TIMEOUT (100 MHz ticks) raises *fault and goes on: a chain
// with a fault is reported as failed by the host, the GPU never hangs on a counter.
struct Gate {
    const unsigned int* wait;      // or null: no wait
    unsigned int target;
    unsigned int* done;            // or null: no signal
    unsigned int* fault;
};
constexpr long long GATE_TIMEOUT = 200000000LL;          // 2 s
#ifndef CCSP_EXPERIMENTS
__device__ __forceinline__ void gate_wait(const Gate&) {}
__device__ __forceinline__ void gate_done(const Gate&) {}
#else
__device__ __forceinline__ void gate_wait(const Gate& g) {
    if (g.wait == nullptr) return;
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        unsigned int spins = 0;
        while ((int)(__hip_atomic_load(g.wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - g.target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0 && wall_clock64() - t0 > GATE_TIMEOUT) { __hip_atomic_store(g.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
#ifndef CCSP_GATE_NOFENCE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}
__device__ __forceinline__ void gate_done(const Gate& g) {
    if (g.done == nullptr) return;
#ifndef CCSP_GATE_NOFENCE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // this thread's stores: acknowledged and written back
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(g.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

// ENERGY = true (denoise_fn.py:373-375): the CSR slot receives -2 d = -2 (o - pose) (the direct term of
// dE/dpose), the decoder pre-activations go to Q (when non-null, for k_edge_bwd) and the workgroup's
// share of sum d^2 to partial[blockIdx.x].
struct EdgeEnergyArgs {
    const int* e_a;          // node of slot 0 / slot 1 of every sorted edge
    const int* e_b;
    const float* xeval;      // [N, P] evaluation point
    float* Q;                // [2 E_act, H/2] or null
    float* partial;          // [gridDim.x]
    const int* skip;         // MALA reuse: if non-null and *skip == 0 the launch returns at once
    Gate gate;               // relay mode
};

template <int H, bool ENERGY>
__global__ __launch_bounds__(256, 3) void k_edge(int E_act, int P,
                                              const int* __restrict__ e_u0, const int* __restrict__ e_u1,
                                              const float* __restrict__ U, const float* __restrict__ Wd1 /*[H/2,H]*/,
                                              const float* __restrict__ bd1, const float* __restrict__ Wd2 /*[P,H/2]*/,
                                              const float* __restrict__ bd2, const int* __restrict__ ent_pos,
                                              float* __restrict__ O, EdgeEnergyArgs en) {
    using Cfg = EdgeCfg<H>;
    constexpr int BM = 32 * Cfg::WM, BN = 32 * Cfg::TN * Cfg::WN, TN = Cfg::TN;
    static_assert(BN == H / 2, "decoder hidden width must fit one column tile");
    constexpr int A_ROWS_PT = BM / 32, B_ROWS_PT = BN / 32;
    constexpr int STAGE = (BM + BN) * LDS_LD;            // floats per stage
    constexpr int S1_LD = BN + 1;
    constexpr int SMEM = (2 * STAGE > BM * S1_LD) ? 2 * STAGE : BM * S1_LD;
    __shared__ float smem[SMEM];
    auto As = [&](int buf) -> float* { return smem + buf * STAGE; };
    auto Bs = [&](int buf) -> float* { return smem + buf * STAGE + BM * LDS_LD; };
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int lr = tid >> 3, lq = tid & 7;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int e0 = (bid >> 1) * BM;
    const int s = bid & 1;
    const float* u0_ptr[A_ROWS_PT];
    const float* u1_ptr[A_ROWS_PT];
    const float* b_ptr[B_ROWS_PT];
#pragma unroll
    for (int i = 0; i < A_ROWS_PT; ++i) {
        int k = e0 + lr + 32 * i;
        k = k < E_act ? k : E_act - 1;
        const int coff = s * H + lq * 4;
        u0_ptr[i] = U + (size_t)e_u0[k] * (2 * H) + coff;
        u1_ptr[i] = U + (size_t)e_u1[k] * (2 * H) + coff;
    }
#pragma unroll
    for (int i = 0; i < B_ROWS_PT; ++i) b_ptr[i] = Wd1 + (size_t)(lr + 32 * i) * H + lq * 4;
    float4 ra[A_ROWS_PT], rb[B_ROWS_PT];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < A_ROWS_PT; ++i) {
            const float4 a = *reinterpret_cast<const float4*>(u0_ptr[i] + c * BK);
            const float4 b = *reinterpret_cast<const float4*>(u1_ptr[i] + c * BK);
            ra[i].x = silu_fast(a.x + b.x);
            ra[i].y = silu_fast(a.y + b.y);
            ra[i].z = silu_fast(a.z + b.z);
            ra[i].w = silu_fast(a.w + b.w);
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
    constexpr int NCH = H / BK;
    for (int c = 0; c < NCH; ++c) {
        const int buf = c & 1;
        if (c + 1 < NCH) load_chunk(c + 1);
        __builtin_amdgcn_sched_barrier(0);       // prefetch stays ahead of the MFMA block (see k_ugemm)
        mfma_chunk<TN>(As(buf), Bs(buf), wm * 32, wn * 32 * TN, acc);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < NCH) store_chunk(buf ^ 1);
        __syncthreads();
    }
    // epilogue 1: q = acc + bd1, s1 = SiLU(q) -> LDS [BM][BN+1]
    float* S1 = smem;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = wn * 32 * TN + j * 32 + (lane & 31);
        const float bj = bd1[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float q = acc[j][r] + bj;
            S1[row * S1_LD + col] = silu_fast(q);
            if constexpr (ENERGY) {
                const int k = e0 + row;
                if (en.Q && k < E_act) en.Q[((size_t)2 * k + s) * BN + col] = q;
            }
        }
    }
    __syncthreads();
    // epilogue 2: o[row, p] = bd2[p] + sum_j S1[row, j] Wd2[p, j]
    float e2 = 0.0f;
    for (int idx = tid; idx < BM * P; idx += 256) {
        const int row = idx % BM;
        const int p = (BM % 64 == 0) ? __builtin_amdgcn_readfirstlane(idx / BM) : idx / BM;   // uniform per wave when BM % 64 == 0: scalar weight loads
        float o = dot4<BN>(S1 + row * S1_LD, Wd2 + (size_t)p * BN) + bd2[p];
        const int k = e0 + row;
        if (k < E_act) {
            if constexpr (ENERGY) {
                const int node = s == 0 ? en.e_a[k] : en.e_b[k];
                const float d = o - en.xeval[(size_t)node * P + p];
                e2 = fmaf(d, d, e2);
                O[(size_t)ent_pos[2 * k + s] * P + p] = -2.0f * d;
            } else {
                O[(size_t)ent_pos[2 * k + s] * P + p] = o;             // straight to the node's CSR slot
            }
        }
    }
    if constexpr (ENERGY) {
        __syncthreads();                                               // S1 is dead: reuse it for the reduction
        const float tot = block_sum_256(e2, smem);
        if (tid == 0) en.partial[blockIdx.x] = tot;
    }
}

