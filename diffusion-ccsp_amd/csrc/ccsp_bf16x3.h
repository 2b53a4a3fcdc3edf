// fp32-accurate GEMMs on the bf16 matrix cores ("bf16x3"): every fp32 operand is split into three
// bf16 terms x = x1 + x2 + x3 (8 significand bits each, 24 together, fp32 exponent range), and a
// product a*b is accumulated in fp32 as the six cross terms of weight >= 2^-16:
//        a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1)
// Each bf16 x bf16 product is exact in fp32; the dropped terms are <= 2^-24 relative.  Measured on
// MI355X (tools/bf16x3_probe.hip, K = 256, data scaled 1e-6 .. 1e8): max |err| / sum|a b| = 2.2e-7 ..
// 2.8e-7, against 2.4e-7 .. 4.5e-7 for the fp32 MFMA chain -- the same accuracy class, so parity bars
// are unchanged.  v_mfma_f32_32x32x16_bf16 issues in 32 cycles for K = 16: six of them replace eight
// 64-cycle v_mfma_f32_32x32x2_f32, 2.67x less matrix-pipe time.
//
// Operands that are constant (weights) are split once at model creation; the pose embeddings are split
// by their producer (the encoder epilogue of k_node); the decoder input h = SiLU(U[u0] + U[u1]) is split in registers by the
// threads that build it.  Included inside the anonymous namespace of ccsp_hip.hip.
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short ushort8 __attribute__((ext_vector_type(8)));

// (bf16_rn_bits / bf16_bits_f / split3 live in ccsp_hip.hip: the node kernel's encoder writes planes too)

// dst[plane][i] = plane-th bf16 term of src[i]   (weights, once per model)
__global__ void k_split3(long n, const float* __restrict__ src, unsigned short* __restrict__ dst) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned short a, b, c;
    split3(src[i], a, b, c);
    dst[i] = a; dst[n + i] = b; dst[2 * n + i] = c;
}

constexpr int BF_BK = 32;                 // K chunk (two MFMA k-steps of 16)
constexpr int BF_LD = BF_BK + 8;          // bf16 elements per LDS row: 80 B, conflict-free ds_read_b128

// six-product MFMA block for one k-step: acc[j] += A(32 x 16) * B_j(16 x 32)
template <int TN>
__device__ __forceinline__ void mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[TN][3], floatx16 (&acc)[TN]) {
    // product-major, tile-minor: consecutive MFMAs go to different accumulators (same sums, same order per accumulator)
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]], b[j][PB[q]], acc[j], 0, 0, 0);
}

// one K chunk from LDS planes.  As/Bs: [3][rows][BF_LD] bf16; lane l reads row (l & 31), k 8*(l>>5)..+7
template <int TN, int A_ROWS, int B_ROWS>
__device__ __forceinline__ void bf_chunk(const unsigned short* __restrict__ As, const unsigned short* __restrict__ Bs,
                                         int a_row0, int b_row0, floatx16 (&acc)[TN]) {
    const int lane = threadIdx.x & 63;
    const int aoff = (a_row0 + (lane & 31)) * BF_LD + (lane >> 5) * 8;
    const int boff = (b_row0 + (lane & 31)) * BF_LD + (lane >> 5) * 8;
#pragma unroll
    for (int ks = 0; ks < BF_BK / 16; ++ks) {
        bf16x8 a[3], b[TN][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            a[p] = *reinterpret_cast<const bf16x8*>(As + p * A_ROWS * BF_LD + aoff + ks * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j][p] = *reinterpret_cast<const bf16x8*>(Bs + p * B_ROWS * BF_LD + boff + j * 32 * BF_LD + ks * 16);
        }
        mfma6<TN>(a, b, acc);
    }
}

// ------------------------------------------------------------------------------------------
// k_rowgemm_bf<KD, ND>: the row GEMM of k_rowgemm on the bf16 matrix cores.
//   A planes  [3][n_src_rows][KD] bf16 (pose embeddings split by k_node), rows gathered by urow_node
//   W planes  [3][n_ts][ND][KD]   bf16
//   64 x 128 tile, 4 waves 2(M) x 2(N), single LDS stage (45 KB -> 3 workgroups per CU) with the next
//   chunk prefetched into registers while the current one is multiplied.
// ------------------------------------------------------------------------------------------
template <int KD, int ND>
__global__ __launch_bounds__(256) void k_rowgemm_bf(const unsigned short* __restrict__ A, size_t a_plane,
                                                    const int* __restrict__ urow_node, const int* __restrict__ tile_row0,
                                                    const int* __restrict__ tile_nrows, const int* __restrict__ tile_ts,
                                                    const unsigned short* __restrict__ W, size_t w_plane, size_t w_stride,
                                                    const float* __restrict__ base, const float* __restrict__ tau_t,
                                                    float* __restrict__ U, StepRef ref, size_t tau_stride) {
    if (ref.tab) tau_t += (size_t)ref.tab[*ref.counter].t * tau_stride;      // hipGraph mode: timestep from the device table
    using Cfg = RowGemmCfg<ND>;
    constexpr int TN_ = Cfg::TN_, TNW = Cfg::TNW, NCT = Cfg::NCT;
    constexpr int A_UNITS = TILE_M * (BF_BK / 8) * 3 / 256;      // 16-byte units per thread (3)
    constexpr int B_UNITS = TN_ * (BF_BK / 8) * 3 / 256;         // (6 for 128 columns)
    __shared__ __attribute__((aligned(16))) unsigned short As[3 * TILE_M * BF_LD];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[3 * TN_ * BF_LD];
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = bid / NCT;
    const int row0 = tile_row0[tile], nrows = tile_nrows[tile], ts = tile_ts[tile];
    const int col0 = (bid % NCT) * TN_;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = tid >> 2, lq = tid & 3;                     // 64 rows x 4 sixteen-byte columns per pass
    // A: the same (row, column) of every plane; B: rows lrow + 64*(i & 1) (TN_ = 128) of plane i >> 1
    const unsigned short* a_ptr;
    {
        const int r = lrow < nrows ? lrow : nrows - 1;
        const int src = urow_node ? urow_node[row0 + r] : row0 + r;
        a_ptr = A + (size_t)src * KD + lq * 8;
    }
    const unsigned short* b_ptr[B_UNITS];
    int b_lds[B_UNITS];
#pragma unroll
    for (int i = 0; i < B_UNITS; ++i) {
        const int plane = i / (TN_ / 64), rr = lrow + 64 * (i % (TN_ / 64));
        b_ptr[i] = W + (size_t)plane * w_plane + (size_t)ts * w_stride + (size_t)(col0 + rr) * KD + lq * 8;
        b_lds[i] = plane * TN_ * BF_LD + rr * BF_LD + lq * 8;
    }
    ushort8 ra[A_UNITS], rb[B_UNITS];
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < A_UNITS; ++i) ra[i] = *reinterpret_cast<const ushort8*>(a_ptr + (size_t)i * a_plane + c * BF_BK);
#pragma unroll
        for (int i = 0; i < B_UNITS; ++i) rb[i] = *reinterpret_cast<const ushort8*>(b_ptr[i] + c * BF_BK);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < A_UNITS; ++i) *reinterpret_cast<ushort8*>(As + i * TILE_M * BF_LD + lrow * BF_LD + lq * 8) = ra[i];
#pragma unroll
        for (int i = 0; i < B_UNITS; ++i) *reinterpret_cast<ushort8*>(Bs + b_lds[i]) = rb[i];
    };
    gload(0);
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
    lstore();
    __syncthreads();
    constexpr int NCH = KD / BF_BK;
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) gload(c + 1);
        __builtin_amdgcn_sched_barrier(0);
        bf_chunk<TNW, TILE_M, TN_>(As, Bs, wm * 32, wn * 32 * TNW, acc);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                          // every wave is done reading the stage
        if (c + 1 < NCH) {
            lstore();
            __syncthreads();
        }
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

// ------------------------------------------------------------------------------------------
// k_rowgemm_bf2<KD, ND>: 128 x 128 tiles, 512 threads (8 waves as 4(M) x 2(N), 32 x 64 each).
// A 128-row tile reads each weight slice half as often as the 64 x 128 kernel above (253 instead of
// 380 MB of operand planes from L2 per launch at C2) and halves the number of workgroups, so that a
// half-batch launch of the two-lane chain is a single resident wave of workgroups.  LDS rows are
// unpadded 64-byte K chunks with the 16-byte piece index XOR-swizzled by (row >> 2) & 3: both the
// ds_read_b128 fragment reads (lane groups of MI355X_MICROARCH.md, LDS table) and the 8-lane
// ds_write_b128 groups are conflict-free, and the stage is 48 KB.
// ------------------------------------------------------------------------------------------
constexpr int RB2_TM = 128, RB2_TN = 128;

__device__ __forceinline__ int rb2_off(int row, int piece) { return row * BF_BK + ((piece ^ ((row >> 2) & 3)) << 3); }

template <int KD, int ND>
__global__ __launch_bounds__(512) void k_rowgemm_bf2(const unsigned short* __restrict__ A, size_t a_plane,
                                                     const int* __restrict__ urow_node, const int* __restrict__ tile_row0,
                                                     const int* __restrict__ tile_nrows, const int* __restrict__ tile_ts,
                                                     const unsigned short* __restrict__ W, size_t w_plane, size_t w_stride,
                                                     const float* __restrict__ base, const float* __restrict__ tau_t,
                                                     float* __restrict__ U, StepRef ref, size_t tau_stride) {
    static_assert(ND % RB2_TN == 0 && KD % BF_BK == 0, "shape");
    if (ref.tab) tau_t += (size_t)ref.tab[*ref.counter].t * tau_stride;      // hipGraph mode: timestep from the device table
    constexpr int NCT = ND / RB2_TN;
    constexpr int PLANE = RB2_TM * BF_BK;                         // ushorts per plane of a stage (A and B alike)
    __shared__ __attribute__((aligned(16))) unsigned short As[3 * PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[3 * PLANE];
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = bid / NCT;
    const int row0 = tile_row0[tile], nrows = tile_nrows[tile], ts = tile_ts[tile];
    const int col0 = (bid % NCT) * RB2_TN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = tid >> 2, lq = tid & 3;                      // one (row, 16-byte piece) of every plane, for A and for B
    const unsigned short* a_ptr;
    {
        const int r = lrow < nrows ? lrow : nrows - 1;
        const int src = urow_node ? urow_node[row0 + r] : row0 + r;
        a_ptr = A + (size_t)src * KD + lq * 8;
    }
    const unsigned short* b_ptr = W + (size_t)ts * w_stride + (size_t)(col0 + lrow) * KD + lq * 8;
    const int st_off = rb2_off(lrow, lq);
    ushort8 ra[3], rb[3];
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            ra[i] = *reinterpret_cast<const ushort8*>(a_ptr + (size_t)i * a_plane + c * BF_BK);
            rb[i] = *reinterpret_cast<const ushort8*>(b_ptr + (size_t)i * w_plane + c * BF_BK);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            *reinterpret_cast<ushort8*>(As + i * PLANE + st_off) = ra[i];
            *reinterpret_cast<ushort8*>(Bs + i * PLANE + st_off) = rb[i];
        }
    };
    gload(0);
    floatx16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = col0 + wn * 64 + j * 32 + (lane & 31);
        const float tv = (tau_t && (ts & 1) == 0) ? tau_t[(size_t)(ts >> 1) * ND + col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            row = row < nrows ? row : nrows - 1;
            acc[j][r] = (base ? base[(size_t)(row0 + row) * ND + col] : 0.0f) + tv;
        }
    }
    lstore();
    __syncthreads();
    const int arow = wm * 32 + (lane & 31), brow = wn * 64 + (lane & 31);
    constexpr int NCH = KD / BF_BK;
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) gload(c + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < BF_BK / 16; ++ks) {
            const int piece = (lane >> 5) + 2 * ks;
            bf16x8 a[3], b[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[p] = *reinterpret_cast<const bf16x8*>(As + p * PLANE + rb2_off(arow, piece));
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j][p] = *reinterpret_cast<const bf16x8*>(Bs + p * PLANE + rb2_off(brow + 32 * j, piece));
            }
            mfma6<2>(a, b, acc);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (c + 1 < NCH) {
            lstore();
            __syncthreads();
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int col = col0 + wn * 64 + j * 32 + (lane & 31);
            if (row < nrows) U[(size_t)(row0 + row) * ND + col] = acc[j][r];
        }
}

// ------------------------------------------------------------------------------------------
// k_edge_bf<H>: k_edge (direct mode) on the bf16 matrix cores.  h = SiLU(U[u0] + U[u1]) is built in
// fp32 and split in registers; B = pose_decoder.0 planes [3][H/2][H].
//   H=256: 64 rows x 128 cols per workgroup (waves 2x2, 32x64 each);  H=64: 128 rows x 32 cols (4x1)
// ------------------------------------------------------------------------------------------
template <int H> struct EdgeBfCfg { static constexpr int WM = 4, WN = 1, TN = H / 64; };      // (any other multiple of 64)
template <> struct EdgeBfCfg<256> { static constexpr int WM = 2, WN = 2, TN = 2; };
template <> struct EdgeBfCfg<128> { static constexpr int WM = 2, WN = 2, TN = 1; };
template <> struct EdgeBfCfg<64> { static constexpr int WM = 4, WN = 1, TN = 1; };

template <int H>
__global__ __launch_bounds__(256) void k_edge_bf(int E_act, int P, const int* __restrict__ e_u0,
                                                 const int* __restrict__ e_u1, const float* __restrict__ U,
                                                 const unsigned short* __restrict__ Wd1S /*[3][H/2][H]*/,
                                                 const float* __restrict__ bd1, const float* __restrict__ Wd2,
                                                 const float* __restrict__ bd2, const int* __restrict__ ent_pos,
                                                 float* __restrict__ O, int* __restrict__ counter_inc) {
    if (counter_inc && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(counter_inc, 1);     // hipGraph mode: next table entry
    using Cfg = EdgeBfCfg<H>;
    constexpr int BM = 32 * Cfg::WM, BN = 32 * Cfg::TN * Cfg::WN, TN = Cfg::TN;
    static_assert(BN == H / 2, "decoder hidden width must fit one column tile");
    constexpr int A_PASSES = BM / 32;             // thread handles rows (tid >> 3) + 32*i, 4 fp32 columns (tid & 7)*4
    constexpr int B_PASSES = (BN * (BF_BK / 8) + 255) / 256;       // row passes per plane (64 rows each)
    constexpr int B_UNITS = 3 * B_PASSES;
    constexpr int STAGE_US = 3 * (BM + BN) * BF_LD;                  // unsigned shorts
    constexpr int S1_LD = BN + 1;
    constexpr int SMEM_BYTES = (STAGE_US * 2 > BM * S1_LD * 4) ? STAGE_US * 2 : BM * S1_LD * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_BYTES];
    unsigned short* As = reinterpret_cast<unsigned short*>(smem_raw);
    unsigned short* Bs = As + 3 * BM * BF_LD;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int e0 = (bid >> 1) * BM;
    const int s = bid & 1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int lr = tid >> 3, lq = tid & 7;
    const float* u0_ptr[A_PASSES];
    const float* u1_ptr[A_PASSES];
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
        int k = e0 + lr + 32 * i;
        k = k < E_act ? k : E_act - 1;
        const int coff = s * H + lq * 4;
        u0_ptr[i] = U + (size_t)e_u0[k] * (2 * H) + coff;
        u1_ptr[i] = U + (size_t)e_u1[k] * (2 * H) + coff;
    }
    const int brow = tid >> 2, bq = tid & 3;
    const unsigned short* b_ptr[B_UNITS];
    int b_lds[B_UNITS];
#pragma unroll
    for (int i = 0; i < B_UNITS; ++i) {
        const int plane = i / B_PASSES, rr = brow + 64 * (i % B_PASSES);
        const int rrc = rr < BN ? rr : BN - 1;
        b_ptr[i] = Wd1S + (size_t)plane * (H / 2) * H + (size_t)rrc * H + bq * 8;
        b_lds[i] = rr < BN ? plane * BN * BF_LD + rr * BF_LD + bq * 8 : -1;
    }
    float4 ua[A_PASSES], ub[A_PASSES];
    ushort8 rb[B_UNITS];
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            ua[i] = *reinterpret_cast<const float4*>(u0_ptr[i] + c * BF_BK);
            ub[i] = *reinterpret_cast<const float4*>(u1_ptr[i] + c * BF_BK);
        }
#pragma unroll
        for (int i = 0; i < B_UNITS; ++i) rb[i] = *reinterpret_cast<const ushort8*>(b_ptr[i] + c * BF_BK);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const float h[4] = {silu_fast(ua[i].x + ub[i].x), silu_fast(ua[i].y + ub[i].y), silu_fast(ua[i].z + ub[i].z),
                                silu_fast(ua[i].w + ub[i].w)};
            unsigned short p1[4], p2[4], p3[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split3(h[e], p1[e], p2[e], p3[e]);
            unsigned short* d = As + (lr + 32 * i) * BF_LD + lq * 4;
            *reinterpret_cast<uint2*>(d) = make_uint2(p1[0] | ((unsigned)p1[1] << 16), p1[2] | ((unsigned)p1[3] << 16));
            *reinterpret_cast<uint2*>(d + BM * BF_LD) = make_uint2(p2[0] | ((unsigned)p2[1] << 16), p2[2] | ((unsigned)p2[3] << 16));
            *reinterpret_cast<uint2*>(d + 2 * BM * BF_LD) = make_uint2(p3[0] | ((unsigned)p3[1] << 16), p3[2] | ((unsigned)p3[3] << 16));
        }
#pragma unroll
        for (int i = 0; i < B_UNITS; ++i)
            if (b_lds[i] >= 0) *reinterpret_cast<ushort8*>(Bs + b_lds[i]) = rb[i];
    };
    gload(0);
    lstore();
    __syncthreads();
    floatx16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    constexpr int NCH = H / BF_BK;
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) gload(c + 1);
        __builtin_amdgcn_sched_barrier(0);
        bf_chunk<TN, BM, BN>(As, Bs, wm * 32, wn * 32 * TN, acc);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (c + 1 < NCH) {
            lstore();
            __syncthreads();
        }
    }
    // epilogue identical to k_edge: bias + SiLU -> LDS -> pose_decoder.2 -> CSR slot
    float* S1 = reinterpret_cast<float*>(smem_raw);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = wn * 32 * TN + j * 32 + (lane & 31);
        const float bj = bd1[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            S1[row * S1_LD + col] = silu_fast(acc[j][r] + bj);
        }
    }
    __syncthreads();
    for (int idx = tid; idx < BM * P; idx += 256) {
        const int row = idx % BM;
        const int p = (BM % 64 == 0) ? __builtin_amdgcn_readfirstlane(idx / BM) : idx / BM;
        const float o = dot4<BN>(S1 + row * S1_LD, Wd2 + (size_t)p * BN) + bd2[p];
        const int k = e0 + row;
        if (k < E_act) O[(size_t)ent_pos[2 * k + s] * P + p] = o;
    }
}

// ------------------------------------------------------------------------------------------
// k_edge_bf2: k_edge_bf for H = 256 with the activation work taken off the critical path.  In k_edge_bf
// a chunk is  wait(U rows) -> SiLU + split (VALU) -> LDS write -> barrier -> LDS read -> MFMA -> barrier,
// all in series; removing the MFMAs alone was worth 15 % of the whole chain (tools/ab.sh).  Here the A
// stage (the activations) is double-buffered, so the SiLU/split of chunk c+1 is issued between the MFMA
// groups of chunk c and runs in their shadow; only the plain copy of the next weight chunk sits between
// the two barriers.  Unpadded swizzled stages (rb2_off): 2 x 12 KB (A) + 24 KB (B) = 48 KB.
// 64 edges x 128 columns per workgroup, waves 2 x 2, one slot-half per workgroup like k_edge_bf.
// ------------------------------------------------------------------------------------------
template <bool ENERGY>
__global__ __launch_bounds__(256) void k_edge_bf2(int E_act, int P, const int* __restrict__ e_u0, const int* __restrict__ e_u1,
                                                  const float* __restrict__ U, const unsigned short* __restrict__ Wd1S /*[3][128][256]*/,
                                                  const float* __restrict__ bd1, const float* __restrict__ Wd2,
                                                  const float* __restrict__ bd2, const int* __restrict__ ent_pos, float* __restrict__ O,
                                                  EdgeEnergyArgs en, int* __restrict__ counter_inc) {
    if (counter_inc && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(counter_inc, 1);     // hipGraph mode: next table entry
    constexpr int H = 256, BM = 64, BN = 128;
    constexpr int APL = BM * BF_BK, BPL = BN * BF_BK;             // ushorts per plane
    constexpr int S1_LD = BN + 1;
    constexpr int STAGE_BYTES = (2 * 3 * APL + 3 * BPL) * 2;      // 49152
    static_assert(STAGE_BYTES >= BM * S1_LD * 4, "epilogue tile must fit the stage");
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[STAGE_BYTES];
    unsigned short* As = reinterpret_cast<unsigned short*>(smem_raw);         // [2][3][APL]
    unsigned short* Bs = As + 2 * 3 * APL;                                    // [3][BPL]
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int e0 = (bid >> 1) * BM;
    const int s = bid & 1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = tid >> 3, lq = tid & 7;                        // A producer: rows lr, lr + 32; fp32 columns lq*4 .. +3 of the chunk
    const float* u0_ptr[2];
    const float* u1_ptr[2];
    int a_st[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int k = e0 + lr + 32 * i;
        k = k < E_act ? k : E_act - 1;
        const int coff = s * H + lq * 4;
        u0_ptr[i] = U + (size_t)e_u0[k] * (2 * H) + coff;
        u1_ptr[i] = U + (size_t)e_u1[k] * (2 * H) + coff;
        a_st[i] = rb2_off(lr + 32 * i, lq >> 1) + (lq & 1) * 4;
    }
    const int brow = tid >> 2, bq = tid & 3;                      // B copy: rows brow, brow + 64, piece bq, every plane
    const unsigned short* b_ptr = Wd1S + (size_t)brow * H + bq * 8;
    const int b_st0 = rb2_off(brow, bq), b_st1 = rb2_off(brow + 64, bq);
    float4 ua[2][2], ub[2][2];                                    // [register set][pass]
    ushort8 rb[6];
    auto gload_a = [&](int c, int set) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ua[set][i] = *reinterpret_cast<const float4*>(u0_ptr[i] + c * BF_BK);
            ub[set][i] = *reinterpret_cast<const float4*>(u1_ptr[i] + c * BF_BK);
        }
    };
    auto gload_b = [&](int c) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            rb[2 * pl] = *reinterpret_cast<const ushort8*>(b_ptr + (size_t)pl * BN * H + c * BF_BK);
            rb[2 * pl + 1] = *reinterpret_cast<const ushort8*>(b_ptr + (size_t)pl * BN * H + (size_t)64 * H + c * BF_BK);
        }
    };
    auto store_a = [&](unsigned short* st, int set, int i) {     // SiLU + 3-way split of one pass -> the A stage
        const float h[4] = {silu_fast(ua[set][i].x + ub[set][i].x), silu_fast(ua[set][i].y + ub[set][i].y),
                            silu_fast(ua[set][i].z + ub[set][i].z), silu_fast(ua[set][i].w + ub[set][i].w)};
        unsigned short p1[4], p2[4], p3[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split3(h[e], p1[e], p2[e], p3[e]);
        unsigned short* d = st + a_st[i];
        *reinterpret_cast<uint2*>(d) = make_uint2(p1[0] | ((unsigned)p1[1] << 16), p1[2] | ((unsigned)p1[3] << 16));
        *reinterpret_cast<uint2*>(d + APL) = make_uint2(p2[0] | ((unsigned)p2[1] << 16), p2[2] | ((unsigned)p2[3] << 16));
        *reinterpret_cast<uint2*>(d + 2 * APL) = make_uint2(p3[0] | ((unsigned)p3[1] << 16), p3[2] | ((unsigned)p3[3] << 16));
    };
    auto store_b = [&]() {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            *reinterpret_cast<ushort8*>(Bs + pl * BPL + b_st0) = rb[2 * pl];
            *reinterpret_cast<ushort8*>(Bs + pl * BPL + b_st1) = rb[2 * pl + 1];
        }
    };
    constexpr int NCH = H / BF_BK;
    gload_a(0, 0);
    gload_b(0);
    gload_a(1, 1);
    floatx16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    store_a(As, 0, 0);
    store_a(As, 0, 1);
    store_b();
    gload_b(1);
    gload_a(2, 0);
    __syncthreads();
    const int arow = wm * 32 + (lane & 31), brw = wn * 64 + (lane & 31);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {                               // fully unrolled: the register-set index is a constant
        const unsigned short* st = As + (c & 1) * 3 * APL;
        unsigned short* nx = As + ((c + 1) & 1) * 3 * APL;
        const int set = (c + 1) & 1;                              // registers holding the U rows of chunk c+1
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int piece = (lane >> 5) + 2 * ks;
            bf16x8 a[3], b[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[p] = *reinterpret_cast<const bf16x8*>(st + p * APL + rb2_off(arow, piece));
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j][p] = *reinterpret_cast<const bf16x8*>(Bs + p * BPL + rb2_off(brw + 32 * j, piece));
            }
            mfma6<2>(a, b, acc);
            if (c + 1 < NCH) store_a(nx, set, ks);                // runs in the shadow of the 12 MFMAs just issued
        }
        if (c + 3 < NCH) gload_a(c + 3, set);                     // that register set is free again
        __syncthreads();                                          // every wave is done with Bs (and with stage c & 1)
        if (c + 1 < NCH) {
            store_b();
            if (c + 2 < NCH) gload_b(c + 2);
            __syncthreads();
        }
    }
    // epilogue identical to k_edge_bf: bias + SiLU -> LDS -> pose_decoder.2 -> CSR slot
    float* S1 = reinterpret_cast<float*>(smem_raw);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = wn * 64 + j * 32 + (lane & 31);
        const float bj = bd1[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float q = acc[j][r] + bj;
            S1[row * S1_LD + col] = silu_fast(q);
            if constexpr (ENERGY) {                               // decoder pre-activations for k_edge_bwd
                const int k = e0 + row;
                if (en.Q && k < E_act) en.Q[((size_t)2 * k + s) * BN + col] = q;
            }
        }
    }
    __syncthreads();
    float e2 = 0.0f;
    for (int idx = tid; idx < BM * P; idx += 256) {
        const int row = idx % BM;
        const int p = __builtin_amdgcn_readfirstlane(idx / BM);
        const float o = dot4<BN>(S1 + row * S1_LD, Wd2 + (size_t)p * BN) + bd2[p];
        const int k = e0 + row;
        if (k < E_act) {
            if constexpr (ENERGY) {                               // same epilogue as k_edge<H, true>
                const int node = s == 0 ? en.e_a[k] : en.e_b[k];
                const float d = o - en.xeval[(size_t)node * P + p];
                e2 = fmaf(d, d, e2);
                O[(size_t)ent_pos[2 * k + s] * P + p] = -2.0f * d;
            } else {
                O[(size_t)ent_pos[2 * k + s] * P + p] = o;
            }
        }
    }
    if constexpr (ENERGY) {
        __syncthreads();
        const float tot = block_sum_256(e2, reinterpret_cast<float*>(smem_raw));
        if (tid == 0) en.partial[blockIdx.x] = tot;
    }
}

// ------------------------------------------------------------------------------------------
// k_edge_bwd_bf: k_edge_bwd (ccsp_energy.h) for H = 256 on the bf16 matrix pipe, with k_edge_bf2's pipeline.
// Rows = (sorted edge, slot s); K = H/2 = 128 decoder hidden units; N = 128 of the H columns per workgroup.
// A[row, j] = (sum_p go[p] Wd2[p, j]) * SiLU'(q[row, j]) is built on the VALU for chunk c+1 between the MFMA
// groups of chunk c and split into planes (double-buffered stage); B = planes of Wd1^T [H, H/2].
// Epilogue: GZ[k, s H + n] = acc * SiLU'(U[u0] + U[u1])[s H + n].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_edge_bwd_bf(int E_act, int P, const int* __restrict__ e_u0, const int* __restrict__ e_u1,
                                                     const int* __restrict__ ent_pos, const float* __restrict__ U,
                                                     const float* __restrict__ Ocsr, const float* __restrict__ Q /*[2E,128]*/,
                                                     const unsigned short* __restrict__ Wd1TS /*[3][256][128]*/,
                                                     const float* __restrict__ Wd2 /*[P,128]*/, float* __restrict__ GZ) {
    constexpr int H = 256, KD = 128, BM = 64, BN = 128;
    constexpr int APL = BM * BF_BK, BPL = BN * BF_BK;
    __shared__ __attribute__((aligned(16))) unsigned short smem_us[2 * 3 * APL + 3 * BPL];      // 48 KB: A stages, then the B stage
    unsigned short* As = smem_us;
    unsigned short* Bs = smem_us + 2 * 3 * APL;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = bid & 1, s = (bid >> 1) & 1, e0 = (bid >> 2) * BM;
    const int n0 = ct * BN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = tid >> 3, lq = tid & 7;
    float go[2][8];
    const float* q_ptr[2];
    int a_st[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int k = e0 + lr + 32 * i;
        k = k < E_act ? k : E_act - 1;
        const size_t row = (size_t)2 * k + s;
        const float* o = Ocsr + (size_t)ent_pos[row] * P;
#pragma unroll
        for (int p = 0; p < 8; ++p) go[i][p] = p < P ? -o[p] : 0.0f;       // 2 d = -(-2 d)
        q_ptr[i] = Q + row * KD + lq * 4;
        a_st[i] = rb2_off(lr + 32 * i, lq >> 1) + (lq & 1) * 4;
    }
    const int brow = tid >> 2, bq = tid & 3;
    const unsigned short* b_ptr = Wd1TS + (size_t)(n0 + brow) * KD + bq * 8;
    const int b_st0 = rb2_off(brow, bq), b_st1 = rb2_off(brow + 64, bq);
    float4 rq[2][2];                                              // [register set][pass]: decoder pre-activations
    ushort8 rb[6];
    auto gload_a = [&](int c, int set) {
#pragma unroll
        for (int i = 0; i < 2; ++i) rq[set][i] = *reinterpret_cast<const float4*>(q_ptr[i] + c * BF_BK);
    };
    auto gload_b = [&](int c) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            rb[2 * pl] = *reinterpret_cast<const ushort8*>(b_ptr + (size_t)pl * H * KD + c * BF_BK);
            rb[2 * pl + 1] = *reinterpret_cast<const ushort8*>(b_ptr + (size_t)pl * H * KD + (size_t)64 * KD + c * BF_BK);
        }
    };
    auto store_a = [&](unsigned short* st, int c, int set, int i) {
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if (p < P) {
                const float4 w2 = *reinterpret_cast<const float4*>(Wd2 + (size_t)p * KD + c * BF_BK + lq * 4);
                g.x = fmaf(go[i][p], w2.x, g.x); g.y = fmaf(go[i][p], w2.y, g.y);
                g.z = fmaf(go[i][p], w2.z, g.z); g.w = fmaf(go[i][p], w2.w, g.w);
            }
        }
        const float4 q = rq[set][i];
        const float h[4] = {g.x * silu_grad_fast(q.x), g.y * silu_grad_fast(q.y), g.z * silu_grad_fast(q.z), g.w * silu_grad_fast(q.w)};
        unsigned short p1[4], p2[4], p3[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split3(h[e], p1[e], p2[e], p3[e]);
        unsigned short* d = st + a_st[i];
        *reinterpret_cast<uint2*>(d) = make_uint2(p1[0] | ((unsigned)p1[1] << 16), p1[2] | ((unsigned)p1[3] << 16));
        *reinterpret_cast<uint2*>(d + APL) = make_uint2(p2[0] | ((unsigned)p2[1] << 16), p2[2] | ((unsigned)p2[3] << 16));
        *reinterpret_cast<uint2*>(d + 2 * APL) = make_uint2(p3[0] | ((unsigned)p3[1] << 16), p3[2] | ((unsigned)p3[3] << 16));
    };
    auto store_b = [&]() {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            *reinterpret_cast<ushort8*>(Bs + pl * BPL + b_st0) = rb[2 * pl];
            *reinterpret_cast<ushort8*>(Bs + pl * BPL + b_st1) = rb[2 * pl + 1];
        }
    };
    constexpr int NCH = KD / BF_BK;                               // 4
    gload_a(0, 0);
    gload_b(0);
    gload_a(1, 1);
    floatx16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    store_a(As, 0, 0, 0);
    store_a(As, 0, 0, 1);
    store_b();
    gload_b(1);
    gload_a(2, 0);
    __syncthreads();
    const int arow = wm * 32 + (lane & 31), brw = wn * 64 + (lane & 31);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const unsigned short* st = As + (c & 1) * 3 * APL;
        unsigned short* nx = As + ((c + 1) & 1) * 3 * APL;
        const int set = (c + 1) & 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int piece = (lane >> 5) + 2 * ks;
            bf16x8 a[3], b[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[p] = *reinterpret_cast<const bf16x8*>(st + p * APL + rb2_off(arow, piece));
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j][p] = *reinterpret_cast<const bf16x8*>(Bs + p * BPL + rb2_off(brw + 32 * j, piece));
            }
            mfma6<2>(a, b, acc);
            if (c + 1 < NCH) store_a(nx, c + 1, set, ks);
        }
        if (c + 3 < NCH) gload_a(c + 3, set);
        __syncthreads();
        if (c + 1 < NCH) {
            store_b();
            if (c + 2 < NCH) gload_b(c + 2);
            __syncthreads();
        }
    }
    // epilogue through LDS: the accumulators (MFMA layout: one column, 16 rows per lane) are re-read as rows of
    // float4, so the U gathers and the GZ stores are 128-byte row segments instead of 4-byte scattered accesses
    constexpr int C_LD = BN + 4;
    static_assert((2 * 3 * APL + 3 * BPL) * 2 >= BM * C_LD * 4, "epilogue tile must fit the stages");
    float* Cs = reinterpret_cast<float*>(As);                     // 64 x 132 floats = 33 KB <= the two stages (As 24 KB + Bs 24 KB are contiguous)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            Cs[row * C_LD + wn * 64 + j * 32 + (lane & 31)] = acc[j][r];
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = lr + 32 * i;
        const int k = e0 + row;
        if (k >= E_act) continue;
        const float* u0 = U + (size_t)e_u0[k] * (2 * H) + s * H + n0;
        const float* u1 = U + (size_t)e_u1[k] * (2 * H) + s * H + n0;
        float* gz = GZ + (size_t)k * (2 * H) + s * H + n0;
#pragma unroll
        for (int mcol = 0; mcol < 4; ++mcol) {
            const int c = lq * 4 + 32 * mcol;
            const float4 a = *reinterpret_cast<const float4*>(u0 + c);
            const float4 b = *reinterpret_cast<const float4*>(u1 + c);
            const float4 v = *reinterpret_cast<const float4*>(Cs + row * C_LD + c);
            *reinterpret_cast<float4*>(gz + c) = make_float4(v.x * silu_grad_fast(a.x + b.x), v.y * silu_grad_fast(a.y + b.y),
                                                             v.z * silu_grad_fast(a.z + b.z), v.w * silu_grad_fast(a.w + b.w));
        }
    }
}
