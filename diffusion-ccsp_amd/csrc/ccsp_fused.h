// k_eval_fused: one evaluation of the per-type edge MLP + pose decoder (networks/denoise_fn.py:313-371) as ONE launch in which the
// pre-activation rows U never leave the compute unit (round 4).  The two-kernel form (k_rowgemm_h2 -> U in memory -> k_edge_h2)
// moves 123 of its 159 MB per C2 evaluation writing U and gathering it back; here a workgroup owns a FUSED TILE
//     (constraint type t, a run of that type's sorted edges whose distinct U rows fit 32 per slot, output half h)
// and does, without leaving the CU:
//   phase 1  U_h[r, :] = 2^-(e_r + e_w) pemb[node(r)] . Wp[t, slot(r)][h H : (h+1) H, :]^T + base[r, h H : (h+1) H] (+ tau, slot 0)
//            for the tile's <= 64 rows (LDS rows 0..31 = slot 0, 32..63 = slot 1) into a 64 x 256 fp32 LDS tile.  The pose-embedding
//            planes of the 64 rows are staged once (LDS-DMA, 64 KB); the weight fragments come STRAIGHT from global memory in
//            v_mfma_f32_32x32x16_f16 operand order (k_pack_wp_frag: every wave-load is 1 KB contiguous, no LDS staging, no
//            barrier in the K loop), four k-steps ahead with counted waits; `base` is DMA-ed into the U tile and the
//            accumulators are added in place.
//   phase 2  the tile's <= 128 edges: A[e, :] = split(2^x SiLU(U_h[lu0(e)] + U_h[lu1(e)])) built from the LDS tile (16 lanes per row:
//            conflict-free b128 gathers whatever the rows), decoder layer 1 on the f16 pipe with fragment-ordered weights from
//            global memory, bias + SiLU, layer 2 on the VALU, outputs to the same CSR slots as k_edge_h2.
// Same operands, same exponents (row maxima of U_h taken from the tile), same MFMA order per accumulator and the same epilogue
// arithmetic as k_rowgemm_h2 + k_edge_h2<false, 1, 0>: the edge outputs are BITWISE those of the two-kernel path, so the node
// kernel, the summation order and every parity bar are untouched (tests/test_hip_parity.py::test_fused_eval_is_bitwise_identical).
// What it costs: a tile streams both slots' weight halves (512 KB of planes) for at most 64 rows -- 2.25x the operand bytes of a
// 128 x 128 row-GEMM tile per row -- through the CU's L2 port; what it saves: U, umax and one launch boundary.
// Round 4, second form: the launch is PERSISTENT -- one 512-thread workgroup per compute unit (the LDS of a tile admits one) walks a
// cost-sorted list of (tile, half) items with stride gridDim -- and software-pipelined across items: nothing else is resident on
// the CU to cover a tile's dependent prologue (descriptor -> row indices -> pose-embedding planes: 7 k of the first form's 32 k
// cycles per tile, phase trace in profiles/r04_findings.md), so the NEXT item's indices are requested under the decoder's K loop and
// its embedding planes and first weight fragments under the decoder's epilogue; the decoder's own weight fragments (32 KB per wave) are
// requested two 64-wide stages ahead through a ring of three register buffers (the first two are the then idle phase-1 ring).
// Included inside the anonymous namespace of ccsp_hip.hip, after ccsp_f16x2.h.
#pragma once

constexpr int FZ_RS = 32;                     // U rows per slot of a tile
constexpr int FZ_ME = 128;                    // edges per tile (four 32-row MFMA tiles)
constexpr int FZ_D = 4;                       // k-steps of weight fragments in flight per wave (phase 1)
constexpr int FZ_APL = 64 * H2_BK;            // fp16 elements per plane of one K chunk of the staged A rows
constexpr int FZ_APL2 = FZ_ME * H2_BK;        // ... of one 32-wide sub-stage of the decoder's A operand
constexpr int FZ_W2_LD = 132;                 // row stride of the staged second-layer weight: four rows read by one b128 group hit different banks
constexpr int FZ_UT_BYTES = 64 * 256 * 4;     // the U tile [64][256] fp32; after the decoder's K loop the S1 tile [128][128] (XOR-swizzled quads)
constexpr int FZ_R2_BYTES = 64 * 256 * 2 * 2; // A planes of phase 1 [8 chunks][2 planes][64][32]; the two A stages of phase 2
constexpr int FZ_LDS_BYTES = FZ_UT_BYTES + FZ_R2_BYTES + 8 * FZ_W2_LD * 4 + 8 * 4 + 64 * 4 + 64 * 4 + FZ_ME * 4;

struct FusedArgs {
    const int* order;                 // [n_items] work list: 2 tile + half, most expensive first
    int n_items;
    const int4* tiles;                // {type, e0, ne, -}
    const int* rows;                  // [n_tiles][128]: [0..63] node (pose-embedding row) of LDS row i, [64..127] U row (row of base)
    const unsigned short* e_lu;       // [E_act] LDS rows of the edge's two operands: lu0 | lu1 << 8
    const int* ent_pos;               // [2 E_act] CSR slot of (sorted edge, slot)
    const unsigned short* A;          // pose-embedding planes [2][N][256]
    size_t a_plane;
    const int* a_exp;                 // [N]
    const unsigned short* WpF;        // fragment-ordered planes of Wp (k_pack_wp_frag)
    int w_exp;
    const float* base;                // [R][512]
    const float* tau_t;               // [C][512] time term + bias of this timestep
    const unsigned short* Wd1F;       // fragment-ordered planes of pose_decoder.0.weight (k_pack_wd1_frag)
    int wd_exp;
    const float* bd1;
    const float* Wd2;                 // [P][128]
    const float* bd2;
    float* O;
    int P;
};

// WpH [2][n_ts][512][256] -> WpF[((((ts * 2 + h) * 4 + wq) * 16 + ks) * 4 + f)][lane][8],  f = 2 j + plane:
// the B fragment of v_mfma_f32_32x32x16_f16 for output columns h 256 + wq 64 + j 32 + (lane & 31), k = 16 ks + 8 (lane >> 5) + 0..7
__global__ void k_pack_wp_frag(long n16, const unsigned short* __restrict__ WpH, size_t plane_stride, unsigned short* __restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n16) return;
    const int lane = (int)(idx & 63), f = (int)(idx >> 6) & 3, ks = (int)(idx >> 8) & 15, wq = (int)(idx >> 12) & 3, h = (int)(idx >> 14) & 1;
    const long ts = idx >> 15;
    const int col = h * 256 + wq * 64 + (f >> 1) * 32 + (lane & 31), k0 = ks * 16 + 8 * (lane >> 5);
    const uint4 v = *reinterpret_cast<const uint4*>(WpH + (size_t)(f & 1) * plane_stride + ((size_t)ts * 512 + col) * 256 + k0);
    *reinterpret_cast<uint4*>(out + (size_t)idx * 8) = v;
}

// Wd1H [2][128][256] -> Wd1F[((nt * 16 + ks) * 2 + plane)][lane][8]: columns nt 32 + (lane & 31), k = 16 ks + 8 (lane >> 5) + 0..7
__global__ void k_pack_wd1_frag(const unsigned short* __restrict__ Wd1H, unsigned short* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 4 * 16 * 2 * 64) return;
    const int lane = idx & 63, p = (idx >> 6) & 1, ks = (idx >> 7) & 15, nt = idx >> 11;
    const int col = nt * 32 + (lane & 31), k0 = ks * 16 + 8 * (lane >> 5);
    const uint4 v = *reinterpret_cast<const uint4*>(Wd1H + (size_t)p * 128 * 256 + (size_t)col * 256 + k0);
    *reinterpret_cast<uint4*>(out + (size_t)idx * 8) = v;
}

// fragment loads the compiler must neither move nor wait for (see h2_ld16): byte offsets as instruction immediates
template <int OFF>
__device__ __forceinline__ void fz_ld_frag(half8& d, const unsigned short* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(d) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ void fz_ld_f32(float& d, const float* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
// s_waitcnt vmcnt(n), n known after unrolling, naming the four fragments it releases
__device__ __forceinline__ void fz_wait4(int n, half8 (&b)[4]) {
#define FZ_W(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) :: "memory")
    if (n >= 16) FZ_W(16); else if (n >= 12) FZ_W(12); else if (n >= 8) FZ_W(8); else if (n >= 4) FZ_W(4); else FZ_W(0);
#undef FZ_W
}
__device__ __forceinline__ void fz_wait8(half8 (&b)[8]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) :: "memory");
}
// maximum over the 16 lanes of a DPP row (h2_max8 + row_mirror)
__device__ __forceinline__ float fz_max16(float m) {
    m = h2_max8(m);
    return fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x140, 0xF, 0xF, true)));
}

// everything of one (tile, half) item that is read from the index tables: requested one item ahead
struct FzItem {
    int type, e0, ne, h;
    int node;                         // pose-embedding row this lane fetches (A planes)
    int ur[8];                        // U rows (rows of base) of the eight tile rows this wave fetches (wave-uniform)
    unsigned int lu[2];               // tile rows of the two operands of the edges this thread builds: pass i (edges 32 i + tid / 16) in
                                      // bits 16 (i & 1) .. + 15 of lu[i >> 1] as row 0 | row 1 << 8 (kept packed: the NEXT item's copy is live
                                      // through the whole decoder loop)
    int o_slot;                       // CSR slot of the output row this thread stores
};

__device__ __forceinline__ void fz_load_item(const FusedArgs& fa, int tid, int wave, int item, FzItem& it) {
    const int lane = tid & 63;
    const int w = fa.order[item];
    const int tile = w >> 1;
    const int4 td = fa.tiles[tile];
    it.type = td.x; it.e0 = td.y; it.ne = td.z; it.h = w & 1;
    const int* trow = fa.rows + (size_t)tile * 128;
    it.node = trow[(wave & 3) * 16 + (lane >> 2)];
    const int urow_l = trow[64 + 8 * wave + (lane & 7)];
#pragma unroll
    for (int j = 0; j < 8; ++j) it.ur[j] = __builtin_amdgcn_readlane(urow_l, j);
    const int br = tid >> 4;
    it.lu[0] = 0u; it.lu[1] = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int el = 32 * i + br;
        el = el < it.ne ? el : it.ne - 1;
        it.lu[i >> 1] |= (unsigned int)fa.e_lu[it.e0 + el] << (16 * (i & 1));
    }
    int row = tid >> 2;
    row = row < it.ne ? row : it.ne - 1;
    it.o_slot = fa.ent_pos[2 * (it.e0 + row) + it.h];
}

// S1 tile [128][128] fp32 in the U tile's LDS: the 16-byte quad q of row r lives at quad q ^ (r & 31) -- rows read one per lane
// (second layer) and 32 consecutive columns of a row written per half-wave (accumulator layout) are both conflict-free
__device__ __forceinline__ int fz_s1_off(int row, int col) { return row * 128 + ((((col >> 2) ^ (row & 31))) << 2) + (col & 3); }

// decoder of the tile's edges (phase 2).  MPW: 32-row MFMA tiles per wave (1: tile wave >> 2 of <= 2; 2: tiles 2 (wave >> 2) + 0 / 1
// of <= 4); every wave takes 32 of the 128 decoder columns.  The wave's weight fragments (8 per 64-wide stage) go through a ring of three
// register buffers, two stages ahead: bq[0..7] / bq[8..15] hold stages 0 / 1, requested by the caller under the U epilogue (16 loads
// that may still be in flight); stage 2 is requested here into b3, stage 3 into bq[0..7] once stage 0 has been multiplied.
template <int MPW>
__device__ __forceinline__ void fz_decode_loop(int tid, int wave, const unsigned short* wd1f_wave, unsigned char* smem, int ne, const int (&lu0)[4],
                                               const int (&lu1)[4], const int (&aexp)[4], half8 (&bq)[16], floatx16 (&acc)[2]) {
    const int lane = tid & 63;
    const int mh = wave >> 2;
    const int MT = (ne + 31) >> 5;
    const float* Ut = reinterpret_cast<const float*>(smem);
    unsigned short* stg = reinterpret_cast<unsigned short*>(smem + FZ_UT_BYTES);        // two stages of [sub 2][plane 2][128][32]
    constexpr int STAGE = 4 * FZ_APL2;
    const int br = tid >> 4, lq16 = tid & 15, sub = lq16 >> 3, lq = lq16 & 7;
    // A rows of the 64-column stage cp, pass i (edges 32 i + br): SiLU + scale + split of four columns -> both planes
    auto build = [&](int cp, int i, const float4& ua, const float4& ub) {
        unsigned short* As = stg + (cp & 1) * STAGE + sub * 2 * FZ_APL2;
        const float hv[4] = {silu_fast(ua.x + ub.x), silu_fast(ua.y + ub.y), silu_fast(ua.z + ub.z), silu_fast(ua.w + ub.w)};
        unsigned short p1[4], p2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split2h(ldexpf(hv[e], aexp[i]), p1[e], p2[e]);
        unsigned short* d = As + h2_off(32 * i + br, lq >> 1) + (lq & 1) * 4;
        *reinterpret_cast<uint2*>(d) = make_uint2(p1[0] | ((unsigned)p1[1] << 16), p1[2] | ((unsigned)p1[3] << 16));
        *reinterpret_cast<uint2*>(d + FZ_APL2) = make_uint2(p2[0] | ((unsigned)p2[1] << 16), p2[2] | ((unsigned)p2[3] << 16));
    };
    auto gather = [&](int cp, int i, float4& ua, float4& ub) {
        ua = *reinterpret_cast<const float4*>(Ut + lu0[i] * 256 + cp * 64 + lq16 * 4);
        ub = *reinterpret_cast<const float4*>(Ut + lu1[i] * 256 + cp * 64 + lq16 * 4);
    };
    auto mma = [&](int cp, int ks, const half8* b /*8 fragments of the stage: [k-step][plane]*/) {      // same product order as h2_kstep
        const unsigned short* As = stg + (cp & 1) * STAGE + (ks >> 1) * 2 * FZ_APL2;
        const int piece = (lane >> 5) + 2 * (ks & 1);
        half8 a[MPW][2];
#pragma unroll
        for (int i = 0; i < MPW; ++i) {
            const int mt = MPW == 1 ? mh : 2 * mh + i;
#pragma unroll
            for (int p = 0; p < 2; ++p) a[i][p] = *reinterpret_cast<const half8*>(As + p * FZ_APL2 + h2_off(mt * 32 + (lane & 31), piece));
        }
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < MPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][PA[q]], b[2 * ks + PB[q]], acc[i], 0, 0, 0);
    };
    const bool wave_has_tile = (MPW == 1 ? mh : 2 * mh) < MT;     // (wave-uniform)
    half8 b3[8];
    auto ldW = [&](int cp, half8* b) {                            // the wave's eight fragments of stage cp: 8 KB contiguous
        const unsigned short* p = wd1f_wave + (size_t)cp * 4 * 2 * 512;
        fz_ld_frag<0>(b[0], p); fz_ld_frag<1024>(b[1], p); fz_ld_frag<2048>(b[2], p); fz_ld_frag<3072>(b[3], p);
        fz_ld_frag<0>(b[4], p + 2048); fz_ld_frag<1024>(b[5], p + 2048); fz_ld_frag<2048>(b[6], p + 2048); fz_ld_frag<3072>(b[7], p + 2048);
    };
    ldW(2, b3);
    __builtin_amdgcn_sched_barrier(0);
    {   // stage 0
        float4 ua, ub;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < MT) { gather(0, i, ua, ub); build(0, i, ua, ub); }
    }
    // stage 0's fragments have landed (younger: stage 1's eight, the caller's index loads of the next item, stage 2's eight)
    asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]), "+v"(bq[4]), "+v"(bq[5]), "+v"(bq[6]), "+v"(bq[7]) :: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    CCSP_TRK(0, 10);
#pragma unroll
    for (int cp = 0; cp < 4; ++cp) {
        half8* bw = cp == 1 ? bq + 8 : (cp == 2 ? b3 : bq);
        if (cp == 1) { ldW(3, bq); __builtin_amdgcn_sched_barrier(0); }      // (stage 0's MFMAs have been issued: its buffer is refilled)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float4 ua, ub;
            const bool bld = cp + 1 < 4 && ks < MT;               // pass ks of the next stage rides behind this k-step's MFMAs
            if (bld) gather(cp + 1, ks, ua, ub);
            if (wave_has_tile) mma(cp, ks, bw);
            if (bld) build(cp + 1, ks, ua, ub);
        }
        __builtin_amdgcn_sched_barrier(0);
        // the next stage's fragments have landed; younger at this point: after stage 0 the eight of stage 2, after stage 1 the eight of stage 3
        if (cp == 0) asm volatile("s_waitcnt vmcnt(8)" : "+v"(bq[8]), "+v"(bq[9]), "+v"(bq[10]), "+v"(bq[11]), "+v"(bq[12]), "+v"(bq[13]), "+v"(bq[14]), "+v"(bq[15]) :: "memory");
        if (cp == 1) asm volatile("s_waitcnt vmcnt(8)" : "+v"(b3[0]), "+v"(b3[1]), "+v"(b3[2]), "+v"(b3[3]), "+v"(b3[4]), "+v"(b3[5]), "+v"(b3[6]), "+v"(b3[7]) :: "memory");
        if (cp == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]), "+v"(bq[4]), "+v"(bq[5]), "+v"(bq[6]), "+v"(bq[7]) :: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ __launch_bounds__(512, 2) void k_eval_fused(FusedArgs fa) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[FZ_LDS_BYTES];
    using gptr = const __attribute__((address_space(1))) void*;
    using lptr = __attribute__((address_space(3))) void*;
    float* Ut = reinterpret_cast<float*>(smem);                                           // [64][256] fp32; S1 after the decoder's K loop
    unsigned short* As = reinterpret_cast<unsigned short*>(smem + FZ_UT_BYTES);           // phase 1: [8 chunks][2 planes][64][32]
    float* W2s = reinterpret_cast<float*>(smem + FZ_UT_BYTES + FZ_R2_BYTES);              // [8][FZ_W2_LD]
    float* sB2 = W2s + 8 * FZ_W2_LD;                                                      // [8] second-layer bias
    int* sExpA = reinterpret_cast<int*>(sB2 + 8);                                         // [64] exponents of the pose-embedding rows
    float* sMax = reinterpret_cast<float*>(sExpA + 64);                                   // [64] max |U_h| per row
    int* sE = reinterpret_cast<int*>(sMax + 64);                                          // [128] exponents of the decoder's A rows
    const int tid0 = threadIdx.x;
    const int P = fa.P;
    CCSP_TRK_RT(0, 30);
    // --- once per launch: second-layer weight and biases
    const float bj = fa.bd1[((tid0 >> 6) & 3) * 32 + (tid0 & 31)];
    for (int i = tid0; i < 8 * 128; i += 512) W2s[(i >> 7) * FZ_W2_LD + (i & 127)] = i < P * 128 ? fa.Wd2[i < P * 128 ? i : 0] : 0.0f;
    if (tid0 < 8) sB2[tid0] = tid0 < P ? fa.bd2[tid0 < P ? tid0 : 0] : 0.0f;
    half8 bq[16];                                                  // phase 1: ring of FZ_D k-steps x 4 fragments; phase 2: the first half of the decoder weight
    auto ldB = [&](int tid, const FzItem& it, int ks, half8* b) {
        const int wave = tid >> 6, lane = tid & 63;
        const unsigned short* p = fa.WpF + ((((size_t)(2 * it.type + (wave >> 2)) * 2 + it.h) * 4 + (wave & 3)) * 16 + ks) * 4 * 512 + lane * 8;
        fz_ld_frag<0>(b[0], p); fz_ld_frag<1024>(b[1], p); fz_ld_frag<2048>(b[2], p); fz_ld_frag<3072>(b[3], p);
    };
    auto dmaA = [&](int tid, const FzItem& it) {                   // A planes of the 64 rows (LDS-DMA, source-side swizzle as k_rowgemm_h2 MODE 2)
        const int wave = tid >> 6, lane = tid & 63;
        const int a_row = (wave & 3) * 16 + (lane >> 2);
        const int plane = wave >> 2, rb = wave & 3;
        const int piece = (lane & 3) ^ ((a_row >> 2) & 3);
        const unsigned short* ga = fa.A + (size_t)plane * fa.a_plane + (size_t)it.node * 256 + piece * 8;
        const int lo = __builtin_amdgcn_readfirstlane(plane * FZ_APL + rb * 16 * H2_BK);
#pragma unroll
        for (int c = 0; c < 8; ++c) __builtin_amdgcn_global_load_lds((gptr)(ga + c * H2_BK), (lptr)(As + c * 2 * FZ_APL + lo), 16, 0, 0);
    };
    FzItem cur, nxt;
    int item = blockIdx.x;
    fz_load_item(fa, tid0, tid0 >> 6, item, cur);
#pragma unroll
    for (int d = 0; d < FZ_D; ++d) ldB(tid0, cur, d, bq + 4 * d);
    dmaA(tid0, cur);
    __builtin_amdgcn_sched_barrier(0);
    for (;;) {
        // (the thread index is laundered once per item: everything derived from it -- dozens of per-lane LDS and global offsets -- would
        // otherwise be hoisted out of this loop and spilled: 210 scratch registers in the first persistent build)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int s = wave >> 2, wq = wave & 3;                    // phase 1: slot (row half of the tile), 64 of the half's 256 columns
        const int nt = wave & 3, mh = wave >> 2;                   // phase 2: 32 of the 128 decoder columns, row half
        const int a_row = (wave & 3) * 16 + (lane >> 2);           // A row this lane fetches in every chunk (plane wave >> 2)
        const unsigned short* wd1f_wave = fa.Wd1F + (size_t)nt * 16 * 2 * 512 + lane * 8;      // this wave's 32 decoder columns
        // the item's A planes and first fragments have landed, the previous item's stores are acknowledged (a store among counted
        // loads would break the counting: loads and stores return out of order), every wave is done with the previous S1 tile
        CCSP_TRK(0, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        CCSP_TRK(0, 2);
        int ea;
        h2_ld4(ea, fa.a_exp + cur.node);
        float tv[2];
        {
            const float* tp = fa.tau_t + (size_t)cur.type * 512 + cur.h * 256 + wq * 64 + (lane & 31);
            fz_ld_f32(tv[0], tp);
            fz_ld_f32(tv[1], tp + 32);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int lo = __builtin_amdgcn_readfirstlane((8 * wave + j) * 256);
            __builtin_amdgcn_global_load_lds((gptr)(fa.base + (size_t)cur.ur[j] * 512 + cur.h * 256 + lane * 4), (lptr)(Ut + lo), 16, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // --- phase 1: 32 rows (slot s) x 64 columns per wave, K = 256 in 16 k-steps; no barrier inside
        floatx16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        half8 af[2][2];                                            // [k-step parity][plane]: the A fragments of k-step ks + 1 are read under the MFMAs of ks
        auto ldA = [&](int ks, half8 (&a)[2]) {
            const unsigned short* Ac = As + (ks >> 1) * 2 * FZ_APL;
            const int piece = (lane >> 5) + 2 * (ks & 1);
            a[1] = *reinterpret_cast<const half8*>(Ac + FZ_APL + h2_off(s * 32 + (lane & 31), piece));
            a[0] = *reinterpret_cast<const half8*>(Ac + h2_off(s * 32 + (lane & 31), piece));
        };
        ldA(0, af[0]);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            half8* b = bq + 4 * (ks % FZ_D);
            __builtin_amdgcn_sched_barrier(0);
            // younger than the fragments of k-step ks: those of ks + 1 .. min(ks + D - 1, 15) (the refill of this slot is issued below);
            // the exponent, tau and base requests above are OLDER than every fragment requested inside the loop
            {
                const int n = 4 * ((ks + FZ_D - 1 < 15 ? ks + FZ_D - 1 : 15) - ks);
#define FZ_W(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) :: "memory")
                if (n >= 12) FZ_W(12); else if (n >= 8) FZ_W(8); else if (n >= 4) FZ_W(4); else FZ_W(0);
#undef FZ_W
            }
            if (ks + 1 < 16) ldA(ks + 1, af[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const half8 a0 = af[ks & 1][0], a1 = af[ks & 1][1];
            // smallest terms first (h2_kstep): (a lo, b hi), (a hi, b lo), (a hi, b hi); b[2 j + plane]
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[2], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[1], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[3], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[2], acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);                     // (the MFMAs have read the slot: it can be refilled)
            if (ks + FZ_D < 16) ldB(tid, cur, ks + FZ_D, b);
            if ((ks & 3) == 3) CCSP_TRK(0, 3 + (ks >> 2));
        }
        __builtin_amdgcn_sched_barrier(0);
        // --- exponents, tau and every base row of this wave have landed; the first half of the decoder weight is requested into the ring
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ea), "+v"(tv[0]), "+v"(tv[1]) :: "memory");
        {
            const unsigned short* p = wd1f_wave;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                fz_ld_frag<0>(bq[4 * g], p + g * 2048); fz_ld_frag<1024>(bq[4 * g + 1], p + g * 2048);
                fz_ld_frag<2048>(bq[4 * g + 2], p + g * 2048); fz_ld_frag<3072>(bq[4 * g + 3], p + g * 2048);
            }
        }
        if (wave < 4 && (lane & 3) == 0) sExpA[a_row] = ea;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // every wave's base rows are in the U tile; the row exponents are visible
        __builtin_amdgcn_sched_barrier(0);
        CCSP_TRK(0, 7);
        // --- U_h = 2^-(e_row + e_w) acc + (base + tau), in place (accumulator layout: 32 consecutive columns of one row per half-wave)
        {
            if (s != 0) { tv[0] = 0.0f; tv[1] = 0.0f; }           // (the time term rides on slot-0 rows; slot-1 rows add +0 like k_rowgemm_h2)
            int ex[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int4 e4 = *reinterpret_cast<const int4*>(sExpA + s * 32 + 8 * q + 4 * (lane >> 5));
                ex[4 * q] = -(e4.x + fa.w_exp); ex[4 * q + 1] = -(e4.y + fa.w_exp); ex[4 * q + 2] = -(e4.z + fa.w_exp); ex[4 * q + 3] = -(e4.w + fa.w_exp);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float* up = Ut + (s * 32 + 4 * (lane >> 5)) * 256 + wq * 64 + j * 32 + (lane & 31);
                float bv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) bv[r] = up[((r & 3) + 8 * (r >> 2)) * 256];
#pragma unroll
                for (int r = 0; r < 16; ++r) up[((r & 3) + 8 * (r >> 2)) * 256] = ldexpf(acc[j][r], ex[r]) + (bv[r] + tv[j]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        CCSP_TRK(0, 8);
        // --- row maxima of U_h (the bound behind the decoder rows' exponents): 16 lanes per row, four 16-byte reads each
        {
            const int c16 = tid & 15, br = tid >> 4;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int row = 32 * pass + br;
                float m = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 v = *reinterpret_cast<const float4*>(Ut + row * 256 + k * 64 + c16 * 4);
                    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                }
                m = fz_max16(m);
                if (c16 == 0) sMax[row] = m;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        CCSP_TRK(0, 9);
        int aexp[4], lu0[4], lu1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lu0[i] = (int)((cur.lu[i >> 1] >> (16 * (i & 1))) & 0xffu);
            lu1[i] = (int)((cur.lu[i >> 1] >> (16 * (i & 1) + 8)) & 0xffu);
            // |SiLU(z)| <= |z| <= max|U_h[u0]| + max|U_h[u1]|   (k_edge_h2 forms the same sum from umax)
            aexp[i] = h2_scale_exp(sMax[lu0[i]] + sMax[lu1[i]]);
            if ((tid & 15) == 0) sE[32 * i + (tid >> 4)] = aexp[i];
        }
        // --- the next item's indices are requested now and used after the decoder's K loop
        const bool more = item + (int)gridDim.x < fa.n_items;      // (uniform)
        if (more) fz_load_item(fa, tid, wave, item + gridDim.x, nxt);
        // --- phase 2: decoder layer 1 over the tile's edges
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        const int ne = cur.ne, MT = (ne + 31) >> 5;
        if (ne > 64) fz_decode_loop<2>(tid, wave, wd1f_wave, smem, ne, lu0, lu1, aexp, bq, acc);
        else fz_decode_loop<1>(tid, wave, wd1f_wave, smem, ne, lu0, lu1, aexp, bq, acc);
        CCSP_TRK(0, 14);
        // --- every wave is past the last stage barrier: the stages (A region) and the U tile are free.  The next item's first weight
        //     fragments and its A planes are requested under this item's epilogue
        if (more) {
#pragma unroll
            for (int d = 0; d < FZ_D; ++d) ldB(tid, nxt, d, bq + 4 * d);
            dmaA(tid, nxt);
        }
        __builtin_amdgcn_sched_barrier(0);
        // epilogue: 2^-(e_row + e_w) acc + bias -> SiLU -> S1 (in the U tile's LDS, swizzled)
        float* S1 = Ut;
        {
            const int MPW = ne > 64 ? 2 : 1;
            const int col = nt * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int mt = MPW == 1 ? mh : 2 * mh + i;
                if (i < MPW && mt < MT) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        const float q = ldexpf(acc[i][r], -(sE[row] + fa.wd_exp)) + bj;
                        S1[fz_s1_off(row, col)] = silu_fast(q);
                    }
                }
            }
        }
        __syncthreads();
        CCSP_TRK(0, 15);
        // second decoder layer: one (row, p) dot product per thread (four lanes per row), the four chains and their sum as
        // k_edge_h2<., ., 0> forms them
        for (int pass = 0; pass < (P > 4 ? 2 : 1); ++pass) {
            const int row = pass == 0 ? tid >> 2 : tid & 127, p = pass == 0 ? tid & 3 : 4 + (tid >> 7);
            if (row < ne && p < P) {
                const float* sr = S1 + row * 128;
                const float4* wr = reinterpret_cast<const float4*>(W2s + p * FZ_W2_LD);
                const int sw = row & 31;
                float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f, o3 = 0.0f;
#pragma unroll 8
                for (int j = 0; j < 32; ++j) {
                    const float4 sv = *reinterpret_cast<const float4*>(sr + ((j ^ sw) << 2)), wv = wr[j];
                    o0 = fmaf(sv.x, wv.x, o0); o1 = fmaf(sv.y, wv.y, o1); o2 = fmaf(sv.z, wv.z, o2); o3 = fmaf(sv.w, wv.w, o3);
                }
                const float o = ((o0 + o1) + (o2 + o3)) + sB2[p];
                const int slot = pass == 0 ? cur.o_slot : fa.ent_pos[2 * (cur.e0 + row) + cur.h];
                fa.O[(size_t)slot * P + p] = o;
            }
        }
        CCSP_TRK(0, 16);
        if (!more) break;
        item += gridDim.x;
        cur = nxt;
    }
#ifdef CCSP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    CCSP_TRK_RT(0, 31);
}

// ------------------------------------------------------------------------------------------
// k_eval_fused4 -- third form (the default of CCSP_EVAL=fused): TWO 256-thread workgroups per compute unit.
// What the first two forms measured (profiles/r04_findings.md): a tile's time is the SUM of a phase bound by the CU's L2 port (the
// weight stream of the row GEMM: 512 KB per tile, 42-53 B/clk/CU against 64 peak, matrix pipe half idle) and of phases that move
// nothing through that port (U epilogue, row maxima, the decoder's SiLU / split on the VALU, its MFMAs, the second layer) -- and one
// workgroup per CU runs them one after the other.  With two independent workgroups on a CU the hardware overlaps one tile's weight
// stream with the other tile's decoder.  That needs a tile in <= 80 KB of LDS and 4 waves per workgroup (<= 256 VGPRs at two waves per
// SIMD): 28 U rows per slot (56 KB U tile; the pose-embedding planes of phase 1 use the same bytes plus the decoder's 16 KB stage,
// the S1 tile of the epilogue the same bytes again), <= 112 edges, decoder K chunks of 32 through ONE stage whose next contents are
// built in registers under the current chunk's MFMAs.  Phase 1: wave = (slot, column half), 32 x 128 per wave, fragments from the
// same k_pack_wp_frag layout (two 4 KB runs per k-step), three k-steps ahead.  Same operands, exponents, MFMA order and epilogue
// arithmetic as the other forms: bitwise the edge outputs of the two-launch path.
// ------------------------------------------------------------------------------------------
constexpr int F4_RS = 28;                     // U rows per slot of a tile (A rows are padded to the 32-row MFMA tile)
constexpr int F4_ME = 112;                    // edges per tile
constexpr int F4_D = 3;                       // k-steps of weight fragments in flight per wave (phase 1): 3 x 8 KB
constexpr int F4_UT_BYTES = 2 * F4_RS * 256 * 4;                   // 57 344: U tile [56][256]; S1 tile [112][128] (swizzled quads)
constexpr int F4_STG_BYTES = 2 * 128 * H2_BK * 2;                  // 16 384: the decoder's A stage [2 planes][128][32]
constexpr int F4_LDS_BYTES = F4_UT_BYTES + F4_STG_BYTES + 8 * FZ_W2_LD * 4 + 8 * 4 + 64 * 4 + 64 * 4 + 128 * 4;
static_assert(F4_UT_BYTES + F4_STG_BYTES >= 8 * 2 * FZ_APL * 2, "the pose-embedding planes of phase 1 span the U tile and the stage");
static_assert(F4_LDS_BYTES <= 80 * 1024, "two workgroups per CU");

template <int N>
__device__ __forceinline__ void f4_wait8(half8* b) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void f4_wait4(half8* b) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N) : "memory");
}

__global__ __launch_bounds__(256, 2) void k_eval_fused4(FusedArgs fa) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[F4_LDS_BYTES];
    using gptr = const __attribute__((address_space(1))) void*;
    using lptr = __attribute__((address_space(3))) void*;
    float* Ut = reinterpret_cast<float*>(smem);                                           // [56][256] fp32; S1 after the decoder's K loop
    unsigned short* As = reinterpret_cast<unsigned short*>(smem);                         // phase 1: [8 chunks][2 planes][64][32] (over Ut and the stage)
    unsigned short* stg = reinterpret_cast<unsigned short*>(smem + F4_UT_BYTES);          // phase 2: [2 planes][128][32]
    float* W2s = reinterpret_cast<float*>(smem + F4_UT_BYTES + F4_STG_BYTES);             // [8][FZ_W2_LD]
    float* sB2 = W2s + 8 * FZ_W2_LD;
    int* sExpA = reinterpret_cast<int*>(sB2 + 8);                                         // [64] exponents of the pose-embedding rows (A row order)
    float* sMax = reinterpret_cast<float*>(sExpA + 64);                                   // [64] max |U_h| per tile row
    int* sE = reinterpret_cast<int*>(sMax + 64);                                          // [128] exponents of the decoder's A rows
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int s = wave >> 1, ch = wave & 1;                        // phase 1: slot, column half (128 of the half's 256 columns)
    const int P = fa.P;
    CCSP_TRK(0, 0);
    CCSP_TRK_RT(0, 30);
    const int w = fa.order[blockIdx.x];
    const int tile = w >> 1, h = w & 1;
    const int4 td = fa.tiles[tile];
    const int type = td.x, e0 = td.y, ne = td.z;
    const int MT = (ne + 31) >> 5;
    // --- the weight stream starts first: two runs of four fragments per k-step (column groups 2 ch and 2 ch + 1 of k_pack_wp_frag)
    const unsigned short* bptr = fa.WpF + ((((size_t)(2 * type + s) * 2 + h) * 4 + 2 * ch) * 16) * 4 * 512 + lane * 8;
    half8 bq[F4_D * 8];
    auto ldB = [&](int ks, half8* b) {
        const unsigned short* p0 = bptr + (size_t)ks * 4 * 512;
        const unsigned short* p1 = p0 + (size_t)16 * 4 * 512;
        fz_ld_frag<0>(b[0], p0); fz_ld_frag<1024>(b[1], p0); fz_ld_frag<2048>(b[2], p0); fz_ld_frag<3072>(b[3], p0);
        fz_ld_frag<0>(b[4], p1); fz_ld_frag<1024>(b[5], p1); fz_ld_frag<2048>(b[6], p1); fz_ld_frag<3072>(b[7], p1);
    };
#pragma unroll
    for (int d = 0; d < F4_D; ++d) ldB(d, bq + 8 * d);
    __builtin_amdgcn_sched_barrier(0);
    // --- index loads (ordinary loads; the compiler's waits for them also cover the fragments above, which are older)
    const int* trow = fa.rows + (size_t)tile * 128;
    const int a_row0 = (2 * (wave & 1)) * 16 + (lane >> 2);        // A rows this lane fetches in every chunk: a_row0, a_row0 + 16 (plane wave >> 1)
    const int node0 = trow[a_row0], node1 = trow[a_row0 + 16];
    const int urow_l = trow[64 + 14 * wave + (lane < 14 ? lane : 13)];       // U rows of the fourteen tile rows whose base this wave fetches
    int ur[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) ur[j] = __builtin_amdgcn_readlane(urow_l, j);
    const int br = tid >> 3, lq = tid & 7;                         // decoder A producer: edges 32 i + br, fp32 columns 4 lq .. + 3 of the chunk
    int lu0[4], lu1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int el = 32 * i + br;
        el = el < ne ? el : ne - 1;
        const unsigned int v = fa.e_lu[e0 + el];
        lu0[i] = (int)(v & 0xffu); lu1[i] = (int)(v >> 8);
    }
    int o_slot[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int row = 64 * i + (tid >> 2);
        row = row < ne ? row : ne - 1;
        o_slot[i] = fa.ent_pos[2 * (e0 + row) + h];
    }
    const float bj = fa.bd1[wave * 32 + (lane & 31)];
    for (int i = tid; i < 8 * 128; i += 256) W2s[(i >> 7) * FZ_W2_LD + (i & 127)] = i < P * 128 ? fa.Wd2[i < P * 128 ? i : 0] : 0.0f;
    if (tid < 8) sB2[tid] = tid < P ? fa.bd2[tid < P ? tid : 0] : 0.0f;
    CCSP_TRK(0, 1);
    // --- A planes of the 64 rows (LDS-DMA, source-side swizzle as k_rowgemm_h2 MODE 2): wave = (plane, two 16-row blocks)
    {
        const int plane = wave >> 1;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = a_row0 + 16 * q;
            const int piece = (lane & 3) ^ ((row >> 2) & 3);
            const unsigned short* ga = fa.A + (size_t)plane * fa.a_plane + (size_t)(q ? node1 : node0) * 256 + piece * 8;
            const int lo = __builtin_amdgcn_readfirstlane(plane * FZ_APL + (2 * (wave & 1) + q) * 16 * H2_BK);
#pragma unroll
            for (int c = 0; c < 8; ++c) __builtin_amdgcn_global_load_lds((gptr)(ga + c * H2_BK), (lptr)(As + c * 2 * FZ_APL + lo), 16, 0, 0);
        }
    }
    int ea0, ea1;
    h2_ld4(ea0, fa.a_exp + node0);
    h2_ld4(ea1, fa.a_exp + node1);
    float tv[4];
    {
        const float* tp = fa.tau_t + (size_t)type * 512 + h * 256 + ch * 128 + (lane & 31);
#pragma unroll
        for (int j = 0; j < 4; ++j) fz_ld_f32(tv[j], tp + 32 * j);
    }
    __builtin_amdgcn_sched_barrier(0);
    // the A planes have landed (younger: exponents 2, tau 4); W2s / sB2 are written; every wave's share is visible after the barrier
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    CCSP_TRK(0, 2);
    // --- phase 1: 32 rows (slot s) x 128 columns per wave, K = 256 in 16 k-steps; no barrier inside
    floatx16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    {
        half8 af[2][2];                                            // [k-step parity][plane]
        auto ldA = [&](int ks, half8 (&a)[2]) {
            const unsigned short* Ac = As + (ks >> 1) * 2 * FZ_APL;
            const int piece = (lane >> 5) + 2 * (ks & 1);
            a[1] = *reinterpret_cast<const half8*>(Ac + FZ_APL + h2_off(s * 32 + (lane & 31), piece));
            a[0] = *reinterpret_cast<const half8*>(Ac + h2_off(s * 32 + (lane & 31), piece));
        };
        ldA(0, af[0]);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            half8* b = bq + 8 * (ks % F4_D);
            __builtin_amdgcn_sched_barrier(0);
            // younger than the fragments of k-step ks: those of ks + 1 .. min(ks + D - 1, 15); the exponent / tau requests are older than
            // every fragment requested inside the loop
            {
                const int n = 8 * ((ks + F4_D - 1 < 15 ? ks + F4_D - 1 : 15) - ks);
                if (n >= 16) f4_wait8<16>(b); else if (n >= 8) f4_wait8<8>(b); else f4_wait8<0>(b);
            }
            if (ks + 1 < 16) ldA(ks + 1, af[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const half8 a0 = af[ks & 1][0], a1 = af[ks & 1][1];
            // smallest terms first (h2_kstep): (a lo, b hi), (a hi, b lo), (a hi, b hi); b[4 (j >> 1) + 2 (j & 1) + plane]
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[4 * (j >> 1) + 2 * (j & 1)], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[4 * (j >> 1) + 2 * (j & 1) + 1], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[4 * (j >> 1) + 2 * (j & 1)], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);                     // (the MFMAs have read the slot: it can be refilled)
            if (ks + F4_D < 16) ldB(ks + F4_D, b);
            if ((ks & 3) == 3) CCSP_TRK(0, 3 + (ks >> 2));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ea0), "+v"(ea1), "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]), "+v"(tv[3]) :: "memory");
    if ((wave >> 1) == 0 && (lane & 3) == 0) { sExpA[a_row0] = ea0; sExpA[a_row0 + 16] = ea1; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // every wave is done reading the A planes: the U tile's bytes are free
    __builtin_amdgcn_sched_barrier(0);
    // --- base rows into the U tile (LDS-DMA), the decoder's first weight fragments behind them
    const unsigned short* wd1f_wave = fa.Wd1F + (size_t)wave * 16 * 2 * 512 + lane * 8;      // this wave's 32 decoder columns
    half8 bw[12];                                                  // ring of three K chunks x 4 fragments ([k-step][plane])
    auto ldW = [&](int c, half8* b) {
        const unsigned short* p = wd1f_wave + (size_t)c * 2 * 2 * 512;
        fz_ld_frag<0>(b[0], p); fz_ld_frag<1024>(b[1], p); fz_ld_frag<2048>(b[2], p); fz_ld_frag<3072>(b[3], p);
    };
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        const int lo = __builtin_amdgcn_readfirstlane((14 * wave + j) * 256);
        __builtin_amdgcn_global_load_lds((gptr)(fa.base + (size_t)ur[j] * 512 + h * 256 + lane * 4), (lptr)(Ut + lo), 16, 0, 0);
    }
    ldW(0, bw); ldW(1, bw + 4); ldW(2, bw + 8);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");             // the base rows (older than the twelve weight fragments) have landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    CCSP_TRK(0, 7);
    // --- U_h = 2^-(e_row + e_w) acc + (base + tau), in place (accumulator layout); A rows 28..31 of the MFMA tile are padding
    {
        if (s != 0) { tv[0] = 0.0f; tv[1] = 0.0f; tv[2] = 0.0f; tv[3] = 0.0f; }      // (the time term rides on slot-0 rows; slot-1 rows add +0)
        int ex[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int4 e4 = *reinterpret_cast<const int4*>(sExpA + s * 32 + 8 * q + 4 * (lane >> 5));
            ex[4 * q] = -(e4.x + fa.w_exp); ex[4 * q + 1] = -(e4.y + fa.w_exp); ex[4 * q + 2] = -(e4.z + fa.w_exp); ex[4 * q + 3] = -(e4.w + fa.w_exp);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float* up = Ut + (s * F4_RS + 4 * (lane >> 5)) * 256 + ch * 128 + j * 32 + (lane & 31);
            float bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                bv[r] = (r < 12 || lane < 32) ? up[rr * 256] : 0.0f;                   // (rows 28..31: r >= 12 in the upper half-wave)
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                if (r < 12 || lane < 32) up[rr * 256] = ldexpf(acc[j][r], ex[r]) + (bv[r] + tv[j]);
            }
        }
    }
    __syncthreads();
    CCSP_TRK(0, 8);
    // --- row maxima of U_h: 16 lanes per row, four 16-byte reads each; 16 rows per pass
    {
        const int c16 = tid & 15, r16 = tid >> 4;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = 16 * pass + r16;
            float m = 0.0f;
            if (row < 2 * F4_RS) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 v = *reinterpret_cast<const float4*>(Ut + row * 256 + k * 64 + c16 * 4);
                    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                }
            }
            m = fz_max16(m);
            if (c16 == 0 && row < 2 * F4_RS) sMax[row] = m;
        }
    }
    __syncthreads();
    CCSP_TRK(0, 9);
    int aexp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // |SiLU(z)| <= |z| <= max|U_h[u0]| + max|U_h[u1]|   (k_edge_h2 forms the same sum from umax)
        aexp[i] = h2_scale_exp(sMax[lu0[i]] + sMax[lu1[i]]);
        if (lq == 0) sE[32 * i + br] = aexp[i];
    }
    // --- phase 2: decoder layer 1, K = 256 in eight chunks of 32 through one stage; wave = 32 of the 128 decoder columns, all row tiles
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    uint2 pb[4][2];                                                // the next chunk's A rows of this thread, both planes: built under the MFMAs
    auto prebuild = [&](int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < MT) {
                const float4 ua = *reinterpret_cast<const float4*>(Ut + lu0[i] * 256 + c * H2_BK + lq * 4);
                const float4 ub = *reinterpret_cast<const float4*>(Ut + lu1[i] * 256 + c * H2_BK + lq * 4);
                const float hv[4] = {silu_fast(ua.x + ub.x), silu_fast(ua.y + ub.y), silu_fast(ua.z + ub.z), silu_fast(ua.w + ub.w)};
                unsigned short p1[4], p2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split2h(ldexpf(hv[e], aexp[i]), p1[e], p2[e]);
                pb[i][0] = make_uint2(p1[0] | ((unsigned)p1[1] << 16), p1[2] | ((unsigned)p1[3] << 16));
                pb[i][1] = make_uint2(p2[0] | ((unsigned)p2[1] << 16), p2[2] | ((unsigned)p2[3] << 16));
            }
        }
    };
    prebuild(0);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        half8* b = bw + 4 * (c % 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < MT) {
                unsigned short* d = stg + h2_off(32 * i + br, lq >> 1) + (lq & 1) * 4;
                *reinterpret_cast<uint2*>(d) = pb[i][0];
                *reinterpret_cast<uint2*>(d + 128 * H2_BK) = pb[i][1];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // this chunk's weight fragments have landed (younger: the chunks c + 1, c + 2 of the ring)
        {
            const int n = 4 * ((c + 2 < 7 ? c + 2 : 7) - c);
            if (n >= 8) f4_wait4<8>(b); else if (n >= 4) f4_wait4<4>(b); else f4_wait4<0>(b);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // the stage holds chunk c
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int piece = (lane >> 5) + 2 * ks;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < MT) {
                    const half8 a0 = *reinterpret_cast<const half8*>(stg + h2_off(i * 32 + (lane & 31), piece));
                    const half8 a1 = *reinterpret_cast<const half8*>(stg + 128 * H2_BK + h2_off(i * 32 + (lane & 31), piece));
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[2 * ks], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[2 * ks + 1], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[2 * ks], acc[i], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c + 3 < 8) ldW(c + 3, b);
        if (c + 1 < 8) prebuild(c + 1);                            // (VALU + LDS reads of the U tile, in the shadow of the MFMAs just issued)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // every wave is done reading the stage
        __builtin_amdgcn_sched_barrier(0);
    }
    CCSP_TRK(0, 14);
    // --- epilogue: 2^-(e_row + e_w) acc + bias -> SiLU -> S1 (in the U tile's LDS, swizzled quads)
    float* S1 = Ut;
    {
        const int col = wave * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < MT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < F4_ME) {
                        const float q = ldexpf(acc[i][r], -(sE[row] + fa.wd_exp)) + bj;
                        S1[fz_s1_off(row, col)] = silu_fast(q);
                    }
                }
            }
        }
    }
    __syncthreads();
    CCSP_TRK(0, 15);
    // second decoder layer: one (row, p) dot product per thread (four lanes per row), chains and sum as k_edge_h2<., ., 0> forms them
    auto layer2 = [&](int row, int p, int slot) {
        const float* sr = S1 + row * 128;
        const float4* wr = reinterpret_cast<const float4*>(W2s + p * FZ_W2_LD);
        const int sw = row & 31;
        float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f, o3 = 0.0f;
#pragma unroll 8
        for (int j = 0; j < 32; ++j) {
            const float4 sv = *reinterpret_cast<const float4*>(sr + ((j ^ sw) << 2)), wv = wr[j];
            o0 = fmaf(sv.x, wv.x, o0); o1 = fmaf(sv.y, wv.y, o1); o2 = fmaf(sv.z, wv.z, o2); o3 = fmaf(sv.w, wv.w, o3);
        }
        fa.O[(size_t)slot * P + p] = ((o0 + o1) + (o2 + o3)) + sB2[p];
    };
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 64 * i + (tid >> 2), p = tid & 3;
        if (row < ne && p < P) layer2(row, p, o_slot[i]);
    }
    for (int p0 = 4; p0 < P; p0 += 2) {                            // pose_dim > 4 (robot: 5): rows tid & 127, components p0 + (tid >> 7)
        const int row = tid & 127, p = p0 + (tid >> 7);
        if (row < ne && p < P) layer2(row, p, fa.ent_pos[2 * (e0 + row) + h]);
    }
#ifdef CCSP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    CCSP_TRK(0, 16);
    CCSP_TRK_RT(0, 31);
}

// ------------------------------------------------------------------------------------------
// k_rowgemm_h2d -- CCSP_ROW_MODE=7 of the forward row GEMM (round 4): what the fused kernels' phase 1 taught, applied to the
// two-launch path.  The other modes stage BOTH operands through LDS chunk by chunk with a barrier per chunk (a workgroup is a
// chain of eight load -> ds_write -> barrier -> ds_read -> MFMA round trips); phase 1 of k_eval_fused ran its K loop at the
// matrix pipe's pace because (a) the tile's pose-embedding planes are resident in LDS for the whole K (one LDS-DMA burst, one
// barrier) and (b) the weight fragments come straight from global memory in MFMA operand order (k_pack_wp_frag), eight k-steps
// ahead with counted waits -- no LDS traffic, no VALU and no barrier for B.  Here: 64-row x 128-column tiles, 4 waves, each wave
// ALL 64 rows x 32 columns (both row tiles share every B fragment: a weight byte is loaded once per workgroup), 64 KB of LDS and
// <= 256 VGPRs: two workgroups per CU.  Same operands, exponents and MFMA order as every other mode: U and umax are bitwise the same.
// Epilogue: accumulators -> wave-private LDS tile (over the A planes, after one barrier) -> rows of 128-byte segments; the maxima
// of a row's 64-column piece are combined across the two waves that hold its halves.
// ------------------------------------------------------------------------------------------
constexpr int RD_D = 8;                       // k-steps of weight fragments in flight per wave (2 KB each)
constexpr int RD_CW_LD = 36;                  // wave-private epilogue tile [64][36] floats
constexpr int RD_LDS_BYTES = 8 * 2 * FZ_APL * 2;                   // 65 536: A planes [8 chunks][2 planes][64][32]; the epilogue tiles after the K loop
static_assert(4 * 64 * RD_CW_LD * 4 + 4 * 64 * 4 <= RD_LDS_BYTES, "epilogue tiles and row maxima fit the A planes' bytes");

__global__ __launch_bounds__(256, 2) void k_rowgemm_h2d(const unsigned short* __restrict__ A, size_t a_plane, const int* __restrict__ a_exp,
                                                        const int* __restrict__ urow_node, const int4* __restrict__ tile_desc,
                                                        const unsigned short* __restrict__ WpF, int w_exp,
                                                        const float* __restrict__ base, const float* __restrict__ tau_t,
                                                        float* __restrict__ U, float* __restrict__ umax, StepRef ref, size_t tau_stride) {
    constexpr int ND = 512, NCT = 4;
    if (ref.skip && *ref.skip == 0) return;                       // (uniform) MALA reuse: the state has not moved since this was computed
    __shared__ __attribute__((aligned(16))) unsigned char smem[RD_LDS_BYTES + 64 * 4];
    using gptr = const __attribute__((address_space(1))) void*;
    using lptr = __attribute__((address_space(3))) void*;
    unsigned short* As = reinterpret_cast<unsigned short*>(smem);
    int* sE = reinterpret_cast<int*>(smem + RD_LDS_BYTES);
    if (ref.tab) tau_t += (size_t)ref.tab[*ref.counter].t * tau_stride;      // hipGraph mode: timestep from the device table
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = bid / NCT, ct = bid % NCT;
    const int4 td = tile_desc[tile];
    const int row0 = td.x, nrows = td.y, ts = td.z;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int colw = ct * 128 + wave * 32;                        // first of this wave's 32 columns
    // --- the weight stream: 2 fragments (planes) per k-step of this wave's 32-column tile, from the 64-column groups of k_pack_wp_frag
    const int c32 = colw >> 5;
    const unsigned short* bptr = WpF + ((((size_t)(ts * 2 + (c32 >> 3)) * 4 + ((c32 >> 1) & 3)) * 16) * 4 + 2 * (c32 & 1)) * 512 + lane * 8;
    half8 bq[RD_D][2];
    auto ldB = [&](int ks, half8 (&b)[2]) {
        const unsigned short* p = bptr + (size_t)ks * 4 * 512;
        fz_ld_frag<0>(b[0], p); fz_ld_frag<1024>(b[1], p);
    };
#pragma unroll
    for (int d = 0; d < RD_D; ++d) ldB(d, bq[d]);
    __builtin_amdgcn_sched_barrier(0);
    // --- A planes of the 64 rows: wave = (plane, two 16-row blocks), source-side swizzle (k_rowgemm_h2 MODE 2)
    const int a_row0 = (2 * (wave & 1)) * 16 + (lane >> 2);
    int node[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int r = a_row0 + 16 * q;
        r = r < nrows ? r : nrows - 1;
        node[q] = urow_node ? urow_node[row0 + r] : row0 + r;
    }
    {
        const int plane = wave >> 1;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = a_row0 + 16 * q;
            const int piece = (lane & 3) ^ ((row >> 2) & 3);
            const unsigned short* ga = A + (size_t)plane * a_plane + (size_t)node[q] * 256 + piece * 8;
            const int lo = __builtin_amdgcn_readfirstlane(plane * FZ_APL + (2 * (wave & 1) + q) * 16 * H2_BK);
#pragma unroll
            for (int c = 0; c < 8; ++c) __builtin_amdgcn_global_load_lds((gptr)(ga + c * H2_BK), (lptr)(As + c * 2 * FZ_APL + lo), 16, 0, 0);
        }
    }
    int ea[2];
    h2_ld4(ea[0], a_exp + node[0]);
    h2_ld4(ea[1], a_exp + node[1]);
    // the time term of the wave's columns in the epilogue's row layout (slot-0 tiles; other tiles read `base` bytes and discard them)
    const int er = lane >> 3, eq = lane & 7;                       // epilogue: rows er + 8 st, columns 4 eq .. + 3
    const bool has_tau = tau_t && (ts & 1) == 0;
    h2_f4 tv;
    h2_ld16(tv, (has_tau ? tau_t + (size_t)(ts >> 1) * ND : base) + colw + 4 * eq);
    __builtin_amdgcn_sched_barrier(0);
    // the A planes have landed (younger: 2 exponents, the time term)
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    floatx16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    h2_f4 bs[8];                                                  // base values of the wave's 64 x 32 tile in the epilogue's layout, requested under the last k-steps
    {
        half8 af[2][2][2];                                         // [k-step parity][row tile][plane]
        auto ldA = [&](int ks, half8 (&a)[2][2]) {
            const unsigned short* Ac = As + (ks >> 1) * 2 * FZ_APL;
            const int piece = (lane >> 5) + 2 * (ks & 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i][1] = *reinterpret_cast<const half8*>(Ac + FZ_APL + h2_off(i * 32 + (lane & 31), piece));
                a[i][0] = *reinterpret_cast<const half8*>(Ac + h2_off(i * 32 + (lane & 31), piece));
            }
        };
        ldA(0, af[0]);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            half8 (&b)[2] = bq[ks % RD_D];
            __builtin_amdgcn_sched_barrier(0);
            // younger than the fragments of k-step ks: those of ks + 1 .. min(ks + D - 1, 15), and (ks >= 12) the base requests below
            {
                const int n = 2 * ((ks + RD_D - 1 < 15 ? ks + RD_D - 1 : 15) - ks) + (ks > 12 ? 8 : 0);
#define RD_W(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(b[0]), "+v"(b[1]) :: "memory")
                if (n >= 14) RD_W(14); else if (n >= 12) RD_W(12); else if (n >= 10) RD_W(10); else if (n >= 8) RD_W(8); else if (n >= 6) RD_W(6);
                else if (n >= 4) RD_W(4); else if (n >= 2) RD_W(2); else RD_W(0);
#undef RD_W
            }
            if (ks + 1 < 16) ldA(ks + 1, af[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            // smallest terms first (h2_kstep): (a lo, b hi), (a hi, b lo), (a hi, b hi)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][i][1], b[0], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][i][0], b[1], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][i][0], b[0], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + RD_D < 16) ldB(ks + RD_D, b);
            if (ks == 12) {                                        // base rows of the tile (clamped: rows past the tile's end are never stored)
#pragma unroll
                for (int st = 0; st < 8; ++st) {
                    const int trow = er + 8 * st;
                    const int tr = trow < nrows ? trow : nrows - 1;
                    h2_ld16_base(bs[st], base + (size_t)(row0 + tr) * ND + colw + 4 * eq);
                }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ea[0]), "+v"(ea[1]), "+v"(tv) :: "memory");
    asm volatile("" : "+v"(bs[0]), "+v"(bs[1]), "+v"(bs[2]), "+v"(bs[3]), "+v"(bs[4]), "+v"(bs[5]), "+v"(bs[6]), "+v"(bs[7]) :: "memory");
    if ((wave >> 1) == 0 && (lane & 3) == 0) { sE[a_row0] = ea[0]; sE[a_row0 + 16] = ea[1]; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // every wave is done reading the A planes; the row exponents are visible
    __builtin_amdgcn_sched_barrier(0);
    if (!has_tau) tv = h2_f4{0.f, 0.f, 0.f, 0.f};
    float* Cw = reinterpret_cast<float*>(smem) + wave * 64 * RD_CW_LD;
    float* sMaxW = reinterpret_cast<float*>(smem) + 4 * 64 * RD_CW_LD;                      // [4 waves][64 rows]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            Cw[rr * RD_CW_LD + (lane & 31)] = acc[i][r];
        }
    asm volatile("" ::: "memory");                                // (compiler ordering only: one wave's LDS operations execute in order)
    float* const Ub = U + (size_t)row0 * ND + colw + 4 * eq;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        const int trow = er + 8 * st;
        const int e = -(sE[trow] + w_exp);
        const float4 v = *reinterpret_cast<const float4*>(Cw + trow * RD_CW_LD + 4 * eq);
        const h2_f4 bt = bs[st] + tv;
        float4 o;
        o.x = ldexpf(v.x, e) + bt[0]; o.y = ldexpf(v.y, e) + bt[1]; o.z = ldexpf(v.z, e) + bt[2]; o.w = ldexpf(v.w, e) + bt[3];
        float m = fmaxf(fmaxf(0.0f, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        m = h2_max8(m);
        if (trow < nrows) h2_store_u(Ub + (size_t)trow * ND, o);
        if (eq == 0) sMaxW[wave * 64 + trow] = m;
    }
    __syncthreads();
    if (tid < 128) {                                              // the two 32-column halves of a 64-column piece
        const int row = tid & 63, pc = tid >> 6;
        if (row < nrows) umax[(size_t)(row0 + row) * 8 + 2 * ct + pc] = fmaxf(sMaxW[(2 * pc) * 64 + row], sMaxW[(2 * pc + 1) * 64 + row]);
    }
}
