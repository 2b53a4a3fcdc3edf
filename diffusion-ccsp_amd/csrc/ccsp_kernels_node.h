// ccsp_kernels_node.h -- the node kernels: k_node (every mode), k_node_direct (the straight-line direct-mode form), their argument blocks.
// A fragment of the ONE translation unit csrc/ccsp_hip.hip (included there, at this position, inside its namespaces): not a standalone header.
// ------------------------------------------------------------------------------------------
// k_node: per node  (1) eps = ordered CSR sum of edge outputs / sqrt(count), mask fill
//                   (2) pose update: ancestral p_sample or one ULA step (+ end-of-timestep reset)
//                   (3) pose encoder of the updated pose for the next evaluation
// ------------------------------------------------------------------------------------------
enum { STEP_NONE = 0, STEP_ANCESTRAL = 1, STEP_ULA = 2, STEP_INIT = 3, STEP_MALA_PROPOSE = 4, STEP_MALA_ACCEPT = 5 };

struct NoiseArg {
    int mode;               // CCSP_NOISE_*
    unsigned long long seed;
    unsigned long long row_offset;
    const float* normal;    // injected block for this call ([N,P]) or nullptr
    unsigned int call;      // philox call index
    const float* uniform;   // injected rand(N) block of this MALA inner step or nullptr
    unsigned int ucall;     // philox uniform-call index
};

// Replayable launches (hipGraph mode of ccsp_chain_run): everything that changes from one evaluation to the
// next -- timestep, update type, schedule scalars, noise call index, history slot -- is read from a device table
// at the position of a device counter instead of arriving as kernel arguments, so one captured graph of
// (1 + S) x 3 launches serves every timestep of every chain on the same ccsp_graph.  The row GEMM reads
// entry [counter] (its timestep), the edge kernel advances the counter, the node kernel reads entry [counter - 1].
struct StepEntry {
    int t, step, reset_mask, hist_slot;
    unsigned int call;
    float a_t, b_t, c1, c2, sigma, kappa, ss, std_;
    int pad[3];
};
struct ChainHeader {
    unsigned long long seed, row_offset, call_base, np_total;
    float* hist;
    const float* normal;
    int noise_mode, pad;
};
// Noise ahead (round 5, profiles/r05_findings.md section 6): the normal draws of an evaluation's node update need no data, but Philox + Box-Muller
// is ~500 dependent VALU instructions -- with one wave per SIMD 5-6 k cycles, the longest single piece of the node kernel (13 k), which sits on the
// chain of every evaluation.  The row GEMM that opens the evaluation carries them out instead: `blocks` extra workgroups behind its tile list write
// z[N, P] (256 elements each, the same ccsp::philox_normal call per element), and the node kernel reads z like an injected stream.
struct NoiseAhead {
    float* z;               // [N, P] or null
    int N, P, blocks;
    unsigned int call;
    unsigned long long seed, row_offset;
};
__device__ __forceinline__ void noise_ahead_block(const NoiseAhead& na, int blk) {
    const long idx = (long)blk * 256 + threadIdx.x;
    if (idx >= (long)na.N * na.P) return;
    const int n = (int)(idx / na.P), p = (int)(idx - (long)n * na.P);
    na.z[idx] = ccsp::philox_normal(na.seed, na.row_offset + (unsigned long long)n, na.call, p);
}

__global__ void k_noise_ahead(NoiseAhead na) { noise_ahead_block(na, (int)blockIdx.x); }      // (experiment paths without k_rowgemm_h2)

struct StepRef {
    const StepEntry* tab;
    int* counter;
    const int* skip;        // MALA reuse (CCSP_MALA_REUSE): if non-null and *skip == 0 the launch returns at once
    Gate gate;              // relay mode
    NoiseAhead na;          // forward GEMM of a direct-mode chain: the evaluation's normal draws (or z == null)
    const int* tile_rows;   // forward GEMM: the plane row of every tile row, padded per tile ([tile][TM]; rows past the tile's end repeat its last row) --
                            // its address needs the workgroup index only, so the gather goes out WITH the tile descriptor instead of behind it
};

struct NodeArgs {
    int N, P, F;
    int normalize;
    int src;                // 0: reduce O through the CSR; 1: eps given in eps_buf; 2: none
    int step;               // STEP_*
    int reset_mask;         // x[mask] = gt[mask] after the update (end of a timestep)
    int do_encode;
    const int* node_ptr;
    const float* O;         // [2 E_act, P] in CSR order (a node's inputs are contiguous)
    const float* xfeat;     // batch.x [N,F]
    int pose_begin;
    const signed char* mask;
    float* x;               // pose state [N,P] (in/out)
    const float* x_in;      // if non-null, evaluate at x_in instead of x (single-evaluation API)
    float* eps_out;         // [N,P] or nullptr
    const float* eps_buf;   // src == 1
    float* hist;            // history slot [N,P] or nullptr (written after the update)
    float* xhat;            // MALA proposal buffer [N,P]
    const float* E_x;       // MALA: batch energy at x and at the proposal (device scalars)
    const float* E_hat;
    const float* E_hat_partial;   // MALA accept: if non-null, E(x_hat) is the sum of these per-workgroup partials of the edge
    int n_hat_partial;            // kernel (same order as k_energy_sum) and E_hat is not read: one launch less per inner step
    // two coupled MALA lanes (round 6, MalaCouple): the batch energies are lane 0's + lane 1's, added in THAT order by both lanes' accept kernels
    // (the reference's energies are one scalar for the whole batch, ddpm.py:1026-1038).  E_x2 == nullptr: one lane
    const float* E_x2;            // the OTHER lane's E(x)
    const float* E_hat_partial2;  // ... and its proposal-energy partials
    int n_hat_partial2;
    int couple_second;            // 1: this lane is lane 1 (its own terms are the second summands)
    int* acc_count;         // MALA: accepted-node counter of this timestep
    int* changed;           // MALA reuse: reset by the propose step, += pose elements the accept step changed bitwise (or null)
    float* margin;          // MALA accept, debugging aid (ccsp_chain_margins): [N] log acceptance ratio - log u of this inner step, or null
    // schedule scalars of this timestep
    float a_t, b_t, c1, c2, sigma, kappa, ss, std_;
    NoiseArg noise;
    // hipGraph mode: the step-dependent fields above come from tab[*counter - 1] and *hdr
    const StepEntry* tab;
    const int* counter;
    const ChainHeader* hdr;
    Gate gate;              // relay mode
};

// The two update formulas of the direct-mode chain, shared by k_node and k_node_direct.  Every product and sum is rounded on
// its own (fp contract off: never fused into an FMA; HIP's __fmul_rn / __fadd_rn are plain operators and do get fused),
// which is the reference's arithmetic -- torch evaluates `grad * ss`, `noise * std` and the additions as separate rounded
// tensor operations -- and makes the two kernels agree bit for bit (left to -ffp-contract, hipcc fused different pairs in
// the two kernels: results one ulp apart).
__device__ __forceinline__ float step_ancestral(float xv, float eps, float z, float a_t, float b_t, float c1, float c2, float sigma) {   // ddpm.py:230-258
#pragma clang fp contract(off)
    const float x0 = a_t * xv - b_t * eps;
    const float mean = c1 * x0 + c2 * xv;
    return mean + sigma * z;
}
__device__ __forceinline__ float step_ula(float xv, float eps, float z, float kappa, float ss, float std_) {                            // ddpm.py:956-966
#pragma clang fp contract(off)
    const float grad = (-eps) * kappa;
    return (xv + grad * ss) + z * std_;
}

template <int H, bool ENCH /*second encoder layer on the f16 pipe (encode_tile_h2)*/>
__device__ __forceinline__ void node_body(NodeArgs a, const EncW w, const EncOut eo) {
    static_assert(!ENCH || H == 256, "the f16 encoder is written for hidden_dim 256");
    constexpr int S1_FLOATS = ENCH ? (2 * NODE_TILE * ENC_H2_LD) / 2 : NODE_TILE * (H / 2 + 1);
    __shared__ float xs[NODE_TILE][8];
    __shared__ __attribute__((aligned(16))) float s1raw[S1_FLOATS];      // layer-1 activations: fp32 rows, or two fp16 planes
    __shared__ float smax[4][NODE_TILE];
    __shared__ int sexp[NODE_TILE];
    // the node kernel is a short latency chain on the critical path of every evaluation; when it shares the
    // CUs with the other lane's GEMM kernels its waves should win the issue arbitration
    CCSP_TRK(2, 0);
    CCSP_TRK_RT(2, 30);
    __builtin_amdgcn_s_setprio(3);
    if (a.tab) {
        const StepEntry e = a.tab[*a.counter - 1];
        const ChainHeader h = *a.hdr;
        a.step = e.step; a.reset_mask = e.reset_mask;
        a.a_t = e.a_t; a.b_t = e.b_t; a.c1 = e.c1; a.c2 = e.c2; a.sigma = e.sigma; a.kappa = e.kappa; a.ss = e.ss; a.std_ = e.std_;
        a.noise.mode = h.noise_mode; a.noise.seed = h.seed; a.noise.row_offset = h.row_offset; a.noise.call = e.call;
        a.noise.normal = h.normal ? h.normal + (size_t)(e.call - h.call_base) * h.np_total : nullptr;
        a.hist = (h.hist && e.hist_slot >= 0) ? h.hist + (size_t)e.hist_slot * h.np_total : nullptr;
    }
    const int node0 = blockIdx.x * NODE_TILE;
    const int tid = threadIdx.x;
    if (a.step == STEP_MALA_PROPOSE && a.changed && blockIdx.x == 0 && tid == 0) *a.changed = 0;
    // The chain of this kernel is CSR range -> edge outputs -> update -> encoder.  Vector-memory loads return in order, so
    // the chain's loads are issued FIRST and the encoder's weights (160 VGPRs of them in the f16 form) behind them: they are
    // in flight under the update and never in front of a load the update waits for.
    int csr_beg = 0, csr_cnt = 0;
    float csr_v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) csr_v[j] = 0.0f;
    if (a.src == 0 && tid < NODE_TILE * 8) {
        const int n = node0 + tid / 8, p = tid % 8;
        if (n < a.N && p < a.P) {
            csr_beg = a.node_ptr[n];
            csr_cnt = a.node_ptr[n + 1] - csr_beg;
            const float* op = a.O + (size_t)csr_beg * a.P + p;
#pragma unroll
            for (int j = 0; j < 16; ++j) csr_v[j] = j < csr_cnt ? op[(size_t)j * a.P] : 0.0f;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    EncPrefetch<H> pf;
    EncPrefetchH pfh;
    if (a.do_encode) {
        if constexpr (ENCH) enc_prefetch_h2(w, pfh);
        else enc_prefetch<H>(w, pf);
    }
    float e_hat = 0.0f;
    if (a.step == STEP_MALA_ACCEPT) {                           // (uniform: kernel argument)
        if (a.E_hat_partial) {
            float v = 0.0f;
            for (int i = tid; i < a.n_hat_partial; i += 256) v += a.E_hat_partial[i];
            e_hat = block_sum_256(v, &smax[0][0]);
            if (a.E_x2) {                                           // (uniform) coupled lanes: lane 0's sum + lane 1's sum
                float v2 = 0.0f;
                for (int i = tid; i < a.n_hat_partial2; i += 256) v2 += a.E_hat_partial2[i];
                __syncthreads();                                    // (block_sum_256's scratch is read by every thread until here)
                const float e2 = block_sum_256(v2, &smax[0][0]);
                e_hat = a.couple_second ? e2 + e_hat : e_hat + e2;
            }
        } else {
            e_hat = a.E_hat[0];
        }
    }
    if (tid < NODE_TILE * 8) {
        const int nl = tid / 8, p = tid % 8;
        const int n = node0 + nl;
        float xnew = 0.0f;
        if (n < a.N && p < a.P) {
            const size_t i = (size_t)n * a.P + p;
            const bool masked = a.mask[n] != 0;
            float eps = 0.0f;
            if (a.src == 0) {
                // sixteen entries per round trip (a node of an 8-object graph has up to ~20), summed in CSR order; the
                // padding terms are +0.0f and change nothing.  The first sixteen were requested at kernel entry.
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < 16; ++j) acc += csr_v[j];
                const float* op = a.O + (size_t)csr_beg * a.P + p;
                for (int q0 = 16; q0 < csr_cnt; q0 += 16) {
                    float v[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) v[j] = q0 + j < csr_cnt ? op[(size_t)(q0 + j) * a.P] : 0.0f;
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc += v[j];
                }
                if (a.normalize) acc = acc / sqrtf((float)csr_cnt);            // 0/0 -> NaN like the reference
                eps = masked ? a.xfeat[(size_t)n * a.F + a.F - a.P + p] : acc; // out[mask] = x[:, -P:][mask]
            } else if (a.src == 1) {
                eps = a.eps_buf[i];
            }
            if (a.eps_out) a.eps_out[i] = eps;
            float xv = a.x_in ? a.x_in[i] : (a.step == STEP_INIT ? 0.0f : a.x[i]);
            const float x_old = xv;
            if (a.step != STEP_NONE) {
                float z = 0.0f;
                if (a.step != STEP_MALA_ACCEPT) {
                    if (a.noise.mode == CCSP_NOISE_INJECTED) z = a.noise.normal[i];
                    else z = ccsp::philox_normal(a.noise.seed, a.noise.row_offset + (unsigned long long)n, a.noise.call, p);
                }
                if (a.step == STEP_ANCESTRAL) {
                    xv = step_ancestral(xv, eps, z, a.a_t, a.b_t, a.c1, a.c2, a.sigma);
                } else if (a.step == STEP_ULA) {
                    xv = step_ula(xv, eps, z, a.kappa, a.ss, a.std_);
                } else if (a.step == STEP_MALA_PROPOSE) {       // ddpm.py:1017-1023: x_hat = (x + grad ss) + noise std
                    xv = step_ula(xv, eps, z, a.kappa, a.ss, a.std_);
                } else if (a.step == STEP_MALA_ACCEPT) {        // ddpm.py:1026-1041
                    // one decision per node row from the batch-scalar energies and the proposal densities
                    // (the reverse density uses the SAME mu as the forward one, like the reference)
                    // Round 4: every thread of a node row used to walk all P components itself -- a loop of three loads and a wait per
                    // component, P dependent round trips in a kernel that is one latency chain (10.8 us at C4).  Now a thread forms the
                    // two density terms of ITS component from the three values it loads once, and the row's threads (eight consecutive
                    // lanes) add the terms up in the same ascending order through lane reads: the same sums, bit for bit.
                    const float var = a.std_ * a.std_, log_scale = logf(a.std_), lc = 0.918938533204672742f;
                    const float xc = a.x[i], hc = a.xhat[i];
                    const float mu = xc + ((-a.eps_buf[i]) * a.kappa) * a.ss;
                    const float dr = xc - mu, df = hc - mu;
                    const float t_rev = -(dr * dr) / (2.0f * var) - log_scale - lc;
                    const float t_fwd = -(df * df) / (2.0f * var) - log_scale - lc;
                    const float ex0 = a.E_x2 ? (a.couple_second ? a.E_x2[0] + a.E_x[0] : a.E_x[0] + a.E_x2[0]) : a.E_x[0];
                    float lrev = 0.0f, lfwd = 0.0f;
                    const int lane0 = (tid & 63) & ~7;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {                // (lanes c < P of the row are live: same node, p = c)
                        const float r = __shfl(t_rev, lane0 + c), f = __shfl(t_fwd, lane0 + c);
                        lrev = c < a.P ? lrev + r : lrev;
                        lfwd = c < a.P ? lfwd + f : lfwd;
                    }
                    const float logp_x = (-ex0) * a.kappa, logp_h = (-e_hat) * a.kappa;
                    const float la = logp_h - logp_x + lrev - lfwd;
                    float u;
                    if (a.noise.mode == CCSP_NOISE_INJECTED) u = a.noise.uniform[n];
                    else u = ccsp::philox_uniform(a.noise.seed, a.noise.row_offset + (unsigned long long)n, a.noise.ucall);
                    const float accf = (u < expf(la)) ? 1.0f : 0.0f;
                    if (p == 0 && accf != 0.0f && a.acc_count) atomicAdd(a.acc_count, 1);
                    if (p == 0 && a.margin) {            // > 0 accepted, < 0 rejected; |margin| small against the terms it is the difference of = a near-tie
                        a.margin[n] = la - logf(u);
                        a.margin[a.N + n] = fabsf(logp_h) + fabsf(logp_x) + fabsf(lrev) + fabsf(lfwd);
                    }
                    xv = accf * hc + (1.0f - accf) * xv;
                } else {                                        // ddpm.py:273
                    xv = 0.5f * z;
                }
                if (a.step == STEP_MALA_PROPOSE) {
                    a.xhat[i] = xv;                             // the chain state x is untouched until the accept step
                } else {
                    if (a.reset_mask && masked) xv = a.xfeat[(size_t)n * a.F + a.pose_begin + p];
                    // MALA reuse: the next gradient evaluation may be skipped only if NO stored element moved.  A rejected node can
                    // move too (0 * Inf = NaN from a non-finite proposal, like the reference), so compare bit patterns
                    if (a.step == STEP_MALA_ACCEPT && a.changed && __float_as_uint(xv) != __float_as_uint(x_old)) atomicAdd(a.changed, 1);
                    a.x[i] = xv;
                    if (a.hist) a.hist[i] = xv;
                }
            }
            xnew = xv;
        }
        xs[nl][p] = xnew;
        if constexpr (ENCH) {        // row exponent of the encoder's layer-1 activations from the bound c1 max|x| + c2 (encode_tile_h2)
            float amax = fabsf(xnew);                           // (fmaxf skips a NaN pose: finite exponent, the NaN travels in the fp16 terms)
            amax = fmaxf(amax, __shfl_xor(amax, 1));
            amax = fmaxf(amax, __shfl_xor(amax, 2));
            amax = fmaxf(amax, __shfl_xor(amax, 4));
            if (p == 0) sexp[nl] = h2_scale_exp(fmaf(w.c1, amax, w.c2));
        }
    }
    if (!a.do_encode) return;
    __syncthreads();
    CCSP_TRK(2, 1);
    if constexpr (ENCH) encode_tile_h2(w, pfh, xs, reinterpret_cast<unsigned short*>(s1raw), sexp, smax, node0, a.N, eo);
    else encode_tile_mfma<H>(w, pf, xs, reinterpret_cast<float (*)[H / 2 + 1]>(s1raw), smax, node0, a.N, eo);
    CCSP_TRK(2, 5);
    CCSP_TRK_RT(2, 31);
}
template <int H, bool ENCH>
__global__ __launch_bounds__(256) void k_node(NodeArgs a, EncW w, EncOut eo) { gate_wait(a.gate); node_body<H, ENCH>(a, w, eo); gate_done(a.gate); }

// ------------------------------------------------------------------------------------------
// k_node_direct: k_node for what a direct-mode chain runs 11 000 times -- CSR reduce (src 0), ancestral or ULA step, f16
// encoder of the new pose, hidden_dim 256 -- as ONE straight-line latency chain.  k_node serves every mode through run-time
// branches, and on gfx950 (loads and stores on one counter, hipcc's wait insertion taking the minimum over control-flow
// paths) that cost it most of its time: sixteen CSR loads each under its own branch, the mask / pose / feature loads issued
// BEHIND the encoder's 128 KB of weights and waited for with vmcnt(0) -- 11 k of its 16.7 k cycles went by before the update
// was done (profiles/r03_findings.md).  Here every load of the chain is issued at entry, unconditionally (clamped indices,
// selects instead of branches), the CSR entries 32 per round trip, the noise draw is computed while they are in flight, and
// the encoder's weights are requested behind them: vector-memory loads return in order, so nothing the update needs waits
// for a weight.  Same arithmetic as k_node (same order of the CSR sum, shared step formulas): results are bitwise equal.
// ------------------------------------------------------------------------------------------
// encode_tile_h2 with the layer-2 weight fragments streamed per k-step (two register sets of 32 VGPRs) instead of held in 128:
// the form that fits next to the edge kernel's registers (node update folded into its tail).  Same products in the same order.
__device__ __forceinline__ void encode_tile_h2_stream(const EncW w, float (*xs)[8], unsigned short* s1h, int* sexp, float (*smax)[NODE_TILE],
                                                      int node0, int N, const EncOut out, int n_lim = -1) {
    constexpr int H = 256, LD = ENC_H2_LD;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const half8* wh = reinterpret_cast<const half8*>(w.W2H) + (size_t)wave * 16 * 64 + lane;
    half8 wa[2][2][4];                                            // [register set][plane][tile]
    auto wload = [&](int ks, int set) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) wa[set][p][q] = wh[(size_t)p * 4096 + (ks * 4 + q) * 64];
    };
    wload(0, 0);
    wload(1, 1);
    {
        const int j = tid % 128;
        float w0[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) w0[d] = d < w.in_dim ? w.W0[j * w.in_dim + d] : 0.0f;
        const float b0 = w.b0[j];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = tid / 128 + 2 * i;
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < 8; ++d) acc = fmaf(xs[n][d], w0[d], acc);         // columns >= in_dim are 0
            unsigned short h1, h2;
            split2h(ldexpf(silu_fast(acc + b0), sexp[n]), h1, h2);
            s1h[n * LD + j] = h1;
            s1h[(NODE_TILE + n) * LD + j] = h2;
        }
    }
    float b2[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) b2[q][r] = w.b2[wave * 64 + q * 16 + 4 * (lane >> 4) + r];
    __syncthreads();
    floatx4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    const unsigned short* bp = s1h + (lane & 15) * LD + 8 * (lane >> 4);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int set = ks & 1;
        const half8 b1 = *reinterpret_cast<const half8*>(bp + ks * 32);
        const half8 b2h = *reinterpret_cast<const half8*>(bp + NODE_TILE * LD + ks * 32);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[set][1][j], b1, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[set][0][j], b2h, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[set][0][j], b1, acc[j], 0, 0, 0);
        if (ks + 2 < 4) wload(ks + 2, set);
    }
    const int eu = -(sexp[lane & 15] + w.w2_exp);
    float v[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[j][r] = silu_fast(ldexpf(acc[j][r], eu) + b2[j][r]);
    enc_store_tile<H>(v, smax, node0, N, out, n_lim);
}

// LDS of one node block (k_node_direct's own; a region of the stages in the fused edge kernel)
struct NodeLds {
    float (*xs)[8];                 // [NODE_TILE][8]
    unsigned short* s1h;            // [2][NODE_TILE][ENC_H2_LD]
    float (*smax)[NODE_TILE];       // [4][NODE_TILE]
    int* sexp;                      // [NODE_TILE]
};
constexpr int NODE_LDS_BYTES = NODE_TILE * 8 * 4 + 2 * NODE_TILE * ENC_H2_LD * 2 + 4 * NODE_TILE * 4 + NODE_TILE * 4;
__device__ __forceinline__ NodeLds node_lds(void* base) {
    char* b = reinterpret_cast<char*>(base);
    NodeLds l;
    l.s1h = reinterpret_cast<unsigned short*>(b);                          // (first: 16-byte aligned fragment reads)
    l.xs = reinterpret_cast<float (*)[8]>(b + 2 * NODE_TILE * ENC_H2_LD * 2);
    l.smax = reinterpret_cast<float (*)[NODE_TILE]>(b + 2 * NODE_TILE * ENC_H2_LD * 2 + NODE_TILE * 8 * 4);
    l.sexp = reinterpret_cast<int*>(b + 2 * NODE_TILE * ENC_H2_LD * 2 + NODE_TILE * 8 * 4 + 4 * NODE_TILE * 4);
    return l;
}

// Tail of the node-grouped edge kernel: the workgroup's rows were the CSR entries [csr0, csr0 + rows) of nodes node0 .. node0 + nn - 1
// and their outputs are in LDS (Os[row][8]), so the update needs no other workgroup: CSR sum in the reference's order from LDS, count-
// normalise, mask fill, ancestral / ULA step, history, encoder of the new pose (streamed weights) -- the arithmetic of node_block_direct,
// bit for bit.  What the update needs from memory does not depend on the tile: node_group_pre requests it at kernel entry (and draws the
// noise under the loads), so the tail starts with everything but the outputs in registers.  All 256 threads.
struct NodeGroupPre {
    int csr_beg, csr_cnt;
    float x_old, xf_fill, xf_reset, z;
    bool masked;
};
__device__ __forceinline__ NodeGroupPre node_group_pre(const NodeArgs& a, int node0, int nn) {
    const int tid = threadIdx.x;
    const int nl = (tid >> 3) & (NODE_TILE - 1), p = tid & 7;
    const int nc = node0 + (nl < nn ? nl : 0), pc = p < a.P ? p : a.P - 1;
    const size_t i = (size_t)nc * a.P + pc;
    NodeGroupPre r;
    r.csr_beg = a.node_ptr[nc];
    r.csr_cnt = a.node_ptr[nc + 1] - r.csr_beg;
    r.masked = a.mask[nc] != 0;
    r.x_old = a.x[i];
    r.xf_fill = a.xfeat[(size_t)nc * a.F + a.F - a.P + pc];
    r.xf_reset = a.xfeat[(size_t)nc * a.F + a.pose_begin + pc];
    const bool injected = a.noise.mode == CCSP_NOISE_INJECTED;
    const float z_inj = (injected ? a.noise.normal : a.x)[i];
    const float z = ccsp::philox_normal(a.noise.seed, a.noise.row_offset + (unsigned long long)nc, a.noise.call, pc);
    r.z = injected ? z_inj : z;
    return r;
}
__device__ __forceinline__ void node_group_tail(const NodeArgs& a, const EncW& w, const EncOut& eo, int node0, int nn, int csr0, const float* __restrict__ Os,
                                                const NodeLds lds, const NodeGroupPre& pre) {
    const int tid = threadIdx.x;
    const int nl = (tid >> 3) & (NODE_TILE - 1), p = tid & 7;
    const bool live = tid < NODE_TILE * 8 && nl < nn && p < a.P;
    const int nc = node0 + (nl < nn ? nl : 0), pc = p < a.P ? p : a.P - 1;
    const size_t i = (size_t)nc * a.P + pc;
    const int csr_cnt = pre.csr_cnt;
    float acc = 0.0f;
    const float* op = Os + (pre.csr_beg - csr0) * 8 + pc;
    for (int q0 = 0; q0 < csr_cnt; q0 += 8) {                             // eight entries per LDS round trip, added in CSR order
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = op[(q0 + j < csr_cnt ? q0 + j : csr_cnt - 1) * 8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = q0 + j < csr_cnt ? acc + v[j] : acc;
    }
    if (a.normalize) acc = acc / sqrtf((float)csr_cnt);                   // 0/0 -> NaN like the reference
    const float eps = pre.masked ? pre.xf_fill : acc;
    float xv = a.step == STEP_ANCESTRAL ? step_ancestral(pre.x_old, eps, pre.z, a.a_t, a.b_t, a.c1, a.c2, a.sigma)
                                        : step_ula(pre.x_old, eps, pre.z, a.kappa, a.ss, a.std_);
    if (a.reset_mask && pre.masked) xv = pre.xf_reset;
    if (live) {
        a.x[i] = xv;
        if (a.hist) a.hist[i] = xv;
    }
    if (tid < NODE_TILE * 8) {
        const float xnew = live ? xv : 0.0f;
        lds.xs[nl][p] = xnew;
        float amax = fabsf(xnew);
        amax = fmaxf(amax, __shfl_xor(amax, 1));
        amax = fmaxf(amax, __shfl_xor(amax, 2));
        amax = fmaxf(amax, __shfl_xor(amax, 4));
        if (p == 0) lds.sexp[nl] = h2_scale_exp(fmaf(w.c1, amax, w.c2));
    }
    __syncthreads();
    encode_tile_h2_stream(w, lds.xs, lds.s1h, lds.sexp, lds.smax, node0, a.N, eo, node0 + nn);
}

// One 16-node block of the direct-mode update: CSR reduce in the reference's order, count-normalise, mask fill, ancestral /
// ULA step with its noise draw, mask reset, history, encoder of the new pose.  All 256 threads; ends with the planes stored.
// FUSED (tail of the edge kernel, run by the workgroup that delivered the block's last edge outputs): the edge outputs were
// stored write-through (sc1) by workgroups on any XCD and are read with sc1 loads -- the L2-served pair of
// cdna_hip_programming.md Guideline 16 -- and the encoder streams its weights.
template <bool FUSED, bool STREAM = FUSED /*encoder weights streamed per k-step instead of prefetched into 128 VGPRs*/>
__device__ __forceinline__ void node_block_direct(const NodeArgs& a, const EncW& w, const EncOut& eo, int n_ent, int node0, const NodeLds lds) {
    const int tid = threadIdx.x;
    const int nl = (tid >> 3) & (NODE_TILE - 1), p = tid & 7;
    const int n = node0 + nl;
    const bool live = tid < NODE_TILE * 8 && n < a.N && p < a.P;          // this thread owns pose element (n, p)
    const int nc = n < a.N ? n : a.N - 1, pc = p < a.P ? p : a.P - 1;     // clamped: every address below is valid for every thread
    const size_t i = (size_t)nc * a.P + pc;
    // ---- the chain's loads, all of them, before anything else
    const int csr_beg = a.node_ptr[nc], csr_end = a.node_ptr[nc + 1];
    const signed char mk = a.mask[nc];
    const float x_old = a.x[i];
    const float xf_fill = a.xfeat[(size_t)nc * a.F + a.F - a.P + pc];     // out[mask] = x[:, -P:][mask]
    const float xf_reset = a.xfeat[(size_t)nc * a.F + a.pose_begin + pc];
    const bool injected = a.noise.mode == CCSP_NOISE_INJECTED;
    const float z_inj = (injected ? a.noise.normal : a.x)[i];             // (a select, not a branch; discarded when not injected)
    const int csr_cnt = csr_end - csr_beg;
    auto o_load = [&](const float* ptr) -> float {
        if constexpr (FUSED) return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return *ptr;
    };
#ifdef CCSP_TRACE
    asm volatile("" :: "v"(csr_cnt));
    CCSP_TRK(2, 7);
#endif
    float v[32];
    {
        const float* op = a.O + pc;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            int e = csr_beg + (j < csr_cnt ? j : 0);
            e = e < n_ent ? e : n_ent - 1;                                // (isolated last node: csr_beg == n_ent)
            v[j] = o_load(op + (size_t)e * a.P);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    EncPrefetchH pfh;
    if constexpr (!STREAM) enc_prefetch_h2(w, pfh);                       // behind the chain: in flight under the update
    // ---- the noise draw needs no data: computed while the loads are in flight
    float z = z_inj;                                                      // (a uniform branch: an injected / drawn-ahead stream skips ~500 instructions)
    if (!injected) z = ccsp::philox_normal(a.noise.seed, a.noise.row_offset + (unsigned long long)nc, a.noise.call, pc);
#ifdef CCSP_TRACE
    asm volatile("" :: "v"(z));
    CCSP_TRK(2, 8);
#endif
    // ---- CSR sum in the reference's order, count-normalise, mask fill
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 32; ++j) acc = j < csr_cnt ? acc + v[j] : acc;
    for (int q0 = 32; q0 < csr_cnt; q0 += 16) {                           // (nodes with more than 32 inputs: rare)
        const float* op = a.O + (size_t)csr_beg * a.P + pc;
        float u[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) u[j] = q0 + j < csr_cnt ? o_load(op + (size_t)(q0 + j) * a.P) : 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = q0 + j < csr_cnt ? acc + u[j] : acc;
    }
    if (a.normalize) acc = acc / sqrtf((float)csr_cnt);                   // 0/0 -> NaN like the reference
#ifdef CCSP_TRACE
    asm volatile("" :: "v"(acc));
    CCSP_TRK(2, 9);
#endif
    const bool masked = mk != 0;
    const float eps = masked ? xf_fill : acc;
    float xv = a.step == STEP_ANCESTRAL ? step_ancestral(x_old, eps, z, a.a_t, a.b_t, a.c1, a.c2, a.sigma)
                                        : step_ula(x_old, eps, z, a.kappa, a.ss, a.std_);
    if (a.reset_mask && masked) xv = xf_reset;
    if (live) {
        a.x[i] = xv;
        if (a.hist) a.hist[i] = xv;
    }
    if (tid < NODE_TILE * 8) {
        const float xnew = live ? xv : 0.0f;
        lds.xs[nl][p] = xnew;
        float amax = fabsf(xnew);                                         // row exponent of the encoder's layer-1 activations (encode_tile_h2)
        amax = fmaxf(amax, __shfl_xor(amax, 1));
        amax = fmaxf(amax, __shfl_xor(amax, 2));
        amax = fmaxf(amax, __shfl_xor(amax, 4));
        if (p == 0) lds.sexp[nl] = h2_scale_exp(fmaf(w.c1, amax, w.c2));
    }
    CCSP_TRK(2, 6);
    __syncthreads();
    CCSP_TRK(2, 1);
    if constexpr (STREAM) encode_tile_h2_stream(w, lds.xs, lds.s1h, lds.sexp, lds.smax, node0, a.N, eo);
    else encode_tile_h2(w, pfh, lds.xs, lds.s1h, lds.sexp, lds.smax, node0, a.N, eo);
}

__global__ __launch_bounds__(256) void k_node_direct(NodeArgs a, EncW w, EncOut eo, int n_ent /*2 E_act >= 1*/) {
    __shared__ __attribute__((aligned(16))) char lds_raw[NODE_LDS_BYTES];
    CCSP_TRK(2, 0);
    CCSP_TRK_RT(2, 30);
    __builtin_amdgcn_s_setprio(3);
    gate_wait(a.gate);
    node_block_direct<false>(a, w, eo, n_ent, blockIdx.x * NODE_TILE, node_lds(lds_raw));
    CCSP_TRK(2, 5);
    CCSP_TRK_RT(2, 31);
    gate_done(a.gate);
}
#ifdef CCSP_EXPERIMENTS
// the same with the encoder's weights streamed (CCSP_NODE=stream, A/B): a third of the registers, so that its waves fit next to
// the other lane's GEMM waves on more SIMDs
__global__ __launch_bounds__(256, 3) void k_node_direct_s(NodeArgs a, EncW w, EncOut eo, int n_ent) {
    __shared__ __attribute__((aligned(16))) char lds_raw[NODE_LDS_BYTES];
    __builtin_amdgcn_s_setprio(3);
    node_block_direct<false, true>(a, w, eo, n_ent, blockIdx.x * NODE_TILE, node_lds(lds_raw));
}
#endif

// the node update folded into the edge kernel's tail (k_edge_h2 / k_edge_h2s, FUSE): which 16-node blocks a workgroup's
// outputs touch, how many workgroups touch each block, and the arrival counters (zeroed when a chain starts; `epoch` = index of
// this evaluation since then, from 1).  The workgroup whose arrival completes a block runs node_block_direct<true> for it --
// nobody waits for anybody, so no grid barrier and no spinning.
struct FuseArgs {
    // node-grouped form (CCSP_FUSE_NODE=2, k_edge_h2<.., NG>): a workgroup's rows ARE the CSR entries of its own run of nodes
    const int4* ng_desc;        // [workgroups] {first node, nodes (<= 16), first CSR entry, entries (<= 64)}
    const int* ng_off0;         // [workgroups][64] element offset into U of the row's first operand: U row * 2H + half * H (padding rows repeat row 0)
    const int* ng_off1;
    const int* wg_blk_ptr;      // [workgroups + 1]
    const int* wg_blk;          // node blocks, ascending, per workgroup
    const int* blk_expect;      // [node blocks]
    unsigned int* blk_count;    // [node blocks]
    unsigned int epoch;
    int n_ent;
    NodeArgs node;
    EncW w;
    EncOut eo;
};

