// ccsp_compose.h -- composition of two constraint domains: elementwise kernels and helpers (entry points: ccsp_abi_compose.h).
// A fragment of the ONE translation unit csrc/ccsp_hip.hip (included there, at this position, inside its namespaces): not a standalone header.


// ------------------------------------------------------------------------------------------
// Composition of two constraint domains on one set of nodes (reference networks/denoise_fn.py:287-291 the second
// encoder / decoder set, :310-311 which constraint types use it, :341-371 the zero column and the composing weights,
// :487-503 the second domain's inputs).  The reference loops over the types of both domains and scatter-adds every
// type's decoded outputs into one [N, P] sum with one count per node; a sum over types is the sum of the two domains'
// sums, so the composed evaluation is TWO ordinary evaluations -- each on its own model and graph, through the same three
// kernels as any other -- taken unnormalised, plus one elementwise kernel:
//     out = (w1 * S1 + w2 * widen(S2)) / sqrt(count1 + count2),   out[mask] = x[:, -P:][mask]
// widen() inserts the zero column (the pose coordinate the second domain does not know: z).  The second domain sees
// poses_2 = [poses[:, :2] | x[:, -(P2 - 2):]] (denoise_fn.py:499), built by k_compose_pack.
// ------------------------------------------------------------------------------------------
__global__ void k_compose_pack(int N, int P, int P2, const float* __restrict__ poses, const float* __restrict__ xfeat, int F, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * P2) return;
    const int n = i / P2, c = i % P2;
    out[i] = c < 2 ? poses[(size_t)n * P + c] : xfeat[(size_t)n * F + F - (P2 - c)];
}

__global__ void k_compose_outputs(int N, int P, int P2, int zero_col, const float* __restrict__ s1, const float* __restrict__ s2,
                                  const int* __restrict__ nptr1, const int* __restrict__ nptr2, float w1, float w2, int normalize,
                                  const signed char* __restrict__ mask, const float* __restrict__ xfeat, int F, float* __restrict__ out) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * P) return;
    const int n = i / P, c = i % P;
    float v = s1[i];
    if (w1 != 1.0f) v = v * w1;                                   // (denoise_fn.py:362-363: applied only when != 1)
    float u = 0.0f;
    if (c != zero_col) {
        u = s2[(size_t)n * P2 + (c < zero_col ? c : c - 1)];
        if (w2 != 1.0f) u = u * w2;
    }
    v = v + u;
    if (normalize) {
        const int cnt = (nptr1 ? nptr1[n + 1] - nptr1[n] : 0) + (nptr2 ? nptr2[n + 1] - nptr2[n] : 0);
        v = v / sqrtf((float)cnt);                                // 0/0 -> NaN like the reference (denoise_fn.py:523-524)
    }
    if (mask[n]) v = xfeat[(size_t)n * F + F - P + c];            // denoise_fn.py:531-532
    out[i] = v;
}

struct ComposeScratch { float *s1, *s2, *p2; };

int compose_energy_eval(ccsp_model* m1, ccsp_graph* g1, ccsp_model* m2, ccsp_graph* g2, const ccsp_compose* c, const float* poses_in, int t,
                        float* p_enc, float* p_tgt, float* E12, float* grad, float* energy, hipStream_t s);

// energy_ok: energy_wrapper models are accepted (their DIRECT evaluation is what forward(tag != 'EBM') returns, denoise_fn.py:535-537,
// and what a chain evaluates when both are energy models is decided by the caller)
int compose_check(const ccsp_model* m1, const ccsp_graph* g1, const ccsp_model* m2, const ccsp_graph* g2, const ccsp_compose* c, const char* who,
                  bool energy_ok = false) {
    if (!m1 || !g1 || !m2 || !g2 || !c) return fail("%s: null argument", who);
    if (g1->m != m1 || g2->m != m2) return fail("%s: a graph belongs to another model", who);
    if (m1->d.model_kind != CCSP_MODEL_DIFFUSION_CCSP || m2->d.model_kind != CCSP_MODEL_DIFFUSION_CCSP) return fail("%s: both domains must be Diffusion-CCSP models", who);
    if (!energy_ok && (m1->d.energy_wrapper || m2->d.energy_wrapper)) return fail("%s: composition is built for direct-mode (non energy_wrapper) models", who);
    if (m1->d.energy_wrapper != m2->d.energy_wrapper) return fail("%s: one domain is an energy_wrapper model and the other is not", who);
    if (m2->d.pose_dim + 1 != m1->d.pose_dim) return fail("%s: the second domain's pose_dim (%d) must be the first's (%d) minus the zero column", who, m2->d.pose_dim, m1->d.pose_dim);
    if (m2->d.pose_dim < 2 || g1->F < m2->d.pose_dim - 2) return fail("%s: bad second-domain pose layout", who);
    if (c->zero_col < 0 || c->zero_col >= m1->d.pose_dim) return fail("%s: zero_col=%d out of range", who, c->zero_col);
    if (g1->N != g2->N) return fail("%s: the two graphs have %d and %d nodes", who, g1->N, g2->N);
    if (m1->d.timesteps != m2->d.timesteps) return fail("%s: the two models have %d and %d timesteps", who, m1->d.timesteps, m2->d.timesteps);
    return 0;
}

// unnormalised sums of one domain at the pose state `poses` (nullptr = the graph's own state g->x, already encoded)
template <int H>
int compose_domain_sums(ccsp_model* m, ccsp_graph* g, const float* poses, int t, float* sums, hipStream_t s) {
    if (poses) {
        NodeArgs a = node_args(m, g);
        a.src = 2; a.step = STEP_NONE; a.do_encode = 1; a.x_in = poses;
        launch_node<H>(m, g, a, s);
    }
    if (launch_eval<H>(m, g, t, s)) return 1;
    NodeArgs b = node_args(m, g);
    b.src = 0; b.step = STEP_NONE; b.do_encode = 0; b.eps_out = sums; b.x_in = poses; b.normalize = 0;
    launch_node<H>(m, g, b, s);
    return 0;
}
int compose_domain_sums(ccsp_model* m, ccsp_graph* g, const float* poses, int t, float* sums, hipStream_t s) {
    return dispatch_h(m->d.hidden_dim, [&](auto hc) { return compose_domain_sums<decltype(hc)::value>(m, g, poses, t, sums, s); });
}

// one composed evaluation at `poses` (or at g1's state): result in `out` [N, P]
int compose_eval(ccsp_model* m1, ccsp_graph* g1, ccsp_model* m2, ccsp_graph* g2, const ccsp_compose* c, const float* poses, int t,
                 const ComposeScratch& w, float* out, hipStream_t s) {
    const int N = g1->N, P = m1->d.pose_dim, P2 = m2->d.pose_dim;
    if (compose_domain_sums(m1, g1, poses, t, w.s1, s)) return 1;
    hipLaunchKernelGGL(k_compose_pack, dim3(nblk((long)N * P2, 256)), dim3(256), 0, s, N, P, P2, poses ? poses : g1->x, g1->xfeat, g1->F, w.p2);
    if (compose_domain_sums(m2, g2, w.p2, t, w.s2, s)) return 1;
    hipLaunchKernelGGL(k_compose_outputs, dim3(nblk((long)N * P, 256)), dim3(256), 0, s, N, P, P2, c->zero_col, w.s1, w.s2,
                       g1->plan.E_act > 0 ? g1->node_ptr : (const int*)nullptr, g2->plan.E_act > 0 ? g2->node_ptr : (const int*)nullptr,
                       c->weight_first, c->weight_second, c->normalize, g1->mask, g1->xfeat, g1->F, out);
    return 0;
}


// composed energy (denoise_fn.py:373-375 on the composed outputs of :341-371): E = E1 + sum over second-domain entries of
// |widen(o2) - poses[node]|^2.  The widened output has a zero at zero_col, so that column contributes poses[n, zero_col]^2
// per entry; the other columns are the second model's own energy with the comparison target [poses without zero_col] while
// its encoder saw poses_2 (k_compose_pack) -- launch_eval_energy(..., x_enc, enc_cols = 2).
__global__ void k_compose_targets(int N, int P, int zero_col, const float* __restrict__ poses, float* __restrict__ out /*[N, P-1]*/) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * (P - 1)) return;
    const int n = i / (P - 1), c = i % (P - 1);
    out[i] = poses[(size_t)n * P + (c < zero_col ? c : c + 1)];
}

__global__ __launch_bounds__(256) void k_compose_energy(int N, int P, int zero_col, const float* __restrict__ poses, const float* __restrict__ g1,
                                                        const float* __restrict__ g2, const int* __restrict__ nptr2, const float* __restrict__ E12 /*[2]*/,
                                                        float* __restrict__ grad, float* __restrict__ energy) {
    // one workgroup: the batch is small next to the evaluations in front of it, and the energy is one ordered sum
    __shared__ float red[8];
    float e = 0.0f;
    for (int i = threadIdx.x; i < N * P; i += 256) {
        const int n = i / P, c = i % P;
        if (!grad) {                    // (uniform) energy only: the zero column's own term, summed in the same order
            if (c == zero_col) {
                const float cnt = nptr2 ? (float)(nptr2[n + 1] - nptr2[n]) : 0.0f;
                const float pz = poses[i];
                e += cnt * pz * pz;
            }
            continue;
        }
        float v = g1[i];
        if (c == zero_col) {
            const float cnt = nptr2 ? (float)(nptr2[n + 1] - nptr2[n]) : 0.0f;
            const float pz = poses[i];
            v += 2.0f * pz * cnt;
            e += cnt * pz * pz;
        } else {
            v += g2[(size_t)n * (P - 1) + (c < zero_col ? c : c - 1)];
        }
        grad[i] = v;
    }
    const float tot = block_sum_256(e, red);
    if (threadIdx.x == 0) energy[0] = (E12[0] + E12[1]) + tot;
}

