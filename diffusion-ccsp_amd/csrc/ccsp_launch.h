// ccsp_launch.h -- launch selection of one direct-mode evaluation: profiling marks, kernel-variant choice by tile count, launch_rowgemm_h2 / launch_edge_h2 / launch_node, launch_eval.
// A fragment of the ONE translation unit csrc/ccsp_hip.hip (included there, at this position, inside its namespaces): not a standalone header.

const char* const kKernelNames[CCSP_K_COUNT] = {"row GEMM (forward)", "edge decoder (forward)", "node update + pose encoder", "edge decoder backward",
                                               "row sum of g_z", "row GEMM (transpose)", "node energy backward", "energy sum", "HMC elementwise",
                                               "StructDiffusion evaluation", "fused evaluation (row GEMM + edge decoder)", "edge decoder forward + backward"};

inline void prof_mark(ccsp_graph* g, hipStream_t s, int id) {
    if (!g->profile || g->kev_used >= g->kev.size()) return;
    if (hipEventRecord(g->kev[g->kev_used], s) != hipSuccess) return;
    g->kev_id[g->kev_used++] = id;
}

template <typename T>
int dev_alloc(std::vector<void*>& reg, T** p, size_t n) {
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, (n ? n : 1) * sizeof(T)));
    reg.push_back(q);
    *p = (T*)q;
    return 0;
}

template <typename T>
int dev_upload(std::vector<void*>& reg, T** p, const std::vector<T>& v, hipStream_t s) {
    if (dev_alloc(reg, p, v.size())) return 1;
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return 0;
}

void cosine_betas(int T, std::vector<double>& betas) {   // ddpm.py:152-162
    const int steps = T + 1;
    const double s = 0.008;
    std::vector<double> ac(steps);
    for (int k = 0; k < steps; ++k) {
        const double xk = (double)k * (double)steps / (double)(steps - 1);
        const double c = cos(((xk / steps) + s) / (1 + s) * M_PI * 0.5);
        ac[k] = c * c;
    }
    const double a0 = ac[0];
    for (auto& v : ac) v /= a0;
    betas.resize(T);
    for (int t = 0; t < T; ++t) {
        const double b = 1 - ac[t + 1] / ac[t];
        betas[t] = b < 0 ? 0 : (b > 0.999 ? 0.999 : b);
    }
}

EncW enc_pose(const ccsp_model* m) { return EncW{m->pe0_w, m->pe0_b, m->pe2_wT, m->pe2_b, m->d.pose_dim, m->pe2_wF, m->pe2_wH, m->pe2_exp, m->pe0_c1, m->pe0_c2}; }

// f16x2 kernels (H = 256): the residency variant is chosen so that the whole tile list is resident at once when it can be
// (ccsp_f16x2.h): row GEMM 2 workgroups per CU with direct-to-LDS staging if the tiles fit, else 3 per CU; edge kernel
// 32-edge tiles at 3 per CU if they fit, else 64-edge tiles
// variant of k_rowgemm_h2 for a launch of `nct` column tiles per row tile (ccsp_f16x2.h): 64-row tiles on a ring of LDS stages
// when even those are at most one workgroup per CU (short tile lists are latency chains: C5 +12 %; with more work than that
// the 128-row forms win, C4 -1 % and C2 -4 % if forced), else 128-row tiles at 2 workgroups per CU with direct-to-LDS
// staging if they fit, else 3 per CU
int rowgemm_h2_mode(const ccsp_model* m, const ccsp_graph* g, int nct, int n_tiles = -1 /*64-row tiles; default: the graph's U-row tiles*/) {
    if (m->row_mode >= 0) return m->row_mode;
    if (n_tiles < 0) n_tiles = g->n_tiles;
    if (n_tiles * nct <= m->ncu) return 4;
    // round 3 (tools/ab_rowmode.sh, same-call A/B): with the straight-line epilogue the register-staged MODE 0 (three workgroups
    // per CU) is ahead of or equal to the direct-to-LDS MODE 2 at every size above the one-round limit -- C2's lanes 471-474
    // against 462, 128 graphs in one lane 287 against 275, 512 graphs 559 against 550, C4 +1 % -- so MODE 2 (and 1, 3) are only
    // reached through CCSP_ROW_MODE now.  Between the two, MODE 6 -- MODE 0's staging on 64-row tiles, four workgroups per CU -- while
    // its tile list still fits a bit more than two per CU (tools/ab_env.sh: 344 workgroups +3.4 %, 560 (C4) +1 %; 636 (a C2 lane) -3 %)
    if (n_tiles * nct <= 9 * m->ncu / 4) return 6;
    return 0;
}

int launch_rowgemm_h2(ccsp_model* m, ccsp_graph* g, const float* tau_t, StepRef ref, size_t tau_stride, hipStream_t s) {   // -> workgroups
    constexpr int H = 256;
    const int mode = rowgemm_h2_mode(m, g, 2 * H / 128);
#ifdef CCSP_EXPERIMENTS
    if (mode == 7 && m->WpF) {                          // resident A planes, weight fragments straight from global memory (ccsp_fused.h)
        if (ref.na.z) hipLaunchKernelGGL(k_noise_ahead, dim3(ref.na.blocks), dim3(256), 0, s, ref.na);
        hipLaunchKernelGGL(k_rowgemm_h2d, dim3(g->n_tiles * 4), dim3(256), 0, s, g->pembH, (size_t)g->N * H, g->pexp, g->urow_node, g->td64, m->WpF,
                           m->wp_exp, g->base, tau_t, g->U, g->umax, ref, tau_stride);
        return g->n_tiles * 4;
    }
#endif
    const bool small = mode == 4 || mode == 6;          // 64-row plan tiles instead of their 128-row pairs
    const int work = (small ? g->n_tiles : g->n_tiles2) * (2 * H / 128);
    ref.tile_rows = small ? g->tr64 : g->tr128;
#define CCSP_ROWGEMM_F(MODE)                                                                                                                          \
    hipLaunchKernelGGL((k_rowgemm_h2<H, 2 * H, MODE>), dim3(work + ref.na.blocks), dim3(256), 0, s, g->pembH, (size_t)g->N * H, g->pexp, g->urow_node,                  \
                       small ? g->td64 : g->td128, m->WpHI,                                                                                             \
                       (size_t)m->d.n_types * 2 * 2 * H * H, (size_t)2 * H * H, m->wp_exp, g->base, tau_t, g->U, g->umax, ref, tau_stride)
    if (mode == 6) CCSP_ROWGEMM_F(6); else if (mode == 4) CCSP_ROWGEMM_F(4);
#ifdef CCSP_TRY_MODE2
    else if (mode == 2) CCSP_ROWGEMM_F(2);
    else if (mode == 9) CCSP_ROWGEMM_F(9);
#endif
#ifdef CCSP_EXPERIMENTS
    else if (mode == 9) CCSP_ROWGEMM_F(9);
    else if (mode == 5) CCSP_ROWGEMM_F(5); else if (mode == 3) CCSP_ROWGEMM_F(3); else if (mode == 2) CCSP_ROWGEMM_F(2); else if (mode == 1) CCSP_ROWGEMM_F(1);
#endif
    else CCSP_ROWGEMM_F(0);
#undef CCSP_ROWGEMM_F
    return work;
}

// edges per workgroup of the f16x2 edge kernel for a batch of E_act active edges: 16 (k_edge_h2s) when most of the chip would
// idle even then, else 32 if the tiles then fit three per CU, else 64
int edge_tile_edges(const ccsp_model* m, int E_act) {
    if (m->edge_small > 0 || (m->edge_small < 0 && m->edge_mt <= 0 && nblk(E_act, 16) <= m->ncu)) return 16;
    const int mt = m->edge_mt > 0 ? m->edge_mt : (nblk(E_act, 32) <= 3 * m->ncu ? 1 : 2);
    return 32 * mt;
}

// returns the number of workgroups (= energy partials).  fu: fold the node update into the kernel's tail (direct mode)
template <bool ENERGY>
int launch_edge_h2(ccsp_model* m, ccsp_graph* g, EdgeEnergyArgs en, int* cinc, hipStream_t s, const FuseArgs* fu = nullptr) {
    const int E_act = g->plan.E_act;
    const int me = edge_tile_edges(m, E_act);
    FuseArgs f0;
    memset(&f0, 0, sizeof(f0));
    // (the fused forms hold the node update's registers: two workgroups per CU, so only for tile lists that fit that)
#ifdef CCSP_EXPERIMENTS
    const bool fuse = !ENERGY && fu != nullptr && me == g->fuse_me && nblk(E_act, me) <= 2 * m->ncu;
#else
    constexpr bool fuse = false;        // (the node update in the edge kernel's tail: an experiment, slower -- DESIGN.md 9)
    (void)fu;
#endif
    const FuseArgs& fa = fuse ? *fu : f0;
    if (me == 16) {
        const int nws = nblk(E_act, 16);
#define CCSP_EDGE_S(FUSE)                                                                                                                            \
        hipLaunchKernelGGL((k_edge_h2s<ENERGY, FUSE>), dim3(nws), dim3(256), 0, s, E_act, m->d.pose_dim, FUSE ? g->fuse_u0 : g->e_u0,                     \
                           FUSE ? g->fuse_u1 : g->e_u1, g->U, g->umax, m->Wd1HI, m->wd_exp, m->pd0_b, m->pd2_w, m->pd2_b, FUSE ? g->fuse_pos : g->ent_pos, \
                           g->O, en, cinc, fa)
#ifdef CCSP_EXPERIMENTS
        if constexpr (!ENERGY) { if (fuse) CCSP_EDGE_S(true); else CCSP_EDGE_S(false); }
        else CCSP_EDGE_S(false);
#else
        CCSP_EDGE_S(false);
#endif
#undef CCSP_EDGE_S
        return nws;
    }
    const int mt = me / 32;
    const int nwg = nblk(E_act, me);
#define CCSP_EDGE_F(MT, L2, FUSE)                                                                                                                    \
    hipLaunchKernelGGL((k_edge_h2<ENERGY, MT, L2, FUSE>), dim3(nwg), dim3(256), 0, s, E_act, m->d.pose_dim, FUSE ? g->fuse_u0 : g->e_u0,                \
                       FUSE ? g->fuse_u1 : g->e_u1, g->U, g->umax, m->Wd1HI, m->wd_exp, m->pd0_b, m->pd2_w, m->pd2_b, FUSE ? g->fuse_pos : g->ent_pos,     \
                       g->O, en, cinc, fa)
#ifdef CCSP_EXPERIMENTS
    if constexpr (!ENERGY) {
        if (fuse && mt == 1) {
            if (nwg <= m->ncu) CCSP_EDGE_F(1, 1, true); else CCSP_EDGE_F(1, 0, true);
            return nwg;
        }
    }
#endif
    if (mt == 1 && nwg <= m->ncu) CCSP_EDGE_F(1, 1, false);  // a single round of workgroups: the short-latency second layer
    else if (mt == 1) CCSP_EDGE_F(1, 0, false);
    else CCSP_EDGE_F(2, 0, false);
#undef CCSP_EDGE_F
    return nwg;
}

#ifdef CCSP_EXPERIMENTS
// Tables of the fused node update for edge tiles of `me` edges.  The edge kernel may take the edges in any order (the decoder is
// shared by all types; every output goes to its own CSR slot), so the fused form walks them NODE-BLOCK-major instead of
// type-major: a tile's outputs then land in one or two 16-node blocks and a block is completed by the few neighbouring tiles
// that feed it -- in the middle of the launch, on many different workgroups.  (Type-major order made the last type's tiles the
// last arrivers of nearly every block: a handful of workgroups ran all the node blocks one after the other, 85 us per launch.)
// Uploads: the permuted edge tables, the blocks each tile touches, the tiles per block.
int fuse_prepare(ccsp_model* m, ccsp_graph* g, int me, hipStream_t s) {
    if (g->fuse_me == me) return 0;
    const ccsp::Plan& p = g->plan;
    const int n_wg = nblk(p.E_act, me), n_blk = nblk(g->N, NODE_TILE);
    std::vector<int> perm(p.E_act);
    for (int k = 0; k < p.E_act; ++k) perm[k] = k;
    std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) {
        const int bx = (p.e_a[x] < p.e_b[x] ? p.e_a[x] : p.e_b[x]) / NODE_TILE, by = (p.e_a[y] < p.e_b[y] ? p.e_a[y] : p.e_b[y]) / NODE_TILE;
        return bx < by;
    });
    std::vector<std::vector<int>> per_wg(n_wg);
    std::vector<int> expect(n_blk, 0), stamp(n_blk, -1);
    for (int w = 0; w < n_wg; ++w) {
        for (int j = w * me; j < (w + 1) * me && j < p.E_act; ++j)
            for (int b : {p.e_a[perm[j]] / NODE_TILE, p.e_b[perm[j]] / NODE_TILE})
                if (stamp[b] != w) { stamp[b] = w; per_wg[w].push_back(b); expect[b]++; }
    }
    for (int b = 0; b < n_blk; ++b)                         // blocks no edge reaches (isolated nodes): their update still has to run
        if (expect[b] == 0) { per_wg[b % n_wg].push_back(b); expect[b] = 1; }
    std::vector<int> ptr(n_wg + 1, 0), list;
    for (int w = 0; w < n_wg; ++w) {
        std::sort(per_wg[w].begin(), per_wg[w].end());
        list.insert(list.end(), per_wg[w].begin(), per_wg[w].end());
        ptr[w + 1] = (int)list.size();
    }
    std::vector<int> pu0(p.E_act), pu1(p.E_act), ppos((size_t)2 * p.E_act);
    for (int j = 0; j < p.E_act; ++j) {
        pu0[j] = p.e_u0[perm[j]]; pu1[j] = p.e_u1[perm[j]];
        ppos[2 * j] = p.ent_pos[2 * perm[j]]; ppos[2 * j + 1] = p.ent_pos[2 * perm[j] + 1];
    }
    HIP_TRY(hipStreamSynchronize(s));                       // (a previous upload may still be reading h_fuse)
    g->h_fuse = ptr;
    g->h_fuse.insert(g->h_fuse.end(), list.begin(), list.end());
    g->h_fuse.insert(g->h_fuse.end(), expect.begin(), expect.end());
    g->h_fuse.insert(g->h_fuse.end(), pu0.begin(), pu0.end());
    g->h_fuse.insert(g->h_fuse.end(), pu1.begin(), pu1.end());
    g->h_fuse.insert(g->h_fuse.end(), ppos.begin(), ppos.end());
    int* d = nullptr;
    if (dev_upload(g->allocs, &d, g->h_fuse, s)) return 1;
    g->fuse_ptr = d; g->fuse_list = d + ptr.size(); g->fuse_expect = g->fuse_list + list.size();
    g->fuse_u0 = g->fuse_expect + n_blk; g->fuse_u1 = g->fuse_u0 + p.E_act; g->fuse_pos = g->fuse_u1 + p.E_act;
    if (!g->fuse_count || g->fuse_blocks != n_blk) { if (dev_alloc(g->allocs, &g->fuse_count, (size_t)n_blk)) return 1; }
    g->fuse_blocks = n_blk;
    g->fuse_me = me;
    return 0;
}

// Tables of the node-grouped edge kernel (k_edge_h2<.., NG>): consecutive nodes are packed into workgroups while their CSR entries fit
// 64 rows (and the nodes one 16-node encoder tile); row r of a workgroup is CSR entry csr0 + r, i.e. (edge k, half s) with 2k + s =
// node_ent[csr0 + r], and carries the element offsets of its two U operands.  Nodes without entries ride along (their update still runs).
int fuse2_prepare(ccsp_model* m, ccsp_graph* g, hipStream_t s) {
    if (g->ng_wgs != 0) return 0;
    const ccsp::Plan& p = g->plan;
    const int H = m->d.hidden_dim;
    std::vector<int> desc, off0, off1;
    int n = 0;
    while (n < g->N) {
        const int n0 = n, c0 = p.node_ptr[n];
        while (n < g->N && n - n0 < NODE_TILE && p.node_ptr[n + 1] - c0 <= 64) ++n;
        if (n == n0) { g->ng_wgs = -1; return 0; }            // a node with more than 64 entries: this graph keeps the separate node kernel
        const int rows = p.node_ptr[n] - c0;
        desc.push_back(n0); desc.push_back(n - n0); desc.push_back(c0); desc.push_back(rows);
        for (int r = 0; r < 64; ++r) {
            int q = c0 + (r < rows ? r : 0);                               // (padding rows repeat row 0: valid addresses, outputs never stored;
            q = q < 2 * p.E_act ? q : 2 * p.E_act - 1;                     //  a workgroup of entry-less nodes reads a later node's first entry, or the last entry)
            const int ent = p.node_ent[q];
            const int k = ent >> 1, half = ent & 1;
            off0.push_back(p.e_u0[k] * 2 * H + half * H);
            off1.push_back(p.e_u1[k] * 2 * H + half * H);
        }
    }
    const int n_wg = (int)desc.size() / 4;
    HIP_TRY(hipStreamSynchronize(s));
    g->h_ng = desc;
    g->h_ng.insert(g->h_ng.end(), off0.begin(), off0.end());
    g->h_ng.insert(g->h_ng.end(), off1.begin(), off1.end());
    int* d = nullptr;
    if (dev_upload(g->allocs, &d, g->h_ng, s)) return 1;
    g->ng_desc = reinterpret_cast<int4*>(d);
    g->ng_off0 = d + desc.size(); g->ng_off1 = g->ng_off0 + off0.size();
    g->ng_wgs = n_wg;
    return 0;
}

#endif  // CCSP_EXPERIMENTS

// fused: if non-null (direct-mode chain, f16x2 kernels), the node update with these arguments is folded into the edge kernel's
// tail and *did_fuse is set; the caller then launches no node kernel
template <int H>
int launch_eval(ccsp_model* m, ccsp_graph* g, int t, hipStream_t s, bool tabled = false, const NodeArgs* fused = nullptr, bool* did_fuse = nullptr,
                const NoiseAhead* na = nullptr /*H = 256, f16x2 only: the evaluation's normal draws, see NoiseAhead*/) {
    // U = pose_emb . Wp^T ; O = decoder(...)
    // tabled (hipGraph mode): the timestep comes from the device step table, see StepEntry
    const ccsp::Plan& p = g->plan;
    if (p.E_act == 0) return 0;
    prof_mark(g, s, CCSP_K_ROWGEMM);
    const int nw_u = g->n_tiles * rowgemm_col_tiles<H, 2 * H>();
    const size_t tau_stride = (size_t)m->d.n_types * 2 * H;
    const float* tau_t = m->tau + (tabled ? 0 : (size_t)t * tau_stride);
    StepRef ref{tabled ? g->d_tab : nullptr, tabled ? g->d_counter : nullptr};
    if (na) ref.na = *na;
    int* const cinc = tabled ? g->d_counter : nullptr;
    if constexpr (H == 256) {
#ifdef CCSP_EXPERIMENTS
        if (m->f16x2 && m->eval_fused && !tabled && g->n_ftiles > 0 && fused == nullptr) {
            prof_mark(g, s, CCSP_K_EVAL_FUSED);
            if (ref.na.z) hipLaunchKernelGGL(k_noise_ahead, dim3(ref.na.blocks), dim3(256), 0, s, ref.na);
            FusedArgs fa;
            fa.order = g->ft_order; fa.n_items = 2 * g->n_ftiles;
            fa.tiles = g->ft_tiles; fa.rows = g->ft_rows; fa.e_lu = g->ft_elu; fa.ent_pos = g->ent_pos;
            fa.A = g->pembH; fa.a_plane = (size_t)g->N * H; fa.a_exp = g->pexp;
            fa.WpF = m->WpF; fa.w_exp = m->wp_exp; fa.base = g->base; fa.tau_t = tau_t;
            fa.Wd1F = m->Wd1F; fa.wd_exp = m->wd_exp; fa.bd1 = m->pd0_b; fa.Wd2 = m->pd2_w; fa.bd2 = m->pd2_b;
            fa.O = g->O; fa.P = m->d.pose_dim;
            if (m->eval_fused == 1) hipLaunchKernelGGL(k_eval_fused4, dim3(fa.n_items), dim3(256), 0, s, fa);
            else hipLaunchKernelGGL(k_eval_fused, dim3(fa.n_items < m->ncu ? fa.n_items : m->ncu), dim3(512), 0, s, fa);
            if (did_fuse) *did_fuse = false;
            prof_mark(g, s, -1);
            g->evals++;
            return 0;
        }
#endif
        if (m->f16x2) {
            launch_rowgemm_h2(m, g, tau_t, ref, tau_stride, s);
            prof_mark(g, s, CCSP_K_EDGE);
#ifndef CCSP_EXPERIMENTS
            (void)fused;
            launch_edge_h2<false>(m, g, EdgeEnergyArgs{}, cinc, s, nullptr);
            if (did_fuse) *did_fuse = false;
#else
            FuseArgs fu;
            if (fused != nullptr && g->ng_use && g->ng_wgs > 0) {          // node-grouped edge tiles with the node update as their tail
                memset(&fu, 0, sizeof(fu));
                fu.ng_desc = g->ng_desc; fu.ng_off0 = g->ng_off0; fu.ng_off1 = g->ng_off1;
                fu.node = *fused; fu.w = enc_pose(m);
                fu.eo.f32 = nullptr; fu.eo.bf3 = nullptr; fu.eo.h2 = g->pembH; fu.eo.h2_exp = g->pexp;
                const EdgeEnergyArgs en0{};
                if (nblk(p.E_act, 32) <= m->ncu)      // (the second decoder layer in the form the three-launch path picks for this batch: same sums, bit for bit)
                    hipLaunchKernelGGL((k_edge_h2<false, 1, 1, false, true>), dim3(g->ng_wgs), dim3(256), 0, s, p.E_act, m->d.pose_dim, g->e_u0, g->e_u1, g->U, g->umax,
                                       m->Wd1HI, m->wd_exp, m->pd0_b, m->pd2_w, m->pd2_b, g->ent_pos, g->O, en0, cinc, fu);
                else
                    hipLaunchKernelGGL((k_edge_h2<false, 1, 0, false, true>), dim3(g->ng_wgs), dim3(256), 0, s, p.E_act, m->d.pose_dim, g->e_u0, g->e_u1, g->U, g->umax,
                                       m->Wd1HI, m->wd_exp, m->pd0_b, m->pd2_w, m->pd2_b, g->ent_pos, g->O, en0, cinc, fu);
                if (did_fuse) *did_fuse = true;
                prof_mark(g, s, -1);
                g->evals++;
                return 0;
            }
            const bool fuse = fused != nullptr && g->fuse_me > 0 && g->fuse_me == edge_tile_edges(m, p.E_act) && g->fuse_me <= 32 &&
                              nblk(p.E_act, g->fuse_me) <= 2 * m->ncu;
            if (fuse) {
                memset(&fu, 0, sizeof(fu));
                fu.wg_blk_ptr = g->fuse_ptr; fu.wg_blk = g->fuse_list; fu.blk_expect = g->fuse_expect; fu.blk_count = g->fuse_count;
                fu.epoch = ++g->fuse_epoch; fu.n_ent = 2 * p.E_act;
                fu.node = *fused; fu.w = enc_pose(m);
                fu.eo.f32 = nullptr; fu.eo.bf3 = nullptr; fu.eo.h2 = g->pembH; fu.eo.h2_exp = g->pexp;
            }
            launch_edge_h2<false>(m, g, EdgeEnergyArgs{}, cinc, s, fuse ? &fu : nullptr);
            if (did_fuse) *did_fuse = fuse;
#endif
            prof_mark(g, s, -1);
            g->evals++;
            return 0;
        }
    }
    if (m->bf16x3) {
        const long npe = (long)g->N * H;
        if (m->row_tile == 128)
            hipLaunchKernelGGL((k_rowgemm_bf2<H, 2 * H>), dim3(g->n_tiles2 * (2 * H / RB2_TN)), dim3(512), 0, s, g->pembS, (size_t)npe, g->urow_node,
                               g->t2_row0, g->t2_nrows, g->t2_ts, m->WpS, (size_t)m->d.n_types * 2 * 2 * H * H, (size_t)2 * H * H, g->base, tau_t, g->U,
                               ref, tau_stride);
        else
        hipLaunchKernelGGL((k_rowgemm_bf<H, 2 * H>), dim3(nw_u), dim3(256), 0, s, g->pembS, (size_t)npe, g->urow_node, g->tile_row0,
                           g->tile_nrows, g->tile_ts, m->WpS, (size_t)m->d.n_types * 2 * 2 * H * H, (size_t)2 * H * H, g->base, tau_t, g->U,
                           ref, tau_stride);
        prof_mark(g, s, CCSP_K_EDGE);
        constexpr int BMB = 32 * EdgeBfCfg<H>::WM;
        if constexpr (H == 256) {
            if (m->edge_kernel == 2) {
                hipLaunchKernelGGL(k_edge_bf2<false>, dim3(2 * nblk(p.E_act, 64)), dim3(256), 0, s, p.E_act, m->d.pose_dim, g->e_u0, g->e_u1, g->U,
                                   m->Wd1S, m->pd0_b, m->pd2_w, m->pd2_b, g->ent_pos, g->O, EdgeEnergyArgs{}, cinc);
                prof_mark(g, s, -1);
                g->evals++;
                return 0;
            }
        }
        hipLaunchKernelGGL(k_edge_bf<H>, dim3(2 * nblk(p.E_act, BMB)), dim3(256), 0, s, p.E_act, m->d.pose_dim, g->e_u0, g->e_u1, g->U,
                           m->Wd1S, m->pd0_b, m->pd2_w, m->pd2_b, g->ent_pos, g->O, cinc);
    } else {
    hipLaunchKernelGGL((k_rowgemm<H, 2 * H>), dim3(nw_u < m->max_wgs ? nw_u : m->max_wgs), dim3(256), 0, s, nw_u, g->pemb, g->urow_node, g->tile_row0,
                       g->tile_nrows, g->tile_ts, m->Wp, (size_t)2 * H * H, g->base, tau_t, g->U);
    prof_mark(g, s, CCSP_K_EDGE);
    constexpr int BM = 32 * EdgeCfg<H>::WM;
    const int nw_e = 2 * nblk(p.E_act, BM);
    hipLaunchKernelGGL((k_edge<H, false>), dim3(nw_e), dim3(256), 0, s, p.E_act, m->d.pose_dim, g->e_u0,
                       g->e_u1, g->U, m->pd0_w, m->pd0_b, m->pd2_w, m->pd2_b, g->ent_pos, g->O, EdgeEnergyArgs{});
    }
    prof_mark(g, s, -1);
    g->evals++;
    return 0;
}

NodeArgs node_args(ccsp_model* m, ccsp_graph* g) {
    NodeArgs a;
    memset(&a, 0, sizeof(a));
    a.N = g->N; a.P = m->d.pose_dim; a.F = g->F;
    a.normalize = m->d.normalize;
    a.node_ptr = g->node_ptr; a.O = g->O;
    a.xfeat = g->xfeat; a.pose_begin = m->d.pose_begin; a.mask = g->mask;
    a.x = g->x;
    return a;
}

template <int H>
void launch_node(ccsp_model* m, ccsp_graph* g, const NodeArgs& a, hipStream_t s) {
    // direct-mode bf16x3 evaluations read the planes only; the fp32 embeddings are for the fp32 / energy / transformer paths
    const bool planes = m->bf16x3 && m->d.model_kind == CCSP_MODEL_DIFFUSION_CCSP;
    const bool h2 = planes && H == 256 && m->f16x2;
    EncOut eo;
    eo.f32 = (!planes || m->d.energy_wrapper) ? g->pemb : nullptr;
    eo.bf3 = (planes && !h2) ? g->pembS : nullptr;
    eo.h2 = h2 ? g->pembH : nullptr;
    eo.h2_exp = h2 ? g->pexp : nullptr;
    prof_mark(g, s, CCSP_K_NODE);
    bool ench = false;
    if constexpr (H == 256) ench = m->pe2_wH != nullptr;
    if constexpr (H == 256) {
        // the straight-line form of the hot case (see k_node_direct); CCSP_NODE=generic keeps k_node for A/B runs
        const bool direct = ench && h2 && !m->node_generic && a.src == 0 && (a.step == STEP_ANCESTRAL || a.step == STEP_ULA) && a.do_encode &&
                            !a.x_in && !a.eps_out && !a.tab && g->plan.E_act > 0 && !eo.f32;
        if (direct) {
#ifdef CCSP_EXPERIMENTS
            if (m->node_stream) hipLaunchKernelGGL(k_node_direct_s, dim3(nblk(g->N, NODE_TILE)), dim3(256), 0, s, a, enc_pose(m), eo, 2 * g->plan.E_act);
            else
#endif
            hipLaunchKernelGGL(k_node_direct, dim3(nblk(g->N, NODE_TILE)), dim3(256), 0, s, a, enc_pose(m), eo, 2 * g->plan.E_act);
            prof_mark(g, s, -1);
            return;
        }
        if (ench) hipLaunchKernelGGL((k_node<H, true>), dim3(nblk(g->N, NODE_TILE)), dim3(256), 0, s, a, enc_pose(m), eo);
    }
    if (!ench) hipLaunchKernelGGL((k_node<H, false>), dim3(nblk(g->N, NODE_TILE)), dim3(256), 0, s, a, enc_pose(m), eo);
    prof_mark(g, s, -1);
}

