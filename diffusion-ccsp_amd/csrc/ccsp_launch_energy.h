// ccsp_launch_energy.h -- energy mode: energy_prepare (tables and workspaces) and launch_eval_energy (forward, hand-derived backward, node gradient + update).
// A fragment of the ONE translation unit csrc/ccsp_hip.hip (included there, at this position, inside its namespaces): not a standalone header.
// ---- energy mode -------------------------------------------------------------------------------
int energy_prepare(ccsp_model* m, ccsp_graph* g, hipStream_t s) {
    if (g->energy_ready) return 0;
    const ccsp::Plan& p = g->plan;
    const int H = m->d.hidden_dim, P = m->d.pose_dim, T = m->d.timesteps;
    auto& reg = g->allocs;
    if (dev_upload(reg, &g->e_a, p.e_a, s) || dev_upload(reg, &g->e_b, p.e_b, s) || dev_upload(reg, &g->row_ptr, p.row_ptr, s) ||
        dev_upload(reg, &g->row_edge, p.row_edge, s) || dev_upload(reg, &g->nrow_ptr, p.nrow_ptr, s) || dev_upload(reg, &g->nrow_idx, p.nrow_idx, s))
        return 1;
    // identity row tiles of the backward row GEMM (same tiles, rows taken as they are)
    if (dev_upload(reg, &g->tileb_row0, p.tile_row0, s) || dev_upload(reg, &g->tileb_nrows, p.tile_nrows, s) || dev_upload(reg, &g->tileb_ts, p.tile_ts, s)) return 1;
    const int BMf = dispatch_h(H, [](auto hc) { return 32 * EdgeCfg<decltype(hc)::value>::WM; });
    g->n_edge_blocks = 2 * nblk(p.E_act, BMf);
    const size_t n_partial = (size_t)(nblk(p.E_act, 16) > g->n_edge_blocks ? nblk(p.E_act, 16) : g->n_edge_blocks) + 1;   // (k_edge_h2s: one per 16 edges)
    // (with the row sums inside the decoder backward -- partial rows, below -- the per-edge gradient array, its fp32 row sums and the U-row
    // products are never written: 41 + 18 + 9 MB at C4 that are not allocated)
    const bool partial_rows = H == 256 && m->f16x2 && m->energy_bwd_h2 && m->WpTH && m->bwd_rowsum_fused && p.E_act > 0;
    if (!partial_rows && (dev_alloc(reg, &g->GZ, (size_t)p.E_act * 2 * H) || dev_alloc(reg, &g->GZR, (size_t)p.R * 2 * H) || dev_alloc(reg, &g->GP, (size_t)p.R * H)))
        return 1;
    if (dev_alloc(reg, &g->Q, (size_t)2 * p.E_act * (H / 2)) ||
        dev_alloc(reg, &g->xhat, (size_t)g->N * P) || dev_alloc(reg, &g->partial, n_partial) ||
        dev_alloc(reg, &g->Escal, 4) || dev_alloc(reg, &g->acc_count, (size_t)T) || dev_alloc(reg, &g->acc_denom, (size_t)T) ||
        dev_alloc(reg, &g->mala_changed, 3))                 // [0], [1] pose elements the accept step of an odd / even inner step moved, [2] evaluations skipped
        return 1;
    HIP_TRY(hipMemsetAsync(g->Escal, 0, 4 * sizeof(float), s));
    HIP_TRY(hipMemsetAsync(g->partial, 0, n_partial * sizeof(float), s));
    if (partial_rows) {
        // (the fused decoder kernel and its stand-alone backward work on 32-edge blocks, round 4's backward on 64-edge blocks)
        if (m->edge_fb > 0) ccsp::build_bwdsum_plan(p, TILE_M, g->bsplan, FB_EDGES, FB_MAXP);
        else ccsp::build_bwdsum_plan(p, TILE_M, g->bsplan);
        g->bs_fb = m->edge_fb;
        const ccsp::BwdSumPlan& b = g->bsplan;
        if (dev_upload(reg, &g->bs_blocks, b.blocks, s) || dev_upload(reg, &g->bs_nrow_ptr, b.nrow_ptr, s) || dev_upload(reg, &g->bs_nrow_idx, b.nrow_idx, s) ||
            dev_alloc(reg, &g->GZPH, (size_t)2 * b.NP * 2 * H) || dev_alloc(reg, &g->bs_gexp, (size_t)b.NP) || dev_alloc(reg, &g->GPP, (size_t)b.NP * H))
            return 1;
        g->h_bstd.clear();
        for (size_t i = 0; i < b.tile_row0.size(); ++i) g->h_bstd.push_back(make_int4(b.tile_row0[i], b.tile_nrows[i], b.tile_ts[i], 0));
        g->bs_tiles = (int)b.tile_row0.size();
        g->bs_tiles2 = 0;
        for (size_t i = 0; i < b.tile_row0.size();) {       // 128-row tiles: consecutive 64-row tiles of one (type, slot) group, two at a time
            const bool pair = i + 1 < b.tile_row0.size() && b.tile_ts[i + 1] == b.tile_ts[i] && b.tile_row0[i + 1] == b.tile_row0[i] + b.tile_nrows[i];
            g->h_bstd.push_back(make_int4(b.tile_row0[i], b.tile_nrows[i] + (pair ? b.tile_nrows[i + 1] : 0), b.tile_ts[i], 0));
            g->bs_tiles2++;
            i += pair ? 2 : 1;
        }
        int4* td = nullptr;
        if (dev_upload(reg, &td, g->h_bstd, s)) return 1;
        g->bs_td64 = td; g->bs_td128 = td + g->bs_tiles;
        g->bs_ready = true;
    }
    g->energy_ready = true;
    return 0;
}

// one energy-mode evaluation at `xeval` (pose embeddings of xeval must already be in g->pemb).
// with_grad: dE/dposes -> g->eps and E -> E_out;  otherwise only E -> E_out.
// E_out == nullptr (energy-only evaluations): leave the per-workgroup partials in g->partial / g->n_part_last for the consumer
template <int H>
int launch_eval_energy(ccsp_model* m, ccsp_graph* g, int t, const float* xeval, bool with_grad, float* E_out, hipStream_t s,
                       const int* skip = nullptr /*MALA reuse: every kernel of the evaluation returns at once if *skip == 0*/,
                       const float* x_enc = nullptr, int enc_cols = 0 /*composed domains: see EnergyNodeArgs*/,
                       const NodeArgs* tail = nullptr, bool* tail_done = nullptr /*the update that consumes the gradient: run in the last kernel if it can be*/) {
    const ccsp::Plan& p = g->plan;
    const int P = m->d.pose_dim;
    g->evals++;
    if (p.E_act == 0) {
        if (E_out) HIP_TRY(hipMemsetAsync(E_out, 0, sizeof(float), s));
        if (with_grad) HIP_TRY(hipMemsetAsync(g->eps, 0, (size_t)g->N * P * sizeof(float), s));
        return 0;
    }
    const int nw_u = g->n_tiles * rowgemm_col_tiles<H, 2 * H>();
    const float* tau_t = m->tau + (size_t)t * m->d.n_types * 2 * H;
    prof_mark(g, s, CCSP_K_ROWGEMM);
    bool h2 = false;
    if constexpr (H == 256) h2 = m->f16x2 != 0;
    if (h2) {            // the forward row GEMM is the direct-mode one (planes written by k_node)
        launch_rowgemm_h2(m, g, tau_t, StepRef{nullptr, nullptr, skip}, (size_t)0, s);
    } else if (m->bf16x3)
        hipLaunchKernelGGL((k_rowgemm_bf2<H, 2 * H>), dim3(g->n_tiles2 * (2 * H / RB2_TN)), dim3(512), 0, s, g->pembS, (size_t)g->N * H, g->urow_node,
                           g->t2_row0, g->t2_nrows, g->t2_ts, m->WpS, (size_t)m->d.n_types * 2 * 2 * H * H, (size_t)2 * H * H, g->base, tau_t, g->U,
                           StepRef{nullptr, nullptr}, (size_t)0);
    else
    hipLaunchKernelGGL((k_rowgemm<H, 2 * H>), dim3(nw_u < m->max_wgs ? nw_u : m->max_wgs), dim3(256), 0, s, nw_u, g->pemb, g->urow_node, g->tile_row0,
                       g->tile_nrows, g->tile_ts, m->Wp, (size_t)2 * H * H, g->base, tau_t, g->U);
    // round 6: a gradient evaluation on the f16x2 kernels runs the decoder forward and backward as ONE kernel (ccsp_edge_fb.h)
    const int fb = (h2 && with_grad && g->bs_ready && m->WpTH != nullptr && m->energy_bwd_h2) ? g->bs_fb : 0;
    prof_mark(g, s, fb == 2 ? CCSP_K_EDGE_FB : CCSP_K_EDGE);
    EdgeEnergyArgs en{g->e_a, g->e_b, xeval, (with_grad && fb != 2) ? g->Q : nullptr, g->partial, skip};
    int n_part = g->n_edge_blocks;                                                           // one energy partial per workgroup
    bool edge_done = false;
    bool bwd_done = false;
    if constexpr (H == 256) {
        if (fb == 2) {
            const EdgeFbArgs fa{m->Wd1THI, m->wd2_absmax, BwdSumArgs{g->bs_blocks, g->GZPH, (size_t)g->bsplan.NP * 2 * H, g->bs_gexp, m->bwd_bound_c}};
            n_part = nblk(p.E_act, FB_EDGES);
            if (P == 4) hipLaunchKernelGGL(k_edge_fb_h2<4>, dim3(n_part), dim3(256), 0, s, p.E_act, P, g->e_u0, g->e_u1, g->U, g->umax, m->Wd1HI, m->wd_exp,
                                           m->pd0_b, m->pd2_w, m->pd2_b, g->ent_pos, g->O, en, fa);
            else hipLaunchKernelGGL(k_edge_fb_h2<0>, dim3(n_part), dim3(256), 0, s, p.E_act, P, g->e_u0, g->e_u1, g->U, g->umax, m->Wd1HI, m->wd_exp,
                                    m->pd0_b, m->pd2_w, m->pd2_b, g->ent_pos, g->O, en, fa);
            edge_done = bwd_done = true;
        } else
        if (h2) {
            n_part = launch_edge_h2<true>(m, g, en, (int*)nullptr, s);
            edge_done = true;
        } else if (m->bf16x3 && m->edge_kernel == 2) {
            n_part = 2 * nblk(p.E_act, 64);
            hipLaunchKernelGGL(k_edge_bf2<true>, dim3(n_part), dim3(256), 0, s, p.E_act, P, g->e_u0, g->e_u1, g->U, m->Wd1S, m->pd0_b, m->pd2_w,
                               m->pd2_b, g->ent_pos, g->O, en, (int*)nullptr);
            edge_done = true;
        }
    }
    if (!edge_done)
    hipLaunchKernelGGL((k_edge<H, true>), dim3(n_part), dim3(256), 0, s, p.E_act, P, g->e_u0, g->e_u1, g->U, m->pd0_w,
                       m->pd0_b, m->pd2_w, m->pd2_b, g->ent_pos, g->O, en);
    g->n_part_last = n_part;
    if (!with_grad) {
        if (E_out) {
            prof_mark(g, s, CCSP_K_ENERGY_SUM);
            hipLaunchKernelGGL(k_energy_sum, dim3(1), dim3(256), 0, s, g->partial, n_part, E_out);
        }
        prof_mark(g, s, -1);
        return 0;
    }
    if (!bwd_done) prof_mark(g, s, CCSP_K_EDGE_BWD);
    constexpr int BMB = 32 * BwdCfg<H>::WM, NCTB = H / (32 * BwdCfg<H>::TN * BwdCfg<H>::WN);
    const bool h2_bwd = h2 && m->WpTH != nullptr && m->energy_bwd_h2;      // backward GEMMs on the f16x2 scheme as well
#ifndef CCSP_EXPERIMENTS
    // The product build's f16x2 backward forms the row sums inside the decoder backward (k_edge_bwd_h2<true>: partial rows, energy_prepare) and has
    // no other form compiled in: if the two conditions ever part ways the kernel would read a null plan and the transpose GEMM planes nobody wrote.
    if (h2_bwd && !g->bs_ready) return fail("energy mode: the f16x2 backward needs the partial-row plan (energy_prepare did not build it: hidden_dim %d, f16x2 %d, "
                                            "energy_bwd_h2 %d) -- set CCSP_ENERGY_BWD=bf16x3 or CCSP_MMA=f32 for this model", m->d.hidden_dim, m->f16x2, m->energy_bwd_h2);
#endif
    if constexpr (H == 256) {
        if (bwd_done) {}
        else if (fb == 1) {                 // the backward alone on the fused kernel's tiles (q from the Q array, go from the CSR slots)
            const EdgeFbArgs fa{m->Wd1THI, m->wd2_absmax, BwdSumArgs{g->bs_blocks, g->GZPH, (size_t)g->bsplan.NP * 2 * H, g->bs_gexp, m->bwd_bound_c}};
            if (P == 4) hipLaunchKernelGGL(k_edge_bwd2_h2<4>, dim3(nblk(p.E_act, FB_EDGES)), dim3(256), 0, s, p.E_act, P, g->e_u0, g->e_u1, g->ent_pos, g->U, g->O,
                                           g->Q, m->wd_exp, m->pd2_w, skip, fa);
            else hipLaunchKernelGGL(k_edge_bwd2_h2<0>, dim3(nblk(p.E_act, FB_EDGES)), dim3(256), 0, s, p.E_act, P, g->e_u0, g->e_u1, g->ent_pos, g->U, g->O,
                                    g->Q, m->wd_exp, m->pd2_w, skip, fa);
            bwd_done = true;
        } else if (h2_bwd) {
            const BwdSumArgs bsa = g->bs_ready ? BwdSumArgs{g->bs_blocks, g->GZPH, (size_t)g->bsplan.NP * 2 * H, g->bs_gexp, m->bwd_bound_c}
                                               : BwdSumArgs{nullptr, nullptr, 0, nullptr, 0.0f};
#define CCSP_EDGE_BWD(SUM, PP)                                                                                                                      \
            hipLaunchKernelGGL((k_edge_bwd_h2<SUM, PP>), dim3(nblk(p.E_act, 64) * 4), dim3(256), 0, s, p.E_act, P, g->e_u0, g->e_u1, g->ent_pos, g->U, g->O, \
                               g->Q, m->Wd1THI, m->wd_exp, m->wd2_absmax, m->pd2_w, g->GZ, skip, bsa)
            const bool p4 = P == 4 && !m->bwd_generic_p;
#ifdef CCSP_EXPERIMENTS
            if (!g->bs_ready) { if (p4) CCSP_EDGE_BWD(false, 4); else CCSP_EDGE_BWD(false, 0); }      // (CCSP_ENERGY_ROWSUM=kernel: round 3's k_rowsum_h2 downstream)
            else
#endif
            { if (p4) CCSP_EDGE_BWD(true, 4); else CCSP_EDGE_BWD(true, 0); }      // (bs_ready whenever these kernels run: energy_prepare)
#undef CCSP_EDGE_BWD
            bwd_done = true;
        } else if (m->bf16x3 && m->edge_kernel == 2) {
            hipLaunchKernelGGL(k_edge_bwd_bf, dim3(nblk(p.E_act, 64) * 4), dim3(256), 0, s, p.E_act, P, g->e_u0, g->e_u1, g->ent_pos, g->U, g->O, g->Q,
                               m->Wd1TS, m->pd2_w, g->GZ);
            bwd_done = true;
        }
    }
    if (!bwd_done)
    hipLaunchKernelGGL(k_edge_bwd<H>, dim3(nblk(p.E_act, BMB) * 2 * NCTB), dim3(256), 0, s, p.E_act, P, g->e_u0, g->e_u1, g->ent_pos,
                       g->U, g->O, g->Q, m->pd0_wT, m->pd2_w, g->GZ);
    const bool bf_bwd = !h2_bwd && H == 256 && m->bf16x3 && m->WpTS != nullptr;      // (the 128-column tiles need H >= 128)
    if (bf_bwd && !g->GZRS && dev_alloc(g->allocs, &g->GZRS, (size_t)3 * p.R * 2 * H)) return 1;
    const bool psum = h2_bwd && g->bs_ready;           // the row sums were formed by the decoder backward: partial rows from here on
#ifdef CCSP_EXPERIMENTS      // (CCSP_ENERGY_ROWSUM=kernel: round 3's k_rowsum_h2 and its fp16 planes of U-row sums)
    if (h2_bwd && !psum && !g->GZRH && (dev_alloc(g->allocs, &g->GZRH, (size_t)2 * p.R * 2 * H) || dev_alloc(g->allocs, &g->gexp, (size_t)p.R))) return 1;
#endif
    if (!psum) prof_mark(g, s, CCSP_K_ROWSUM);
    if (psum) {}
#ifdef CCSP_EXPERIMENTS
    else if (h2_bwd)
        hipLaunchKernelGGL(k_rowsum_h2, dim3(nblk(p.R, 4)), dim3(256), 0, s, p.R, g->row_ptr, g->row_edge, g->GZ, g->GZRH, g->gexp, skip);
#endif
    else
    hipLaunchKernelGGL(k_rowsum, dim3(nblk((long)p.R * (2 * H / 4), 256)), dim3(256), 0, s, p.R, 2 * H, g->row_ptr, g->row_edge, g->GZ, g->GZR,
                       bf_bwd ? g->GZRS : (unsigned short*)nullptr);
    const int* no_map = nullptr;
    const float* nof = nullptr;
    prof_mark(g, s, CCSP_K_ROWGEMM_T);
    if (h2_bwd) {
        if constexpr (H == 256) {       // g_p[row] = g_z[row] . Wp[type, slot]: the forward kernel with K = 2H, N = H, identity rows, no base
            const int mode = rowgemm_h2_mode(m, g, H / 128, psum ? g->bs_tiles : g->n_tiles);
            const bool small = mode == 4 || mode == 6;
            const int work = (small ? (psum ? g->bs_tiles : g->n_tiles) : (psum ? g->bs_tiles2 : g->n_tiles2)) * (H / 128);
            float* nou = nullptr;
            const unsigned short* a_pl = psum ? g->GZPH : g->GZRH;
            const size_t a_stride = (size_t)(psum ? g->bsplan.NP : p.R) * 2 * H;
            const int* a_ex = psum ? g->bs_gexp : g->gexp;
            const int4* tdesc = small ? (psum ? g->bs_td64 : g->td64) : (psum ? g->bs_td128 : g->td128);
            float* gp_out = psum ? g->GPP : g->GP;
#define CCSP_ROWGEMM_T(MODE)                                                                                                                        \
            hipLaunchKernelGGL((k_rowgemm_h2<2 * H, H, MODE>), dim3(work), dim3(256), 0, s, a_pl, a_stride, a_ex, no_map, tdesc, m->WpTHI,              \
                               (size_t)m->d.n_types * 2 * 2 * H * H, (size_t)2 * H * H, m->wp_exp, nof, nof, gp_out, nou, StepRef{nullptr, nullptr, skip}, \
                               (size_t)0)
            if (mode == 6) CCSP_ROWGEMM_T(6); else if (mode == 4) CCSP_ROWGEMM_T(4);
#ifdef CCSP_EXPERIMENTS
            else if (mode == 5) CCSP_ROWGEMM_T(5); else if (mode == 3) CCSP_ROWGEMM_T(3); else if (mode == 2) CCSP_ROWGEMM_T(2); else if (mode == 1) CCSP_ROWGEMM_T(1);
#endif
            else CCSP_ROWGEMM_T(0);
#undef CCSP_ROWGEMM_T
        }
    } else if (bf_bwd) {
        if constexpr (H == 256)
            hipLaunchKernelGGL((k_rowgemm_bf2<2 * H, H>), dim3(g->n_tiles2 * (H / RB2_TN)), dim3(512), 0, s, g->GZRS, (size_t)p.R * 2 * H, no_map,
                               g->t2_row0, g->t2_nrows, g->t2_ts, m->WpTS, (size_t)m->d.n_types * 2 * 2 * H * H, (size_t)2 * H * H, nof, nof, g->GP,
                               StepRef{nullptr, nullptr}, (size_t)0);
    } else {
    const int nw_b = g->n_tiles * rowgemm_col_tiles<2 * H, H>();
    hipLaunchKernelGGL((k_rowgemm<2 * H, H>), dim3(nw_b < m->max_wgs ? nw_b : m->max_wgs), dim3(256), 0, s, nw_b, g->GZR, no_map, g->tileb_row0,
                       g->tileb_nrows, g->tileb_ts, m->WpT, (size_t)2 * H * H, nof, nof, g->GP);
    }
    EnergyNodeArgs a{g->N, P, g->node_ptr, g->O, psum ? g->bs_nrow_ptr : g->nrow_ptr, psum ? g->bs_nrow_idx : g->nrow_idx, psum ? g->GPP : g->GP, xeval, g->eps,
                     g->partial, n_part, E_out,
                     m->pe0_w, m->pe0_b, m->pe2_w, m->pe2_wT, m->pe2_b, skip, x_enc, enc_cols, skip ? g->mala_changed + 2 : nullptr};
    if (tail_done) *tail_done = false;
    const bool valu_node_energy = m->valu_node_energy != 0 && 256 % H == 0;               // the pre-MFMA kernel, kept for A/B runs (widths that divide 256)
    prof_mark(g, s, CCSP_K_NODE_ENERGY);
#ifdef CCSP_EXPERIMENTS
    if (valu_node_energy) { if constexpr (256 % H == 0) hipLaunchKernelGGL(k_node_energy<H>, dim3(nblk(g->N, NODE_TILE)), dim3(256), 0, s, a); }
    else
#endif
    {
        bool h2n = false;
        if constexpr (H == 256) {
            h2n = m->pe2_wTH != nullptr;
            if (h2n && tail && m->node_energy_fused && m->pe2_wH && m->bf16x3 && m->f16x2 && m->d.model_kind == CCSP_MODEL_DIFFUSION_CCSP) {
                // (launch_node's EncOut for an energy_wrapper model on the f16x2 path: fp32 embeddings and the fp16 planes)
                EncOut eo;
                eo.f32 = g->pemb; eo.bf3 = nullptr; eo.h2 = g->pembH; eo.h2_exp = g->pexp;
                hipLaunchKernelGGL(k_node_energy_h2_update, dim3(nblk(g->N, NODE_TILE)), dim3(256), 0, s, a, enc_pose(m), (const unsigned short*)m->pe2_wTH, *tail,
                                   enc_pose(m), eo);
                *tail_done = true;
            } else if (h2n) hipLaunchKernelGGL(k_node_energy_h2, dim3(nblk(g->N, NODE_TILE)), dim3(256), 0, s, a, enc_pose(m), (const unsigned short*)m->pe2_wTH);
        }
        if (!h2n) hipLaunchKernelGGL(k_node_energy_mfma<H>, dim3(nblk(g->N, NODE_TILE)), dim3(256), 0, s, a, (const float*)m->pe2_wF);
    }
    prof_mark(g, s, -1);
    return 0;
}
