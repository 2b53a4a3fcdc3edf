// ccsp_abi_compose.h -- C ABI, part 4: composed domains (ccsp_compose_*), the host-only planners (ccsp_plan_*_host).
// A fragment of the ONE translation unit csrc/ccsp_hip.hip (included there, at this position, inside its namespaces): not a standalone header.
namespace {
// the composed energy and its gradient at poses_in (the body of ccsp_compose_energy_grad; also one evaluation of an energy-mode
// chain of a composed model, ccsp_compose_chain_run).  p_enc / p_tgt: [N, P2] scratch, E12: 2 floats of scratch
// grad == nullptr: the energy only (forward passes of both domains, no backward: MALA's evaluation at the proposal, HMC's two energies per inner step)
int compose_energy_eval(ccsp_model* m1, ccsp_graph* g1, ccsp_model* m2, ccsp_graph* g2, const ccsp_compose* c, const float* poses_in, int t,
                        float* p_enc, float* p_tgt, float* E12, float* grad, float* energy, hipStream_t s) {
    const bool with_grad = grad != nullptr;
    const int N = g1->N, P = m1->d.pose_dim, P2 = m2->d.pose_dim;
    hipLaunchKernelGGL(k_compose_pack, dim3(nblk((long)N * P2, 256)), dim3(256), 0, s, N, P, P2, poses_in, g1->xfeat, g1->F, p_enc);
    hipLaunchKernelGGL(k_compose_targets, dim3(nblk((long)N * P2, 256)), dim3(256), 0, s, N, P, c->zero_col, poses_in, p_tgt);
    const int rc = dispatch_h(m1->d.hidden_dim, [&](auto hc) {
        constexpr int HH = decltype(hc)::value;
        NodeArgs a = node_args(m1, g1);
        a.src = 2; a.step = STEP_NONE; a.do_encode = 1; a.x_in = poses_in;
        launch_node<HH>(m1, g1, a, s);
        if (launch_eval_energy<HH>(m1, g1, t, poses_in, with_grad, E12, s)) return 1;
        NodeArgs b = node_args(m2, g2);
        b.src = 2; b.step = STEP_NONE; b.do_encode = 1; b.x_in = p_enc;
        launch_node<HH>(m2, g2, b, s);
        return launch_eval_energy<HH>(m2, g2, t, p_tgt, with_grad, E12 + 1, s, nullptr, p_enc, 2);
    });
    if (rc) return 1;
    hipLaunchKernelGGL(k_compose_energy, dim3(1), dim3(256), 0, s, N, P, c->zero_col, poses_in, g1->eps, g2->eps,
                       g2->plan.E_act > 0 ? g2->node_ptr : (const int*)nullptr, E12, grad, energy);
    return 0;
}
}  // namespace

extern "C" {

int ccsp_compose_chain_run(ccsp_model* m1, ccsp_graph* g1, ccsp_model* m2, ccsp_graph* g2, const ccsp_compose* c, int32_t sampler,
                           const ccsp_noise* nz, float* x, int32_t init, int32_t t_first, int32_t t_last, float* history, float* accept, void* stream) {
    if (compose_check(m1, g1, m2, g2, c, "compose_chain_run", true)) return 1;
    if (!nz || !x) return fail("compose_chain_run: null argument");
    ccsp_model* m = m1;
    ccsp_graph* g = g1;
    const int T = m->d.timesteps, P = m->d.pose_dim;
    const bool hmc = sampler == CCSP_SAMPLER_HMC;
    const bool mala = sampler == CCSP_SAMPLER_MALA || hmc;          // (what the two Metropolis samplers share: acceptance counters, uniform draws)
    if (sampler != CCSP_SAMPLER_NONE && sampler != CCSP_SAMPLER_ULA && sampler != CCSP_SAMPLER_ULA_PLUS && !(mala && m1->d.energy_wrapper))
        return fail("compose_chain_run: sampler %d: composed models run the ancestral / ULA / ULA+ samplers (on the denoiser output, or on the energy gradient "
                    "when both are energy_wrapper models) and, as energy_wrapper models, MALA and HMC", sampler);
    if (hmc && m1->d.timesteps < 4) return fail("compose_chain_run: HMC indexes the schedule with its inner step 0..3 (ddpm.py:1076-1084)");
    if (hmc && (m1->energy_hook || m1->rccl_comm || m2->energy_hook || m2->rccl_comm))
        return fail("compose_chain_run: a shard energy hook / communicator is installed, but the HMC chain does not reduce its energies across shards "
                    "(only MALA does): the shards would silently decouple -- remove it (ccsp_model_set_energy_hook(model, NULL, NULL)) or run MALA");
    if (sampler == CCSP_SAMPLER_MALA && (m2->energy_hook || m2->rccl_comm) && !(m1->energy_hook || m1->rccl_comm))
        return fail("compose_chain_run: the shard energy hook / communicator must be installed on the FIRST domain's model (the one whose chain this is)");
    // energy mode (both energy_wrapper models; ComposedEBMDenoiseFn.forward: epsilon = dE/dposes, ddpm.py:940-966 on it): every evaluation
    // is the composed energy gradient of ccsp_compose_energy_grad
    const bool energy = m1->d.energy_wrapper != 0;
    if (energy) {
        if (c->zero_col < 2) return fail("compose_chain_run: zero_col=%d (the second domain's encoder takes pose columns 0 and 1)", c->zero_col);
        if (c->weight_first != 1.0f || c->weight_second != 1.0f) return fail("compose_chain_run: composing weights other than (1, 1) are built for the direct mode only");
        if (m1->d.hidden_dim != m2->d.hidden_dim) return fail("compose_chain_run: the two domains differ in hidden_dim");
    }
    if (t_first >= T || t_last < 0 || t_first < t_last - 1) return fail("compose_chain_run: bad timestep range [%d,%d]", t_first, t_last);
    if (nz->mode != CCSP_NOISE_PHILOX && nz->mode != CCSP_NOISE_INJECTED) return fail("compose_chain_run: unknown noise mode %d", nz->mode);
    if (nz->mode == CCSP_NOISE_INJECTED && !nz->normal) return fail("compose_chain_run: injected noise without a normal stream");
    hipStream_t s = (hipStream_t)stream;
    const size_t N = (size_t)g->N, NP = N * P;
    StreamBuf b1(s), b2(s), b3(s), b4(s);
    if (b1.alloc(NP * sizeof(float)) || b2.alloc(N * m2->d.pose_dim * sizeof(float)) || b3.alloc(N * m2->d.pose_dim * sizeof(float)) ||
        b4.alloc(6 * sizeof(float))) return 1;
    if (mala && nz->mode == CCSP_NOISE_INJECTED && !nz->uniform) return fail("compose_chain_run: MALA with injected noise needs a uniform stream");
    const ComposeScratch w{b1.f(), b2.f(), b3.f()};
    if (energy && (energy_prepare(m1, g1, s) || energy_prepare(m2, g2, s))) return 1;
    std::vector<uint64_t> call0(T), ucall0(T, 0);
    {
        uint64_t k = 1, u = 0;
        for (int t = T - 1; t >= 0; --t) {      // (HMC draws the momentum once per timestep on top of its S refreshments, ddpm.py:1090,1096)
            const uint64_t S = (uint64_t)steps_at(m, sampler, t);
            call0[t] = k; ucall0[t] = u; k += 1 + S + (hmc && S > 0 ? 1 : 0); u += S;
        }
    }
    if (mala) {         // acceptance counters of the first domain's graph (energy_prepare below allocates them)
        if (energy_prepare(m1, g1, s)) return 1;
        HIP_TRY(hipMemsetAsync(g1->acc_count, 0, (size_t)T * sizeof(int), s));
        HIP_TRY(hipStreamSynchronize(s));      // (a previous chain may still be reading h_denom)
        g1->h_denom.assign(T, 0);
        for (int t = 0; t < T; ++t) g1->h_denom[t] = g1->N * steps_at(m, sampler, t);
        HIP_TRY(hipMemcpyAsync(g1->acc_denom, g1->h_denom.data(), (size_t)T * sizeof(int), hipMemcpyHostToDevice, s));
    }
    auto noise_for = [&](uint64_t call, NoiseArg& na) -> int {
        na.mode = nz->mode; na.seed = nz->seed; na.row_offset = nz->row_offset;
        na.call = (unsigned int)call; na.normal = nullptr; na.uniform = nullptr; na.ucall = 0;
        if (nz->mode == CCSP_NOISE_INJECTED) {
            if (call < nz->call_base || call - nz->call_base >= nz->n_normal) return fail("compose_chain_run: injected normal stream exhausted at call %llu", (unsigned long long)call);
            na.normal = nz->normal + (size_t)(call - nz->call_base) * NP;
        }
        return 0;
    };
    auto node = [&](const NodeArgs& a) { dispatch_h(m->d.hidden_dim, [&](auto hc) { launch_node<decltype(hc)::value>(m, g, a, s); return 0; }); };
    g->evals = 0; g->kev_used = 0;
    if (!g->have_events) { HIP_TRY(hipEventCreate(&g->ev0)); HIP_TRY(hipEventCreate(&g->ev1)); g->have_events = true; }
    HIP_TRY(hipEventRecord(g->ev0, s));
    {
        NodeArgs a = node_args(m, g);
        a.src = 2; a.do_encode = 1;
        if (init) {
            a.step = STEP_INIT; a.reset_mask = 1; a.hist = history;
            if (noise_for(0, a.noise)) return 1;
        } else {
            HIP_TRY(hipMemcpyAsync(g->x, x, NP * sizeof(float), hipMemcpyDeviceToDevice, s));
            a.step = STEP_NONE;
        }
        node(a);
    }
    for (int t = t_first; t >= t_last; --t) {
        const int S = steps_at(m, sampler, t);
        for (int e = 0; e <= (hmc ? 0 : S); ++e) {
            if (energy) {      // gradient at the state (w.s1: the gradient; g1->eps / g2->eps hold the two domains' own gradients)
                if (compose_energy_eval(m1, g1, m2, g2, c, g->x, t, w.s2, w.p2, b4.f(), w.s1, b4.f() + 2, s)) return 1;
            } else if (compose_eval(m1, g1, m2, g2, c, nullptr, t, w, g->eps, s)) return 1;
            NodeArgs a = node_args(m, g);
            a.src = 1; a.eps_buf = energy ? w.s1 : g->eps; a.do_encode = 1;
            a.step = e == 0 ? STEP_ANCESTRAL : STEP_ULA;
            a.reset_mask = (e == S);
            a.hist = (e == S && history) ? history + (size_t)(T - t) * NP : nullptr;
            a.a_t = m->sqrt_recip_ac[t]; a.b_t = m->sqrt_recipm1_ac[t]; a.c1 = m->coef1[t]; a.c2 = m->coef2[t];
            a.sigma = t != 0 ? expf(0.5f * m->post_lv[t]) : 0.0f;
            a.kappa = m->kappa[t]; a.ss = m->step[t]; a.std_ = sqrtf(2.0f * m->step[t]);
            if (noise_for(call0[t] + (uint64_t)e, a.noise)) return 1;
            if (mala && !hmc && e >= 1) {
                // AnnealedMALASampler.sample_step (ddpm.py:1013-1041) on the composed model: the gradient evaluation above also left E(x)
                // in b4[2]; propose, evaluate the composed energy at the proposal (its gradient goes to scratch), accept per node row
                // from the batch-scalar energies
                a.step = STEP_MALA_PROPOSE; a.do_encode = 0; a.xhat = g->xhat; a.reset_mask = 0; a.hist = nullptr;
                node(a);
                if (compose_energy_eval(m1, g1, m2, g2, c, g->xhat, t, w.s2, w.p2, b4.f(), nullptr, b4.f() + 3, s)) return 1;      // (energy only)
                NodeArgs b = a;
                b.step = STEP_MALA_ACCEPT;
                b.E_x = b4.f() + 2; b.E_hat = b4.f() + 3; b.acc_count = g->acc_count + t;
                b.margin = margin_at(g, ucall0[t] + (uint64_t)(e - 1) - ucall0[t_first]);
                // MALA across shards (ccsp_model_set_energy_hook / _allreduce on the FIRST domain's model): {E(x), E(x_hat)} of this shard ->
                // sums over all shards, in place (b4[2], b4[3] are adjacent and rewritten by the next inner step's evaluations), on this stream
                if (m1->rccl_comm) {
                    RcclApi* ra = rccl_api();
                    const int rc = ra ? ra->all_reduce(b4.f() + 2, b4.f() + 2, 2, 7 /*ncclFloat32*/, 0 /*ncclSum*/, m1->rccl_comm, s) : -1;
                    if (rc != 0) return fail("compose_chain_run: ncclAllReduce of the batch energies failed: %s", rccl_err(ra, rc));
                } else if (m1->energy_hook && m1->energy_hook(m1->energy_hook_ctx, b4.f() + 2, (void*)s)) return fail("compose_chain_run: the energy hook failed");
                b.reset_mask = (e == S);
                b.hist = (e == S && history) ? history + (size_t)(T - t) * NP : nullptr;
                const uint64_t uc = ucall0[t] + (uint64_t)(e - 1);
                b.noise.ucall = (unsigned int)uc;
                if (nz->mode == CCSP_NOISE_INJECTED) {
                    if (uc < nz->ucall_base || uc - nz->ucall_base >= nz->n_uniform)
                        return fail("compose_chain_run: injected uniform stream exhausted at call %llu", (unsigned long long)uc);
                    b.noise.uniform = nz->uniform + (size_t)(uc - nz->ucall_base) * N;
                }
                node(b);
                continue;
            }
            node(a);
        }
        if (hmc && S > 0) {
            // AnnealedMUHASampler.sample_step (ddpm.py:1087-1128; chain_run_impl's HMC block with the composed energy): the leapfrog runs at
            // the INNER index e (step size, mass, gradient timestep), the energies at the real t.  Every evaluation encodes its own poses.
            if (!g->hmc_vk && (dev_alloc(g->allocs, &g->hmc_vk, NP) || dev_alloc(g->allocs, &g->hmc_vp, NP) || dev_alloc(g->allocs, &g->hmc_vl, NP))) return 1;
            const dim3 hgrid(nblk((long)NP, 256));
            auto hargs = [&](int mode) {
                HmcArgs h;
                memset(&h, 0, sizeof(h));
                h.N = g->N; h.P = P; h.F = g->F; h.mode = mode;
                h.x = g->x; h.xl = g->xhat; h.vk = g->hmc_vk; h.vp = g->hmc_vp; h.vl = g->hmc_vl; h.eps = w.s1;
                h.m_t = 9.0f * m->betas[t]; h.kappa_t = m->kappa[t];
                h.mask = g->mask; h.xfeat = g->xfeat; h.pose_begin = m->d.pose_begin;
                return h;
            };
            auto grad_at = [&](const float* poses, int tt, float* grad_out, float* e_out) {
                return compose_energy_eval(m1, g1, m2, g2, c, poses, tt, w.s2, w.p2, b4.f(), grad_out, e_out, s);
            };
            {
                HmcArgs h = hargs(HMC_MOMENTUM);
                if (noise_for(call0[t] + 1, h.noise)) return 1;
                hipLaunchKernelGGL(k_hmc, hgrid, dim3(256), 0, s, h);
            }
            for (int e = 0; e < S; ++e) {
                HmcArgs r = hargs(HMC_REFRESH);
                if (noise_for(call0[t] + 2 + (uint64_t)e, r.noise)) return 1;
                hipLaunchKernelGGL(k_hmc, hgrid, dim3(256), 0, s, r);
                const float m_i = 9.0f * m->betas[e];
                for (int lf = 0; lf < 2; ++lf) {
                    if (lf == 0 && grad_at(g->xhat, e, w.s1, b4.f() + 4)) return 1;
                    HmcArgs la = hargs(HMC_LEAP_A);
                    la.ss_i = m->step[e]; la.md_i = m_i * m_i; la.kap_i = m->kappa[e];
                    hipLaunchKernelGGL(k_hmc, hgrid, dim3(256), 0, s, la);
                    if (grad_at(g->xhat, e, w.s1, b4.f() + 4)) return 1;
                    HmcArgs lb = la;
                    lb.mode = HMC_LEAP_B;
                    hipLaunchKernelGGL(k_hmc, hgrid, dim3(256), 0, s, lb);
                }
                if (grad_at(g->x, t, nullptr, b4.f() + 2) || grad_at(g->xhat, t, nullptr, b4.f() + 3)) return 1;                     // (energies only)
                HmcArgs ac = hargs(HMC_ACCEPT);
                ac.E_x = b4.f() + 2; ac.E_hat = b4.f() + 3; ac.acc_count = g->acc_count + t;
                ac.margin = margin_at(g, ucall0[t] + (uint64_t)e - ucall0[t_first]);
                ac.reset_mask = (e == S - 1);
                ac.hist = (e == S - 1 && history) ? history + (size_t)(T - t) * NP : nullptr;
                ac.noise.mode = nz->mode; ac.noise.seed = nz->seed; ac.noise.row_offset = nz->row_offset;
                const uint64_t uc = ucall0[t] + (uint64_t)e;
                ac.noise.ucall = (unsigned int)uc;
                if (nz->mode == CCSP_NOISE_INJECTED) {
                    if (uc < nz->ucall_base || uc - nz->ucall_base >= nz->n_uniform)
                        return fail("compose_chain_run: injected uniform stream exhausted at call %llu", (unsigned long long)uc);
                    ac.noise.uniform = nz->uniform + (size_t)(uc - nz->ucall_base) * N;
                }
                hipLaunchKernelGGL(k_hmc, hgrid, dim3(256), 0, s, ac);
            }
        }
    }
    if (mala && accept) hipLaunchKernelGGL(k_accept_rates, dim3(nblk(T, 256)), dim3(256), 0, s, T, g->acc_count, g->acc_denom, accept);
    HIP_TRY(hipMemcpyAsync(x, g->x, NP * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipEventRecord(g->ev1, s));
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccsp_plan_host(int32_t N, int32_t E, int32_t C, const int64_t* edge_index, const float* edge_attr, int32_t* counts,
                   int32_t* e_orig, int32_t* e_type, int32_t* e_u0, int32_t* e_u1, int32_t* urow_node, int32_t* urow_ts,
                   int32_t* tile_row0, int32_t* tile_nrows, int32_t* tile_ts, int32_t* node_ptr, int32_t* node_ent) {
    ccsp::Plan p;
    const char* perr = "";
    if (ccsp::build_plan(N, E, C, TILE_M, edge_index, edge_attr, p, &perr)) return fail("plan_host: %s", perr);
    counts[0] = p.E_act; counts[1] = p.R; counts[2] = (int32_t)p.tile_row0.size();
    auto cp = [](int32_t* dst, const std::vector<int32_t>& v) { if (dst && !v.empty()) memcpy(dst, v.data(), v.size() * sizeof(int32_t)); };
    cp(e_orig, p.e_orig); cp(e_type, p.e_type); cp(e_u0, p.e_u0); cp(e_u1, p.e_u1);
    cp(urow_node, p.urow_node); cp(urow_ts, p.urow_ts);
    cp(tile_row0, p.tile_row0); cp(tile_nrows, p.tile_nrows); cp(tile_ts, p.tile_ts);
    cp(node_ptr, p.node_ptr); cp(node_ent, p.node_ent);
    return 0;
}

#ifdef CCSP_EXPERIMENTS
int ccsp_plan_fused_host(int32_t N, int32_t E, int32_t C, const int64_t* edge_index, const float* edge_attr, int32_t rows_per_slot,
                         int32_t max_edges, int32_t* n_tiles, int32_t* tiles, int32_t* rows, uint16_t* e_lu) {
    if (rows_per_slot < 1 || rows_per_slot > 32 || max_edges < 1 || max_edges > 128) return fail("plan_fused_host: rows_per_slot in 1..32, max_edges in 1..128");
    ccsp::Plan p;
    const char* perr = "";
    if (ccsp::build_plan(N, E, C, TILE_M, edge_index, edge_attr, p, &perr)) return fail("plan_fused_host: %s", perr);
    ccsp::FusedPlan f;
    ccsp::build_fused_plan(p, rows_per_slot, max_edges, f);
    *n_tiles = f.n_tiles;
    if (tiles && !f.tiles.empty()) memcpy(tiles, f.tiles.data(), f.tiles.size() * sizeof(int32_t));
    if (rows && !f.rows.empty()) memcpy(rows, f.rows.data(), f.rows.size() * sizeof(int32_t));
    if (e_lu && !f.e_lu.empty()) memcpy(e_lu, f.e_lu.data(), f.e_lu.size() * sizeof(uint16_t));
    return 0;
}
#endif

int ccsp_plan_bwdsum_host(int32_t N, int32_t E, int32_t C, const int64_t* edge_index, const float* edge_attr, int32_t* n_blocks, int32_t* n_partial,
                          int32_t* blocks, int32_t* prow_urow, int32_t* nrow_ptr, int32_t* nrow_idx) {
    return ccsp_plan_bwdsum_blocks_host(N, E, C, edge_index, edge_attr, ccsp::BS_EDGES, 128, n_blocks, n_partial, blocks, prow_urow, nrow_ptr, nrow_idx);
}

int ccsp_plan_bwdsum_blocks_host(int32_t N, int32_t E, int32_t C, const int64_t* edge_index, const float* edge_attr, int32_t block_edges, int32_t max_parts,
                                 int32_t* n_blocks, int32_t* n_partial, int32_t* blocks, int32_t* prow_urow, int32_t* nrow_ptr, int32_t* nrow_idx) {
    if (!n_blocks || !n_partial) return fail("plan_bwdsum_blocks_host: null count");
    if (block_edges < 1 || block_edges > ccsp::BS_EDGES || max_parts < 2 * block_edges) return fail("plan_bwdsum_blocks_host: block_edges %d (1 .. %d), max_parts %d (>= 2 block_edges)", block_edges, ccsp::BS_EDGES, max_parts);
    ccsp::Plan p;
    const char* perr = "";
    if (ccsp::build_plan(N, E, C, TILE_M, edge_index, edge_attr, p, &perr)) return fail("plan_bwdsum_host: %s", perr);
    ccsp::BwdSumPlan b;
    ccsp::build_bwdsum_plan(p, TILE_M, b, block_edges, max_parts);
    *n_blocks = b.n_blocks;
    *n_partial = b.NP;
    if (blocks && !b.blocks.empty()) memcpy(blocks, b.blocks.data(), b.blocks.size() * sizeof(int32_t));
    if (prow_urow && !b.prow_urow.empty()) memcpy(prow_urow, b.prow_urow.data(), b.prow_urow.size() * sizeof(int32_t));
    if (nrow_ptr) memcpy(nrow_ptr, b.nrow_ptr.data(), b.nrow_ptr.size() * sizeof(int32_t));
    if (nrow_idx && !b.nrow_idx.empty()) memcpy(nrow_idx, b.nrow_idx.data(), b.nrow_idx.size() * sizeof(int32_t));
    return 0;
}

}  // extern "C"
