// ccsp_graph_build.h -- graph_build: the one-time plan, index tables, geometry / grasp embeddings and `base` rows of a collated batch.
// A fragment of the ONE translation unit csrc/ccsp_hip.hip (included there, at this position, inside its namespaces): not a standalone header.
int graph_build(ccsp_model* m, int N, int E, int F, const float* x, const signed char* mask, std::vector<int64_t>&& ei,
                std::vector<float>&& ea, hipStream_t s, ccsp_graph** out) {
    const ccsp_model_desc& d = m->d;
    const int H = d.hidden_dim, P = d.pose_dim;
    ccsp_graph* g = new ccsp_graph();
    g->m = m; g->N = N; g->E = E; g->F = F;
    const char* perr = "";
    if (ccsp::build_plan(N, E, d.n_types, TILE_M, ei.data(), ea.data(), g->plan, &perr)) {
        delete g;
        return fail("graph_create: %s", perr);
    }
    m->graphs.push_back(g);
    g->h_ei = std::move(ei);
    g->h_ea = std::move(ea);
    const ccsp::Plan& p = g->plan;
    g->n_tiles = (int)p.tile_row0.size();
    auto& reg = g->allocs;
#define TRY(x) do { if (x) { ccsp_graph_destroy(g); return 1; } } while (0)
    TRY(dev_alloc(reg, &g->xfeat, (size_t)N * F));
    HIP_TRY(hipMemcpyAsync(g->xfeat, x, (size_t)N * F * sizeof(float), hipMemcpyDeviceToDevice, s));
    TRY(dev_alloc(reg, &g->mask, (size_t)N));
    HIP_TRY(hipMemcpyAsync(g->mask, mask, (size_t)N, hipMemcpyDeviceToDevice, s));
    TRY(dev_upload(reg, &g->e_type, p.e_type, s));
    TRY(dev_upload(reg, &g->e_u0, p.e_u0, s));
    TRY(dev_upload(reg, &g->e_u1, p.e_u1, s));
    TRY(dev_upload(reg, &g->e_orig, p.e_orig, s));
    TRY(dev_upload(reg, &g->urow_node, p.urow_node, s));
    TRY(dev_upload(reg, &g->tile_row0, p.tile_row0, s));
    TRY(dev_upload(reg, &g->tile_nrows, p.tile_nrows, s));
    TRY(dev_upload(reg, &g->tile_ts, p.tile_ts, s));
    TRY(dev_upload(reg, &g->node_ptr, p.node_ptr, s));
    TRY(dev_upload(reg, &g->node_ent, p.node_ent, s));
    TRY(dev_upload(reg, &g->ent_pos, p.ent_pos, s));
    TRY(dev_upload(reg, &g->urow_ts, p.urow_ts, s));
    {   // 128-row tiles: consecutive 64-row plan tiles of one (type, slot) group, two at a time
        std::vector<int> r0, nr, tsv;
        for (size_t i = 0; i < p.tile_row0.size();) {
            const bool pair = i + 1 < p.tile_row0.size() && p.tile_ts[i + 1] == p.tile_ts[i] &&
                              p.tile_row0[i + 1] == p.tile_row0[i] + p.tile_nrows[i];
            r0.push_back(p.tile_row0[i]);
            nr.push_back(p.tile_nrows[i] + (pair ? p.tile_nrows[i + 1] : 0));
            tsv.push_back(p.tile_ts[i]);
            i += pair ? 2 : 1;
        }
        g->n_tiles2 = (int)r0.size();
        g->h_t2.assign(r0.begin(), r0.end());
        g->h_t2.insert(g->h_t2.end(), nr.begin(), nr.end());
        g->h_t2.insert(g->h_t2.end(), tsv.begin(), tsv.end());
        int* t2 = nullptr;
        TRY(dev_upload(reg, &t2, g->h_t2, s));
        g->t2_row0 = t2; g->t2_nrows = t2 + g->n_tiles2; g->t2_ts = t2 + 2 * g->n_tiles2;
        for (size_t i = 0; i < p.tile_row0.size(); ++i) g->h_td.push_back(make_int4(p.tile_row0[i], p.tile_nrows[i], p.tile_ts[i], 0));
        for (size_t i = 0; i < r0.size(); ++i) g->h_td.push_back(make_int4(r0[i], nr[i], tsv[i], 0));
        int4* td = nullptr;
        TRY(dev_upload(reg, &td, g->h_td, s));
        g->td64 = td; g->td128 = td + p.tile_row0.size();
        // the forward row GEMM's gather per tile row (64-row tiles, then their 128-row pairs): one dependent round trip less in front of its first operands
        std::vector<int>& tr = g->h_tr;
        tr.reserve((p.tile_row0.size() * 64 + r0.size() * 128));
        for (size_t i = 0; i < p.tile_row0.size(); ++i)
            for (int r = 0; r < 64; ++r) tr.push_back(p.tile_nrows[i] > 0 ? p.urow_node[p.tile_row0[i] + std::min(r, p.tile_nrows[i] - 1)] : 0);
        for (size_t i = 0; i < r0.size(); ++i)
            for (int r = 0; r < 128; ++r) tr.push_back(nr[i] > 0 ? p.urow_node[r0[i] + std::min(r, nr[i] - 1)] : 0);
        if (!tr.empty()) {
            int* trd = nullptr;
            TRY(dev_upload(reg, &trd, tr, s));
            g->tr64 = trd; g->tr128 = trd + p.tile_row0.size() * 64;
        }
    }
#ifdef CCSP_EXPERIMENTS
    if (m->f16x2 && m->WpF && m->eval_fused && p.E_act > 0) {   // fused tiles: <= 28 (32) U rows per slot, <= 112 (128) edges
        ccsp::build_fused_plan(p, m->eval_fused == 1 ? F4_RS : FZ_RS, m->eval_fused == 1 ? F4_ME : FZ_ME, g->fplan);
        g->n_ftiles = g->fplan.n_tiles;
        int* ft = nullptr;
        TRY(dev_upload(reg, &ft, g->fplan.tiles, s));
        g->ft_tiles = reinterpret_cast<int4*>(ft);
        TRY(dev_upload(reg, &g->ft_rows, g->fplan.rows, s));
        TRY(dev_upload(reg, &g->ft_elu, g->fplan.e_lu, s));
        {   // items by decreasing cost (matrix-pipe time: the row GEMM of a tile is constant, the decoder grows with the 32-edge blocks);
            // a stable sort keeps a type's tiles together (they stream the same weights through the XCDs' L2s)
            std::vector<int> key(g->n_ftiles);
            for (int i = 0; i < g->n_ftiles; ++i) key[i] = (g->fplan.tiles[4 * i + 2] + 31) / 32;
            g->h_forder.resize((size_t)2 * g->n_ftiles);
            for (int i = 0; i < 2 * g->n_ftiles; ++i) g->h_forder[i] = i;
            std::stable_sort(g->h_forder.begin(), g->h_forder.end(), [&](int a, int b) { return key[a >> 1] > key[b >> 1]; });
            TRY(dev_upload(reg, &g->ft_order, g->h_forder, s));
        }
    }
#endif
    TRY(dev_alloc(reg, &g->base, (size_t)p.R * 2 * H));
    TRY(dev_alloc(reg, &g->U, (size_t)p.R * 2 * H));
    TRY(dev_alloc(reg, &g->O, (size_t)2 * p.E_act * P));
    TRY(dev_alloc(reg, &g->pemb, (size_t)N * H));
    TRY(dev_alloc(reg, &g->pembS, (size_t)3 * N * H));
    if (m->f16x2) {
        TRY(dev_alloc(reg, &g->pembH, (size_t)2 * N * H));
        TRY(dev_alloc(reg, &g->pexp, (size_t)N));
        TRY(dev_alloc(reg, &g->umax, (size_t)p.R * 8));
    }
    TRY(dev_alloc(reg, &g->x, (size_t)N * P));
    TRY(dev_alloc(reg, &g->eps, (size_t)N * P));
    // chain-constant part: geometry (and grasp) embeddings -> per-row products base[r] (the reference
    // re-evaluates the geometry encoder and these products on every call, denoise_fn.py:474-475)
    if (d.model_kind == CCSP_MODEL_STRUCT_DIFFUSION) {
        // the transformer reads the embeddings themselves; the constraint edges are not used
        TRY(dev_alloc(reg, &g->gemb, (size_t)N * H));
        const EncW wg{m->ge0_w, m->ge0_b, m->ge2_wT, m->ge2_b, d.geom_dim, nullptr};
        const EncW wr{m->gr0_w, m->gr0_b, m->gr2_wT, m->gr2_b, d.grasp_dim, nullptr};
        if (d.grasp_dim > 0) TRY(dev_alloc(reg, &g->remb, (size_t)N * H));
        dispatch_h(H, [&](auto hc) {
            constexpr int HH = decltype(hc)::value;
            hipLaunchKernelGGL(k_encode<HH>, dim3(nblk(N, NODE_TILE)), dim3(256), 0, s, N, g->xfeat, F, 0, wg, g->gemb);
            if (d.grasp_dim > 0) hipLaunchKernelGGL(k_encode<HH>, dim3(nblk(N, NODE_TILE)), dim3(256), 0, s, N, g->xfeat, F, d.grasp_begin, wr, g->remb);
            return 0;
        });
    } else if (p.E_act > 0) {
        float *gemb = nullptr, *UR = nullptr, *remb = nullptr;
        TRY(dev_alloc(reg, &gemb, (size_t)N * H));
        const EncW wg{m->ge0_w, m->ge0_b, m->ge2_wT, m->ge2_b, d.geom_dim, nullptr};
        const int gwork = g->n_tiles * (2 * H / TILE_N);
        const dim3 ggrid(gwork < m->max_wgs ? gwork : m->max_wgs);
        const float* nof = nullptr;
        dispatch_h(H, [&](auto hc) {
            constexpr int HH = decltype(hc)::value;
            hipLaunchKernelGGL(k_encode<HH>, dim3(nblk(N, NODE_TILE)), dim3(256), 0, s, N, g->xfeat, F, 0, wg, gemb);
            hipLaunchKernelGGL((k_rowgemm<HH, 2 * HH>), ggrid, dim3(256), 0, s, gwork, gemb, g->urow_node, g->tile_row0, g->tile_nrows, g->tile_ts, m->Wg, (size_t)2 * H * H, nof, nof, g->base);
            return 0;
        });
        if (d.grasp_dim > 0) {
            TRY(dev_alloc(reg, &remb, (size_t)N * H));
            TRY(dev_alloc(reg, &UR, (size_t)p.R * 2 * H));
            const EncW wr{m->gr0_w, m->gr0_b, m->gr2_wT, m->gr2_b, d.grasp_dim, nullptr};
            dispatch_h(H, [&](auto hc) {
                constexpr int HH = decltype(hc)::value;
                hipLaunchKernelGGL(k_encode<HH>, dim3(nblk(N, NODE_TILE)), dim3(256), 0, s, N, g->xfeat, F, d.grasp_begin, wr, remb);
                hipLaunchKernelGGL((k_rowgemm<HH, 2 * HH>), ggrid, dim3(256), 0, s, gwork, remb, g->urow_node, g->tile_row0, g->tile_nrows, g->tile_ts, m->Wr, (size_t)2 * H * H, nof, nof, UR);
                return 0;
            });
            hipLaunchKernelGGL(k_rowbase, dim3(nblk((long)p.R * 2 * H, 256)), dim3(256), 0, s, p.R, 2 * H, g->urow_ts, UR, g->base);
        }
    }
#undef TRY
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        ccsp_graph_destroy(g);
        return fail("graph_create: device set-up failed: %s", hipGetErrorString(hipGetLastError()));
    }
    *out = g;
    return 0;
}
