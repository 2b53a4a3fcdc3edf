// ccsp_abi_graph.h -- C ABI, part 2: the sub-module operators (visualize_energy.py:402-450), ccsp_graph_create / destroy.
// A fragment of the ONE translation unit csrc/ccsp_hip.hip (included there, at this position, inside its namespaces): not a standalone header.
// ---- operator-level entry points (visualize_energy.py:402-450 calls the denoiser's sub-modules on its own tensors)
int ccsp_encode(ccsp_model* m, int32_t which, int32_t n, const float* in, float* out, void* stream) {
    if (!m || !in || !out) return fail("encode: null argument");
    if (n < 1) return fail("encode: n=%d", n);
    const ccsp_model_desc& d = m->d;
    EncW w;
    if (which == CCSP_ENC_GEOM) w = EncW{m->ge0_w, m->ge0_b, m->ge2_wT, m->ge2_b, d.geom_dim, nullptr};
    else if (which == CCSP_ENC_POSE) w = EncW{m->pe0_w, m->pe0_b, m->pe2_wT, m->pe2_b, d.pose_dim, nullptr};
    else if (which == CCSP_ENC_GRASP) {
        if (d.grasp_dim <= 0) return fail("encode: the model has no grasp encoder");
        w = EncW{m->gr0_w, m->gr0_b, m->gr2_wT, m->gr2_b, d.grasp_dim, nullptr};
    } else return fail("encode: unknown encoder %d", which);
    hipStream_t s = (hipStream_t)stream;
    dispatch_h(d.hidden_dim, [&](auto hc) {
        hipLaunchKernelGGL(k_encode<decltype(hc)::value>, dim3(nblk(n, NODE_TILE)), dim3(256), 0, s, n, in, w.in_dim, 0, w, out);
        return 0;
    });
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccsp_time_mlp(ccsp_model* m, int32_t n, const float* t_values, float* out, void* stream) {
    if (!m || !t_values || !out) return fail("time_mlp: null argument");
    if (n < 0) return fail("time_mlp: n=%d", n);
    if (n == 0) return 0;                                     // an empty t gives an empty [0, H] result, like the encoders
    const int H = m->d.hidden_dim;
    hipStream_t s = (hipStream_t)stream;
    StreamBuf sinus(s), hid(s);
    if (sinus.alloc((size_t)n * H * sizeof(float)) || hid.alloc((size_t)n * 4 * H * sizeof(float))) return 1;
    hipLaunchKernelGGL(k_sinusoid_values, dim3(nblk((long)n * (H / 2), 256)), dim3(256), 0, s, n, H, t_values, sinus.f());
    hipLaunchKernelGGL(k_linear_rows, dim3(nblk((long)n * 4 * H, 256)), dim3(256), 0, s, n, H, 4 * H, sinus.f(), H, m->tm1_w, H, m->tm1_b, 1, hid.f(), 4 * H);
    hipLaunchKernelGGL(k_linear_rows, dim3(nblk((long)n * H, 256)), dim3(256), 0, s, n, 4 * H, H, hid.f(), 4 * H, m->tm3_w, 4 * H, m->tm3_b, 0, out, H);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccsp_process_constraint(ccsp_model* m, int32_t type, int32_t n, const float* geoms_emb, const float* poses_emb, const float* time_emb,
                            const float* grasp_emb, float* out, void* stream) {
    if (!m || !geoms_emb || !poses_emb || !time_emb || !out) return fail("process_constraint: null argument");
    const ccsp_model_desc& d = m->d;
    if (d.model_kind != CCSP_MODEL_DIFFUSION_CCSP) return fail("process_constraint: StructDiffusion has no per-constraint MLPs");
    if (type < 0 || type >= d.n_types) return fail("process_constraint: constraint type %d out of range", type);
    if (n < 0) return fail("process_constraint: n=%d", n);
    if (n == 0) return 0;
    if ((d.grasp_dim > 0) != (grasp_emb != nullptr)) return fail("process_constraint: grasp_emb must be given exactly for 'robot' models");
    const int H = d.hidden_dim, P = d.pose_dim;
    const size_t WS = (size_t)2 * H * H;
    hipStream_t s = (hipStream_t)stream;
    StreamBuf hb(s), qb(s);
    if (hb.alloc((size_t)n * 2 * H * sizeof(float)) || qb.alloc((size_t)n * 2 * (H / 2) * sizeof(float))) return 1;
    float *h = hb.f(), *q = qb.f();
    hipLaunchKernelGGL(k_type_mlp_rows, dim3(nblk((long)n * 2 * H, 256)), dim3(256), 0, s, n, H, grasp_emb, geoms_emb, poses_emb, time_emb,
                       m->Wr ? m->Wr + (size_t)(2 * type) * WS : (const float*)nullptr, m->Wg + (size_t)(2 * type) * WS, m->Wg + (size_t)(2 * type + 1) * WS,
                       m->Wp + (size_t)(2 * type) * WS, m->Wp + (size_t)(2 * type + 1) * WS, m->Wt + (size_t)type * WS, m->bt + (size_t)type * 2 * H, h);
    // pose_decoder on both halves: h [n, 2H] read as [2n, H]  (denoise_fn.py:357-366)
    hipLaunchKernelGGL(k_linear_rows, dim3(nblk((long)2 * n * (H / 2), 256)), dim3(256), 0, s, 2 * n, H, H / 2, h, H, m->pd0_w, H, m->pd0_b, 2, q, H / 2);
    hipLaunchKernelGGL(k_linear_rows, dim3(nblk((long)2 * n * P, 256)), dim3(256), 0, s, 2 * n, H / 2, P, q, H / 2, m->pd2_w, H / 2, m->pd2_b, 0, out, P);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccsp_graph_create(ccsp_model* m, int32_t N, int32_t E, int32_t F, const float* x, const int64_t* edge_index,
                      const float* edge_attr, const int8_t* mask, void* stream, ccsp_graph** out) {
    if (!m || !x || !mask || !out || (E > 0 && (!edge_index || !edge_attr))) return fail("graph_create: null argument");
    const ccsp_model_desc& d = m->d;
    const int H = d.hidden_dim, P = d.pose_dim;
    if (N < 1 || E < 0) return fail("graph_create: bad sizes N=%d E=%d", N, E);
    if (F < d.pose_begin + P || F < d.geom_dim || F < P) return fail("graph_create: F=%d too small for the model's dims", F);
    if (d.grasp_dim > 0 && F < d.grasp_begin + d.grasp_dim) return fail("graph_create: F=%d too small for the grasp columns", F);
    hipStream_t s = (hipStream_t)stream;
    // one-time read-back of the edge lists (denoise_fn.py:317-318 does this on every evaluation)
    std::vector<int64_t> ei((size_t)2 * E);
    std::vector<float> ea((size_t)E);
    if (E > 0) {
        HIP_TRY(hipMemcpyAsync(ei.data(), edge_index, ei.size() * sizeof(int64_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(ea.data(), edge_attr, ea.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return graph_build(m, N, E, F, x, (const signed char*)mask, std::move(ei), std::move(ea), s, out);
}

void ccsp_graph_destroy(ccsp_graph* g) {
    if (!g) return;
    if (g->m) {
        if (!g->children.empty())                   // lane streams belong to the model; drain them first
            for (hipStream_t st : g->m->lane_streams) (void)hipStreamSynchronize(st);
        auto& reg = g->m->graphs;
        for (size_t i = 0; i < reg.size(); ++i)
            if (reg[i] == g) { reg[i] = reg.back(); reg.pop_back(); break; }
    }                                               // (an orphan: ccsp_model_destroy drained and destroyed the streams)
    for (ccsp_graph* c : g->children) ccsp_graph_destroy(c);
    for (auto& kv : g->execs) (void)hipGraphExecDestroy(kv.second);
    for (void* p : g->allocs) (void)hipFree(p);
    if (g->have_events) { (void)hipEventDestroy(g->ev0); (void)hipEventDestroy(g->ev1); }
    for (hipEvent_t e : g->kev) (void)hipEventDestroy(e);
    delete g;
}

