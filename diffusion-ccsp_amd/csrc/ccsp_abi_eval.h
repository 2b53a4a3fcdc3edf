// ccsp_abi_eval.h -- C ABI, part 3: ccsp_denoise, sequences of the transformer baseline, ccsp_edge_outputs, ccsp_energy_grad, ccsp_chain_run, profiling / statistics.
// A fragment of the ONE translation unit csrc/ccsp_hip.hip (included there, at this position, inside its namespaces): not a standalone header.
int ccsp_denoise(ccsp_model* m, ccsp_graph* g, const float* poses_in, int32_t t, float* out, void* stream) {
    if (!m || !g || !poses_in || !out) return fail("denoise: null argument");
    if (g->m != m) return fail("denoise: graph belongs to another model");
    if (t < 0 || t >= m->d.timesteps) return fail("denoise: t=%d out of range", t);
    hipStream_t s = (hipStream_t)stream;
    NodeArgs a = node_args(m, g);
    a.src = 2; a.step = STEP_NONE; a.do_encode = 1; a.x_in = poses_in;
    NodeArgs b = node_args(m, g);
    b.src = 0; b.step = STEP_NONE; b.do_encode = 0; b.eps_out = out; b.x_in = poses_in;
    if (m->d.model_kind == CCSP_MODEL_STRUCT_DIFFUSION) {
        if (dispatch_h(m->d.hidden_dim, [&](auto hc) { constexpr int HH = decltype(hc)::value; launch_node<HH>(m, g, a, s); return launch_eval_sd<HH>(m, g, t, s); })) return 1;
        HIP_TRY(hipMemcpyAsync(out, g->eps, (size_t)g->N * m->d.pose_dim * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (dispatch_h(m->d.hidden_dim, [&](auto hc) {
            constexpr int HH = decltype(hc)::value;
            launch_node<HH>(m, g, a, s);
            if (launch_eval<HH>(m, g, t, s)) return 1;
            launch_node<HH>(m, g, b, s);
            return 0;
        })) return 1;
    HIP_TRY(hipGetLastError());
    return 0;
}

namespace {
// token layout of graphs [b0, b0 + B) of a batch of B_total graphs whose node counts are cnt_all; graph_of / pos_of: the sub-batch's
// nodes (graph ids relative to b0).  The attention mask of (graph b, head h) is the one of graph (b heads + h) mod B_total OF THE
// WHOLE BATCH (`(repeat b)` vs MHA's (b heads) ordering, denoise_fn.py:434): a static property of the batch, so a lane keeps it.
int sequences_build(ccsp_graph* g, int B, int b0, const std::vector<int>& cnt_all, const std::vector<int>& graph_of, const std::vector<int>& pos_of, hipStream_t s) {
    ccsp_model* m = g->m;
    const int N = g->N, B_total = (int)cnt_all.size();
    std::vector<int> cnt(B, 0), tok_node((size_t)B * SD_L, -1), tok_pos((size_t)B * SD_L, 0), node_tok(N), mask_from((size_t)B * SD_HEADS);
    for (int n = 0; n < N; ++n) {
        const int b = graph_of[n];
        if (cnt[b] >= SD_L) return fail("graph_set_sequences: graph %d has more than %d nodes (max_seq_len, denoise_fn.py:272)", b0 + b, SD_L);
        node_tok[n] = b * SD_L + cnt[b];
        tok_node[(size_t)b * SD_L + cnt[b]] = n;
        tok_pos[(size_t)b * SD_L + cnt[b]] = pos_of.empty() ? cnt[b] : pos_of[n];
        cnt[b]++;
    }
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < SD_HEADS; ++h) {
            const int c = cnt_all[(size_t)(((long)(b0 + b) * SD_HEADS + h) % B_total)];
            mask_from[(size_t)b * SD_HEADS + h] = c == SD_L ? 0 : c;   // no padding: `[-0:]` marks everything
        }
    const int M = B * SD_L, Wd = m->Wd;
    auto& reg = g->allocs;
    if (dev_upload(reg, &g->tok_node, tok_node, s) || dev_upload(reg, &g->tok_pos, tok_pos, s) || dev_upload(reg, &g->node_tok, node_tok, s) ||
        dev_upload(reg, &g->mask_from, mask_from, s) || dev_alloc(reg, &g->sdX, (size_t)M * Wd) || dev_alloc(reg, &g->sdY, (size_t)SD_KSPLIT * M * Wd) ||
        dev_alloc(reg, &g->sdQKV, (size_t)2 * M * 3 * Wd) || dev_alloc(reg, &g->sdA, (size_t)M * Wd) || dev_alloc(reg, &g->sdF, (size_t)M * 4 * Wd) ||
        dev_alloc(reg, &g->sdMax, (size_t)4 * M))
        return 1;
    HIP_TRY(hipMemsetAsync(g->sdMax, 0, (size_t)4 * M * sizeof(unsigned int), s));
    HIP_TRY(hipStreamSynchronize(s));       // the host vectors go out of scope
    g->sd_B = B; g->sd_M = M;
    g->seq_ready = true;
    return 0;
}
}  // namespace

int ccsp_graph_set_sequences(ccsp_graph* g, const int64_t* batch, const int64_t* shuffled, void* stream) {
    if (!g || !batch) return fail("graph_set_sequences: null argument");
    ccsp_model* m = g->m;
    if (!m) return fail("graph_set_sequences: the graph's model was destroyed");
    if (m->d.model_kind != CCSP_MODEL_STRUCT_DIFFUSION) return fail("graph_set_sequences: the model is not a StructDiffusion model");
    if (g->seq_ready) return fail("graph_set_sequences: sequences already set for this graph");
    hipStream_t s = (hipStream_t)stream;
    const int N = g->N;
    std::vector<int64_t> hb(N), hs;
    HIP_TRY(hipMemcpyAsync(hb.data(), batch, (size_t)N * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    if (shuffled) { hs.resize(N); HIP_TRY(hipMemcpyAsync(hs.data(), shuffled, (size_t)N * sizeof(int64_t), hipMemcpyDeviceToHost, s)); }
    HIP_TRY(hipStreamSynchronize(s));
    int B = 0;
    for (int n = 0; n < N; ++n) {
        if (hb[n] < 0 || hb[n] >= N) return fail("graph_set_sequences: batch[%d]=%lld out of range", n, (long long)hb[n]);
        if ((int)hb[n] + 1 > B) B = (int)hb[n] + 1;
    }
    std::vector<int> graph_of(N), pos_of, cnt(B, 0);
    for (int n = 0; n < N; ++n) { graph_of[n] = (int)hb[n]; cnt[graph_of[n]]++; }
    if (shuffled) {
        pos_of.resize(N);
        for (int n = 0; n < N; ++n) {
            if (hs[n] < 0 || hs[n] >= cnt[graph_of[n]]) return fail("graph_set_sequences: shuffled[%d]=%lld outside its graph's %d positions", n, (long long)hs[n], cnt[graph_of[n]]);
            pos_of[n] = (int)hs[n];
        }
    }
    if (sequences_build(g, B, 0, cnt, graph_of, pos_of, s)) return 1;
    g->h_seq_graph = std::move(graph_of); g->h_seq_pos = std::move(pos_of); g->h_seq_cnt = std::move(cnt);
    return 0;
}

int ccsp_edge_outputs(ccsp_model* m, ccsp_graph* g, const float* poses_in, int32_t t, float* out, void* stream) {
    if (!m || !g || !poses_in || !out) return fail("edge_outputs: null argument");
    if (g->m != m) return fail("edge_outputs: graph belongs to another model");
    if (m->d.model_kind != CCSP_MODEL_DIFFUSION_CCSP) return fail("edge_outputs: StructDiffusion has no per-edge outputs");
    if (t < 0 || t >= m->d.timesteps) return fail("edge_outputs: t=%d out of range", t);
    hipStream_t s = (hipStream_t)stream;
    const int P = m->d.pose_dim;
    NodeArgs a = node_args(m, g);
    a.src = 2; a.step = STEP_NONE; a.do_encode = 1; a.x_in = poses_in;
    if (dispatch_h(m->d.hidden_dim, [&](auto hc) { constexpr int HH = decltype(hc)::value; launch_node<HH>(m, g, a, s); return launch_eval<HH>(m, g, t, s); })) return 1;
    if (g->E > 0) hipLaunchKernelGGL(k_fill, dim3(nblk((long)g->E * 2 * P, 256)), dim3(256), 0, s, out, (long)g->E * 2 * P, nanf(""));
    if (g->plan.E_act > 0)
        hipLaunchKernelGGL(k_unsort_edges, dim3(nblk((long)g->plan.E_act * 2 * P, 256)), dim3(256), 0, s, g->plan.E_act, P, g->e_orig, g->ent_pos, g->O, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccsp_energy_grad(ccsp_model* m, ccsp_graph* g, const float* poses_in, int32_t t, float* grad, float* energy, void* stream) {
    if (!m || !g || !poses_in || !grad || !energy) return fail("energy_grad: null argument");
    if (g->m != m) return fail("energy_grad: graph belongs to another model");
    if (m->d.model_kind != CCSP_MODEL_DIFFUSION_CCSP) return fail("energy_grad: StructDiffusion has no energy mode");
    if (t < 0 || t >= m->d.timesteps) return fail("energy_grad: t=%d out of range", t);
    hipStream_t s = (hipStream_t)stream;
    if (energy_prepare(m, g, s)) return 1;
    const size_t NP = (size_t)g->N * m->d.pose_dim;
    NodeArgs a = node_args(m, g);
    a.src = 2; a.step = STEP_NONE; a.do_encode = 1; a.x_in = poses_in;
    if (dispatch_h(m->d.hidden_dim, [&](auto hc) {
            constexpr int HH = decltype(hc)::value;
            launch_node<HH>(m, g, a, s);
            return launch_eval_energy<HH>(m, g, t, poses_in, true, energy, s);
        })) return 1;
    HIP_TRY(hipMemcpyAsync(grad, g->eps, NP * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccsp_chain_run(ccsp_model* m, ccsp_graph* g, int32_t sampler, const ccsp_noise* nz, float* x, int32_t init,
                   int32_t t_first, int32_t t_last, float* history, float* accept, void* stream) {
    if (!m || !g || !nz || !x) return fail("chain_run: null argument");
    if (g->m != m) return fail("chain_run: graph belongs to another model");
    const int T = m->d.timesteps;
    if (sampler < 0 || sampler > 4) return fail("chain_run: unknown sampler %d", sampler);
    if (t_first >= T || t_last < 0 || t_first < t_last - 1) return fail("chain_run: bad timestep range [%d,%d]", t_first, t_last);
    if (nz->mode != CCSP_NOISE_PHILOX && nz->mode != CCSP_NOISE_INJECTED) return fail("chain_run: unknown noise mode %d", nz->mode);
    if (nz->mode == CCSP_NOISE_INJECTED && !nz->normal) return fail("chain_run: injected noise without a normal stream");
    if ((sampler == CCSP_SAMPLER_MALA || sampler == CCSP_SAMPLER_HMC) && !m->d.energy_wrapper) return fail("chain_run: MALA / HMC need an energy_wrapper model (train_utils.py:115-116)");
    if (sampler == CCSP_SAMPLER_HMC && T < 4) return fail("chain_run: HMC indexes the schedule with its inner step 0..3 (ddpm.py:1076-1084); timesteps=%d is too short", T);
    hipStream_t s = (hipStream_t)stream;
    const size_t NP_total = (size_t)g->N * m->d.pose_dim;
    // Concurrent lanes (direct mode): graphs are independent, so the batch is cut into sub-batches whose
    // chains run on their own streams, enqueued interleaved.  A chain is three dependent kernels per
    // evaluation, each with fill/drain phases that leave most of the 256 CUs idle; two lanes overlap one
    // lane's latency-bound node kernel and tile tails with the other's GEMMs (+10 % samples/s at C2,
    // bitwise-identical results: noise rows are global).  CCSP_LANES=<k> overrides (1 = off).
    int want = m->lanes;
    // below ~6000 edges the half-batch kernels are too small to overlap usefully (C2-shaped batches: 64 graphs / 5.1 k
    // edges 143 vs 132 samples/s with 1 vs 2 lanes, 96 graphs / 7.6 k edges 167 vs 188); CCSP_LANE_MIN_EDGES overrides
    const bool small = m->d.model_kind == CCSP_MODEL_STRUCT_DIFFUSION ? g->sd_M < m->lane_min_tokens : g->plan.E_act < m->lane_min_edges;
    // energy mode: MALA may run as TWO coupled lanes (MalaCouple, ccsp_chain.h); everything else that evaluates energies stays on one stream
    const bool mala2 = m->d.energy_wrapper && sampler == CCSP_SAMPLER_MALA && m->mala_lanes == 2 && m->d.model_kind == CCSP_MODEL_DIFFUSION_CCSP &&
                       m->d.hidden_dim == 256 && m->f16x2 && m->energy_hook == nullptr && m->rccl_comm == nullptr && g->margin_buf == nullptr;
    if ((m->d.energy_wrapper && !mala2) || g->profile || small || g->N < 2 * want) want = 1;
    if (mala2 && want > 2) want = 2;
    std::vector<Lane> lanes;
    if (want > 1) {
        if (ensure_children(m, g, want, s)) return 1;
        if (!g->children.empty()) {                    // (graph_build synchronised the stream it was built on)
            for (size_t i = 0; i < g->children.size(); ++i) lanes.push_back(Lane{g->children[i], m->lane_streams[i], g->child_node0[i]});
        }
    }
    if (!g->have_events) { HIP_TRY(hipEventCreate(&g->ev0)); HIP_TRY(hipEventCreate(&g->ev1)); g->have_events = true; }
    // relay chains of a device run one after the other: each may then count on the whole chip's workgroup slots (Relay)
    hipEvent_t relay_tail = nullptr;
    const bool relay = m->relay && !m->d.energy_wrapper && m->d.model_kind == CCSP_MODEL_DIFFUSION_CCSP && !g->profile;
    if (relay) {
        if (relay_tail_get(&relay_tail)) return 1;
        HIP_TRY(hipStreamWaitEvent(s, relay_tail, 0));
    }
    HIP_TRY(hipEventRecord(g->ev0, s));
    const bool forked = !lanes.empty();
    if (forked) {
        HIP_TRY(hipEventRecord(m->fork_event, s));
        for (const Lane& L : lanes) HIP_TRY(hipStreamWaitEvent(L.s, m->fork_event, 0));
    } else {
        lanes.push_back(Lane{g, s, 0});
    }
    g->lanes_last = (int)lanes.size();
    MalaCouple couple;
    const bool coupled = forked && m->d.energy_wrapper && lanes.size() == 2;
    if (forked && m->d.energy_wrapper && !coupled) return fail("chain_run: an energy-mode chain runs as one lane or as two coupled MALA lanes");
    if (coupled) {
        for (int i = 0; i < 2; ++i) {
            couple.recF[i].store(0); couple.recA[i].store(0);
            couple.E_x[i] = nullptr; couple.hat_partial[i] = nullptr; couple.n_hat[i] = 0;
        }
        couple.failed.store(0);
        if (!m->mala_ev[0])
            for (int i = 0; i < 4; ++i) HIP_TRY(hipEventCreateWithFlags(&m->mala_ev[i], hipEventDisableTiming));
        couple.evF[0] = m->mala_ev[0]; couple.evF[1] = m->mala_ev[1]; couple.evA[0] = m->mala_ev[2]; couple.evA[1] = m->mala_ev[3];
    }
    for (size_t i = 0; i < lanes.size(); ++i) {
        lanes[i].couple = coupled ? &couple : nullptr;
        lanes[i].idx = (int)i;
        lanes[i].relay_slots = relay ? 2 * m->ncu / (int)lanes.size() : 0;
    }
    int rc = 0;
    auto run = [&](const std::vector<Lane>& ls) -> int {
        return dispatch_h(m->d.hidden_dim, [&](auto hc) {
            return chain_run_impl<decltype(hc)::value>(m, ls, NP_total, sampler, nz, x, init, t_first, t_last, history, accept);
        });
    };
    if (!forked) {
        rc = run(lanes);
    } else {
        // one enqueueing host thread per lane: a single thread alternating between streams is launch-bound
        // (~18 us per launch measured), two threads keep both streams fed
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        std::vector<std::thread> th;
        std::vector<int> rcs(lanes.size(), 0);
        std::vector<std::string> errs(lanes.size());
        for (size_t i = 0; i < lanes.size(); ++i)
            th.emplace_back([&, i]() {
                if (hipSetDevice(dev) != hipSuccess) { rcs[i] = 1; errs[i] = "hipSetDevice failed in lane thread"; return; }
#ifdef CCSP_EXPERIMENTS
                static const int stagger_us = getenv("CCSP_LANE_STAGGER_US") ? atoi(getenv("CCSP_LANE_STAGGER_US")) : 0;
                if (stagger_us > 0 && i > 0) hipLaunchKernelGGL(k_delay, dim3(1), dim3(1), 0, lanes[i].s, (long long)stagger_us * 100 * (long long)i);
#endif
                rcs[i] = run(std::vector<Lane>{lanes[i]});
                if (rcs[i]) { errs[i] = g_err; if (coupled) couple.failed.store(1); }       // (the other lane's hand-shake stops waiting)
            });
        for (auto& t : th) t.join();
        for (size_t i = 0; i < lanes.size(); ++i)
            if (rcs[i]) { rc = fail("%s", errs[i].c_str()); break; }
    }
    if (forked) {
        int64_t ev = 0;
        for (size_t i = 0; i < lanes.size(); ++i) {
            HIP_TRY(hipEventRecord(m->lane_events[i], lanes[i].s));
            HIP_TRY(hipStreamWaitEvent(s, m->lane_events[i], 0));
            ev = lanes[i].g->evals > ev ? lanes[i].g->evals : ev;
        }
        g->evals = ev;
        g->kev_used = 0;
        if (coupled && accept && rc == 0)
            hipLaunchKernelGGL(k_accept_rates2, dim3(nblk(T, 256)), dim3(256), 0, s, T, lanes[0].g->acc_count, lanes[0].g->acc_denom, lanes[1].g->acc_count,
                               lanes[1].g->acc_denom, accept);
    }
    HIP_TRY(hipEventRecord(g->ev1, s));
    if (relay) HIP_TRY(hipEventRecord(relay_tail, s));
    return rc;
}

#ifdef CCSP_TRACE2
int ccsp_debug_trace2(unsigned int* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace2), sizeof(unsigned int) * 4096 * 40) == hipSuccess ? 0 : 1;
}
#endif

#ifdef CCSP_TRACE
int ccsp_debug_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * 3 * 256 * 32) == hipSuccess ? 0 : 1;
}
#endif

int ccsp_profile_enable(ccsp_graph* g, int32_t on) {
    if (!g) return fail("profile_enable: null graph");
    g->profile = on;
    if (on && g->kev.empty()) {
        g->kev.resize(CCSP_PROFILE_MARKS);
        g->kev_id.assign(CCSP_PROFILE_MARKS, -1);
        for (auto& e : g->kev) HIP_TRY(hipEventCreate(&e));
    }
    return 0;
}

int ccsp_kernel_stats(ccsp_graph* g, int32_t which, int64_t* calls, float* ms_mean, char* name, int32_t name_len) {
    if (!g) return fail("kernel_stats: null graph");
    if (which < 0 || which >= CCSP_K_COUNT) return fail("kernel_stats: bad selector %d", which);
    if (!g->have_events) return fail("kernel_stats: no chain has run on this graph");
    HIP_TRY(hipEventSynchronize(g->ev1));
    int64_t n = 0;
    double acc = 0.0;
    for (size_t i = 0; i + 1 < g->kev_used; ++i) {
        if (g->kev_id[i] != which) continue;
        float v = 0.0f;
        HIP_TRY(hipEventElapsedTime(&v, g->kev[i], g->kev[i + 1]));
        acc += v;
        ++n;
    }
    if (calls) *calls = n;
    if (ms_mean) *ms_mean = n ? (float)(acc / (double)n) : 0.0f;
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s", kKernelNames[which]);
    return 0;
}

int ccsp_graph_variant(ccsp_graph* g, int32_t* row_mode, int32_t* edge_tile) {
    if (!g) return fail("graph_variant: null graph");
    ccsp_model* m = g->m;
    if (!m) return fail("graph_variant: the graph's model was destroyed");
    const bool h2 = m->f16x2 && m->d.hidden_dim == 256 && m->d.model_kind == CCSP_MODEL_DIFFUSION_CCSP && g->plan.E_act > 0;
    if (row_mode) *row_mode = h2 ? rowgemm_h2_mode(m, g, 4) : -1;
    if (edge_tile) *edge_tile = h2 ? edge_tile_edges(m, g->plan.E_act) : -1;
    return 0;
}

int ccsp_chain_margins(ccsp_graph* g, float* margins, int64_t n_floats) {
    if (!g) return fail("chain_margins: null graph");
    if (margins && n_floats < 0) return fail("chain_margins: negative size");
    g->margin_buf = margins;
    g->margin_cap = margins ? n_floats : 0;
    return 0;
}

int ccsp_chain_lanes(ccsp_graph* g, int32_t* lanes) {
    if (!g || !lanes) return fail("chain_lanes: null argument");
    *lanes = g->lanes_last;
    return 0;
}

int ccsp_chain_skipped(ccsp_graph* g, int64_t* evaluations_skipped) {
    if (!g || !evaluations_skipped) return fail("chain_skipped: null argument");
    if (!g->have_events) return fail("chain_skipped: no chain has run on this graph");
    HIP_TRY(hipEventSynchronize(g->ev1));
    int n = 0;
    if (g->lanes_last > 1) {                       // coupled MALA lanes: each lane skips on its own state
        for (ccsp_graph* c : g->children) {
            int k = 0;
            if (c->mala_changed) HIP_TRY(hipMemcpy(&k, c->mala_changed + 2, sizeof(int), hipMemcpyDeviceToHost));
            n += k;
        }
    } else if (g->mala_changed) HIP_TRY(hipMemcpy(&n, g->mala_changed + 2, sizeof(int), hipMemcpyDeviceToHost));
    *evaluations_skipped = n;
    return 0;
}

int ccsp_chain_stats(ccsp_graph* g, int64_t* evals, float* ms_total, float* ms_ugemm, float* ms_edge) {
    if (!g) return fail("chain_stats: null graph");
    if (!g->have_events) return fail("chain_stats: no chain has run on this graph");
    HIP_TRY(hipEventSynchronize(g->ev1));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, g->ev0, g->ev1));
    if (evals) *evals = g->evals;
    if (ms_total) *ms_total = ms;
    for (int k = 0; k < 2; ++k) {
        int64_t n = 0;
        float mean = 0.0f;
        if (ccsp_kernel_stats(g, k == 0 ? CCSP_K_ROWGEMM : CCSP_K_EDGE, &n, &mean, nullptr, 0)) return 1;
        if (k == 0 && ms_ugemm) *ms_ugemm = mean;
        if (k == 1 && ms_edge) *ms_edge = mean;
    }
    return 0;
}

// Host-only planning entry (no device needed): lets CPU tests check the index tables.
// Arrays are HOST pointers sized by the caller: per-edge arrays [E], urow_* [2E], tile_* [2E + 2C],
// node_ptr [N+1], node_ent [2E].  counts = {E_act, R, n_tiles}.
int ccsp_compose_denoise(ccsp_model* m1, ccsp_graph* g1, ccsp_model* m2, ccsp_graph* g2, const ccsp_compose* c, const float* poses_in,
                         int32_t t, float* out, void* stream) {
    if (compose_check(m1, g1, m2, g2, c, "compose_denoise", true)) return 1;      // (energy_wrapper models: their direct output, forward(tag != 'EBM'))
    if (!poses_in || !out) return fail("compose_denoise: null argument");
    if (t < 0 || t >= m1->d.timesteps) return fail("compose_denoise: t=%d out of range", t);
    hipStream_t s = (hipStream_t)stream;
    const size_t N = (size_t)g1->N;
    StreamBuf b1(s), b2(s), b3(s);
    if (b1.alloc(N * m1->d.pose_dim * sizeof(float)) || b2.alloc(N * m2->d.pose_dim * sizeof(float)) || b3.alloc(N * m2->d.pose_dim * sizeof(float))) return 1;
    const ComposeScratch w{b1.f(), b2.f(), b3.f()};
    if (compose_eval(m1, g1, m2, g2, c, poses_in, t, w, out, s)) return 1;
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccsp_compose_energy_grad(ccsp_model* m1, ccsp_graph* g1, ccsp_model* m2, ccsp_graph* g2, const ccsp_compose* c, const float* poses_in,
                             int32_t t, float* grad, float* energy, void* stream) {
    if (!m1 || !g1 || !m2 || !g2 || !c || !poses_in || !grad || !energy) return fail("compose_energy_grad: null argument");
    if (g1->m != m1 || g2->m != m2) return fail("compose_energy_grad: a graph belongs to another model");
    if (m1->d.model_kind != CCSP_MODEL_DIFFUSION_CCSP || m2->d.model_kind != CCSP_MODEL_DIFFUSION_CCSP) return fail("compose_energy_grad: both domains must be Diffusion-CCSP models");
    if (!m1->d.energy_wrapper || !m2->d.energy_wrapper) return fail("compose_energy_grad: both models must be energy_wrapper models");
    if (m2->d.pose_dim + 1 != m1->d.pose_dim || m2->d.pose_dim < 2) return fail("compose_energy_grad: the second domain's pose_dim must be the first's minus the zero column");
    if (c->zero_col < 2 || c->zero_col >= m1->d.pose_dim) return fail("compose_energy_grad: zero_col=%d (the second domain's encoder takes pose columns 0 and 1)", c->zero_col);
    if (c->weight_first != 1.0f || c->weight_second != 1.0f) return fail("compose_energy_grad: composing weights other than (1, 1) are built for the direct mode only");
    if (g1->N != g2->N || m1->d.hidden_dim != m2->d.hidden_dim) return fail("compose_energy_grad: the two domains differ in nodes or hidden_dim");
    if (t < 0 || t >= m1->d.timesteps || t >= m2->d.timesteps) return fail("compose_energy_grad: t=%d out of range", t);
    hipStream_t s = (hipStream_t)stream;
    if (energy_prepare(m1, g1, s) || energy_prepare(m2, g2, s)) return 1;
    const int N = g1->N, P2 = m2->d.pose_dim;
    StreamBuf b1(s), b2(s), b3(s);
    if (b1.alloc((size_t)N * P2 * sizeof(float)) || b2.alloc((size_t)N * P2 * sizeof(float)) || b3.alloc(2 * sizeof(float))) return 1;
    if (compose_energy_eval(m1, g1, m2, g2, c, poses_in, t, b1.f(), b2.f(), b3.f(), grad, energy, s)) return 1;
    HIP_TRY(hipGetLastError());
    return 0;
}

