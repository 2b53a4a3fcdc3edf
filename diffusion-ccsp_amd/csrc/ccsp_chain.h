// ccsp_chain.h -- chain drivers: margins, samples-per-step, lane streams, chain_run_impl (ancestral / ULA / ULA+ / MALA / HMC), children (lanes) of a graph.
// A fragment of the ONE translation unit csrc/ccsp_hip.hip (included there, at this position, inside its namespaces): not a standalone header.

// slot k (= index of the accept step within this call, chain order) of the margin buffer installed by ccsp_chain_margins, or null
float* margin_at(const ccsp_graph* g, uint64_t k) {
    if (!g->margin_buf || (int64_t)((k + 1) * 2 * (uint64_t)g->N) > g->margin_cap) return nullptr;
    return g->margin_buf + (size_t)k * 2 * g->N;
}

int steps_at(const ccsp_model* m, int sampler, int t) {
    if (sampler == CCSP_SAMPLER_NONE) return 0;
    if (t % m->d.ebm_per_steps != 0) return 0;                 // ddpm.py:330
    if (sampler == CCSP_SAMPLER_HMC) return 4;                 // samples_per_step = 4, ddpm.py:311
    if (sampler == CCSP_SAMPLER_ULA_PLUS) {                    // ddpm.py:297-299
        const int n = m->d.timesteps / 4;
        int q = n > 0 ? t / n : 3;
        if (q > 3) q = 3;
        return 4 * (q + 1);
    }
    return m->sps[t];
}

// (experiment, CCSP_LANE_STAGGER_US) holds a lane's stream back at the start of a chain so that the lanes' kernels of the same kind do
// not run side by side; wall_clock64 ticks at 100 MHz
#ifdef CCSP_EXPERIMENTS
__global__ void k_delay(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
#endif

// Lane streams are shared by every model of the process, one pool per device.  HIP maps streams onto a handful of hardware queues in creation
// order, so the streams a SECOND model created for itself could land on one queue next to each other: its two lanes then ran one after the other
// (round 5, bench.py's strict-fp32 sub-run: 176 samples/s on a second model's own streams against 257 in a process of its own).  Pooled, every
// model's lane k is the same stream; chains of different models enqueued on it simply queue up like work on the caller's stream.
int lane_stream_get(size_t k, hipStream_t* out) {
    static std::mutex mu;
    static std::map<int, std::vector<hipStream_t>> pool;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    std::vector<hipStream_t>& v = pool[dev];
    while (v.size() <= k) {
        hipStream_t cs = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        v.push_back(cs);
    }
    *out = v[k];
    return 0;
}

// the event behind the device's last relay chain (Relay)
int relay_tail_get(hipEvent_t* out) {
    static std::mutex mu;
    static std::map<int, hipEvent_t> tail;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    auto it = tail.find(dev);
    if (it == tail.end()) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        it = tail.emplace(dev, e).first;
    }
    *out = it->second;
    return 0;
}

// one concurrently running sub-batch of a chain
// Two coupled MALA lanes (round 6; VERDICT r03-r05 "two MALA lanes with the pair {E(x), E(x_hat)} exchanged through events").  The reference's
// accept test uses ONE scalar energy for the whole batch (ddpm.py:1026-1038), so two sub-batches on two streams are independent except at the
// accept step, which needs both lanes' E(x) and E(x_hat).  Each lane's accept kernel reads the other lane's E(x) scalar and proposal-energy
// partials straight from its buffers and adds lane 0's sum + lane 1's sum (both lanes the same order: the same decision inputs).  Ordering:
//   evF[l]: recorded by lane l behind its forward pass at the proposal (its E(x), written by the gradient evaluation before, and its partials are
//           final); the OTHER lane's accept kernel waits for it;
//   evA[l]: recorded behind lane l's accept kernel (it has read the other lane's buffers); the other lane waits for it before the next kernel
//           that rewrites them (its next gradient evaluation -- ancestral or Langevin).
// An event must have been RECORDED (host call) before another thread's hipStreamWaitEvent can name that record, so the two enqueueing threads
// shake hands through recF / recA (how many records each has made); a lane can never be a whole inner step ahead of the other, which is also
// what makes two events per lane enough.  Nothing waits on the device for anything that is not already enqueued: no hang is possible.
struct MalaCouple {
    std::atomic<long> recF[2], recA[2];
    std::atomic<int> failed;
    hipEvent_t evF[2], evA[2];
    const float* E_x[2];            // each lane's E(x) device scalar
    const float* hat_partial[2];    // ... and proposal-energy partials (per workgroup of its forward decoder kernel)
    int n_hat[2];
};

struct Lane {
    ccsp_graph* g;
    hipStream_t s;
    int node0;          // global index of the lane's first node (noise rows, output slices)
    int idx = 0;        // lane index (relay mode: which pair of pooled streams)
    int relay_slots = 0;   // relay mode: workgroup slots this lane may hold at once (0 = relay off), see relay_begin
    MalaCouple* couple = nullptr;   // MALA on two coupled lanes
};

// Relay mode of one lane (Gate).  Safe only while EVERY workgroup of the lane's three kernels can be resident at once -- a workgroup that
// polls a counter holds its slot, so a producer that found no room would never run.  Slot model: any mix of two workgroups of these kernels fits
// a CU (LDS <= 74 KB, <= 248 VGPRs per wave, one wave per SIMD each), so 2 x CUs workgroups of any mix are always placeable (if one were not,
// every CU would hold two already); the lanes of a chain share that budget and relay chains of a device run one after the other (relay_tail in
// ccsp_chain_run).  Lists above the budget run the stream-ordered launches.
struct Relay {
    bool on = false;
    hipStream_t sE = nullptr, sN = nullptr;
    unsigned int nR = 0, nE = 0, nN = 0, ev = 0;
};
__global__ void k_relay_fault(const unsigned int* ctr, float* x, long n) {      // a gate timed out: the chain's result is void
    if (ctr[3] == 0) return;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = __builtin_nanf("");
}
int relay_begin(ccsp_model* m, const Lane& L, Relay* r) {
    ccsp_graph* g = L.g;
    r->on = false;
#ifndef CCSP_EXPERIMENTS
    (void)m; (void)g;
    return 0;
#else
    if (L.relay_slots <= 0 || !m->f16x2 || !m->bf16x3 || !m->pe2_wH || m->node_generic || m->d.model_kind != CCSP_MODEL_DIFFUSION_CCSP ||
        m->d.energy_wrapper || g->plan.E_act <= 0 || g->profile || m->d.hidden_dim != 256) return 0;
    constexpr int H = 256;
    const int mode = rowgemm_h2_mode(m, g, 2 * H / 128);
    if (mode != 0 && mode != 4 && mode != 6) return 0;
    r->nR = (unsigned int)(((mode == 4 || mode == 6) ? g->n_tiles : g->n_tiles2) * (2 * H / 128));
    r->nE = (unsigned int)nblk(g->plan.E_act, edge_tile_edges(m, g->plan.E_act));
    r->nN = (unsigned int)nblk(g->N, NODE_TILE);
    if ((long)r->nR + r->nE + r->nN > (long)L.relay_slots) return 0;
    if (!g->relay_ctr) {
        if (dev_alloc(g->allocs, &g->relay_ctr, 4)) return 1;
        for (int i = 0; i < 3; ++i) HIP_TRY(hipEventCreateWithFlags(&g->relay_ev[i], hipEventDisableTiming));
    }
    if (lane_stream_get(8 + 2 * (size_t)L.idx, &r->sE) || lane_stream_get(9 + 2 * (size_t)L.idx, &r->sN)) return 1;
    HIP_TRY(hipMemsetAsync(g->relay_ctr, 0, 4 * sizeof(unsigned int), L.s));
    HIP_TRY(hipEventRecord(g->relay_ev[0], L.s));            // (behind the chain's first node launch: the state and its embeddings)
    HIP_TRY(hipStreamWaitEvent(r->sE, g->relay_ev[0], 0));
    HIP_TRY(hipStreamWaitEvent(r->sN, g->relay_ev[0], 0));
    r->ev = 0;
    r->on = true;
    g->relay_chains++;
    return 0;
#endif
}
int relay_end(const ccsp_model* m, const Lane& L, const Relay& r) {
    ccsp_graph* g = L.g;
    HIP_TRY(hipEventRecord(g->relay_ev[1], r.sE));
    HIP_TRY(hipEventRecord(g->relay_ev[2], r.sN));
    HIP_TRY(hipStreamWaitEvent(L.s, g->relay_ev[1], 0));
    HIP_TRY(hipStreamWaitEvent(L.s, g->relay_ev[2], 0));
    const long n = (long)g->N * m->d.pose_dim;
    hipLaunchKernelGGL(k_relay_fault, dim3(nblk(n, 256)), dim3(256), 0, L.s, g->relay_ctr, g->x, n);
    return 0;
}

// Enqueues timesteps t_first..t_last for every lane, interleaved kernel by kernel so that all lane
// streams advance together.  NP_total = rows x P of the whole batch (history / injected-noise stride).
template <int H>
int chain_run_impl(ccsp_model* m, const std::vector<Lane>& lanes, size_t NP_total, int sampler, const ccsp_noise* nz, float* x_io,
                   int init, int t_first, int t_last, float* history, float* accept) {
    const int T = m->d.timesteps, P = m->d.pose_dim;
    std::vector<uint64_t> call0(T);
    {   // HMC draws the momentum once per timestep on top of its S refreshments (ddpm.py:1090,1096)
        uint64_t c = 1;
        for (int t = T - 1; t >= 0; --t) {
            call0[t] = c;
            const int S = steps_at(m, sampler, t);
            c += 1 + (uint64_t)S + (sampler == CCSP_SAMPLER_HMC && S > 0 ? 1 : 0);
        }
    }
    auto noise_for = [&](const Lane& L, uint64_t call, NoiseArg& na) -> int {
        na.mode = nz->mode; na.seed = nz->seed; na.row_offset = nz->row_offset + (unsigned long long)L.node0;
        na.call = (unsigned int)call; na.normal = nullptr; na.uniform = nullptr; na.ucall = 0;
        if (nz->mode == CCSP_NOISE_INJECTED) {
            if (call < nz->call_base || call - nz->call_base >= nz->n_normal) return fail("chain_run: injected normal stream exhausted at call %llu", (unsigned long long)call);
            na.normal = nz->normal + (size_t)(call - nz->call_base) * NP_total + (size_t)L.node0 * P;
        }
        return 0;
    };
    auto hist_at = [&](const Lane& L, int k) -> float* { return history ? history + (size_t)k * NP_total + (size_t)L.node0 * P : nullptr; };
    auto sched = [&](NodeArgs& a, int t) {
        a.a_t = m->sqrt_recip_ac[t]; a.b_t = m->sqrt_recipm1_ac[t]; a.c1 = m->coef1[t]; a.c2 = m->coef2[t];
        a.sigma = t != 0 ? expf(0.5f * m->post_lv[t]) : 0.0f;
        a.kappa = m->kappa[t]; a.ss = m->step[t]; a.std_ = sqrtf(2.0f * m->step[t]);
    };
    for (const Lane& L : lanes) {
        ccsp_graph* g = L.g;
        g->evals = 0; g->kev_used = 0;
        if (init) {
            NodeArgs a = node_args(m, g);
            a.src = 2; a.step = STEP_INIT; a.reset_mask = 1; a.do_encode = 1; a.hist = hist_at(L, 0);
            if (noise_for(L, 0, a.noise)) return 1;
            launch_node<H>(m, g, a, L.s);
        } else {
            HIP_TRY(hipMemcpyAsync(g->x, x_io + (size_t)L.node0 * P, (size_t)g->N * P * sizeof(float), hipMemcpyDeviceToDevice, L.s));
            NodeArgs a = node_args(m, g);
            a.src = 2; a.step = STEP_NONE; a.do_encode = 1;
            launch_node<H>(m, g, a, L.s);
        }
    }
    const bool energy = m->d.energy_wrapper != 0;
    if (energy) {
        // energy mode couples the whole batch through one scalar: always a single lane
        ccsp_graph* g = lanes[0].g;
        hipStream_t s = lanes[0].s;
        const Lane& L = lanes[0];
        const int N = g->N;
        std::vector<uint64_t> ucall0(T, 0);
        if (energy_prepare(m, g, s)) return 1;
        HIP_TRY(hipMemsetAsync(g->acc_count, 0, (size_t)T * sizeof(int), s));
        HIP_TRY(hipMemsetAsync(g->mala_changed, 0, 3 * sizeof(int), s));
        HIP_TRY(hipStreamSynchronize(s));      // a previous chain may still be reading h_denom
        g->h_denom.assign(T, 0);
        uint64_t uc0 = 0;
        for (int t = T - 1; t >= 0; --t) {
            ucall0[t] = uc0;
            if (sampler == CCSP_SAMPLER_MALA || sampler == CCSP_SAMPLER_HMC) { uc0 += (uint64_t)steps_at(m, sampler, t); g->h_denom[t] = N * steps_at(m, sampler, t); }
        }
        HIP_TRY(hipMemcpyAsync(g->acc_denom, g->h_denom.data(), (size_t)T * sizeof(int), hipMemcpyHostToDevice, s));
        // coupled lanes (MalaCouple): this lane's side of the hand-shake
        MalaCouple* cp = L.couple;
        const int me = L.idx, other = 1 - L.idx;
        const size_t N_total = NP_total / (size_t)P;
        long kstep = 0;                                  // accept steps enqueued so far (both lanes count the same steps)
        bool owe_wait = false;                           // the other lane's last accept kernel may still read this lane's E(x) / partials
        auto spin = [&](std::atomic<long>& ctr, long want) -> int {
            while (ctr.load(std::memory_order_acquire) < want) {
                if (cp->failed.load(std::memory_order_relaxed)) return fail("chain_run: the other MALA lane failed");
                std::this_thread::yield();
            }
            return 0;
        };
        auto before_rewrite = [&]() -> int {             // in front of every evaluation that follows an accept step
            if (!cp || !owe_wait) return 0;
            if (spin(cp->recA[other], kstep)) return 1;
            HIP_TRY(hipStreamWaitEvent(s, cp->evA[other], 0));
            owe_wait = false;
            return 0;
        };
        if (cp) { cp->E_x[me] = g->Escal; cp->hat_partial[me] = g->partial; }
        for (int t = t_first; t >= t_last; --t) {
            // epsilon = dE/dposes (ComposedEBMDenoiseFn.forward); MALA re-evaluates E at the proposal
            // (energy_function, ddpm.py:285-289) -- the gradient pass already gave E(x)
            const int S = steps_at(m, sampler, t);
            float* E_x = g->Escal, *E_hat = g->Escal + 1;
            {
                NodeArgs a = node_args(m, g);
                a.src = 1; a.eps_buf = g->eps; a.do_encode = 1; a.step = STEP_ANCESTRAL;
                a.reset_mask = (S == 0);
                a.hist = S == 0 ? hist_at(L, T - t) : nullptr;
                sched(a, t);
                if (noise_for(L, call0[t], a.noise)) return 1;
                bool tail_done = false;
                if (before_rewrite()) return 1;
                if (launch_eval_energy<H>(m, g, t, g->x, true, E_x, s, nullptr, nullptr, 0, &a, &tail_done)) return 1;
                if (!tail_done) launch_node<H>(m, g, a, s);
            }
            if (sampler == CCSP_SAMPLER_HMC && S > 0) {
                // AnnealedMUHASampler.sample_step (ddpm.py:1087-1128); see ccsp_hmc.h.  The leapfrog runs at
                // the INNER index e (step size, mass, gradient timestep), the energies at the real t.
                if (!g->hmc_vk && (dev_alloc(g->allocs, &g->hmc_vk, (size_t)N * P) || dev_alloc(g->allocs, &g->hmc_vp, (size_t)N * P) ||
                                   dev_alloc(g->allocs, &g->hmc_vl, (size_t)N * P))) return 1;
                const dim3 hgrid(nblk((long)N * P, 256));
                auto hargs = [&](int mode) {
                    HmcArgs h;
                    memset(&h, 0, sizeof(h));
                    h.N = N; h.P = P; h.F = g->F; h.mode = mode;
                    h.x = g->x; h.xl = g->xhat; h.vk = g->hmc_vk; h.vp = g->hmc_vp; h.vl = g->hmc_vl; h.eps = g->eps;
                    h.m_t = 9.0f * m->betas[t]; h.kappa_t = m->kappa[t];
                    h.mask = g->mask; h.xfeat = g->xfeat; h.pose_begin = m->d.pose_begin;
                    return h;
                };
                auto encode_at = [&](const float* xe) {          // pose embeddings of xe -> g->pemb
                    NodeArgs a = node_args(m, g);
                    a.src = 2; a.step = STEP_NONE; a.do_encode = 1; a.x_in = xe;
                    launch_node<H>(m, g, a, s);
                };
                {
                    HmcArgs h = hargs(HMC_MOMENTUM);
                    if (noise_for(L, call0[t] + 1, h.noise)) return 1;
                    hipLaunchKernelGGL(k_hmc, hgrid, dim3(256), 0, s, h);
                }
                for (int e = 0; e < S; ++e) {
                    HmcArgs r = hargs(HMC_REFRESH);
                    if (noise_for(L, call0[t] + 2 + (uint64_t)e, r.noise)) return 1;
                    hipLaunchKernelGGL(k_hmc, hgrid, dim3(256), 0, s, r);
                    const float m_i = 9.0f * m->betas[e];
                    for (int lf = 0; lf < 2; ++lf) {
                        // (the reference re-evaluates the gradient at an unchanged x between leapfrogs; it is
                        // deterministic, so the evaluation after LEAP_A serves both half steps around it)
                        if (lf == 0) { encode_at(g->xhat); if (launch_eval_energy<H>(m, g, e, g->xhat, true, E_hat, s)) return 1; }
                        HmcArgs a = hargs(HMC_LEAP_A);
                        a.ss_i = m->step[e]; a.md_i = m_i * m_i; a.kap_i = m->kappa[e];
                        hipLaunchKernelGGL(k_hmc, hgrid, dim3(256), 0, s, a);
                        encode_at(g->xhat);
                        if (launch_eval_energy<H>(m, g, e, g->xhat, true, E_hat, s)) return 1;
                        HmcArgs b = a;
                        b.mode = HMC_LEAP_B;
                        hipLaunchKernelGGL(k_hmc, hgrid, dim3(256), 0, s, b);
                    }
                    encode_at(g->x);
                    if (launch_eval_energy<H>(m, g, t, g->x, false, E_x, s)) return 1;
                    encode_at(g->xhat);
                    if (launch_eval_energy<H>(m, g, t, g->xhat, false, E_hat, s)) return 1;
                    HmcArgs c = hargs(HMC_ACCEPT);
                    c.E_x = E_x; c.E_hat = E_hat; c.acc_count = g->acc_count + t;
                    c.margin = margin_at(g, ucall0[t] + (uint64_t)e - ucall0[t_first]);
                    c.reset_mask = (e == S - 1);
                    c.hist = e == S - 1 ? hist_at(L, T - t) : nullptr;
                    c.noise.mode = nz->mode; c.noise.seed = nz->seed; c.noise.row_offset = nz->row_offset;
                    const uint64_t uc = ucall0[t] + (uint64_t)e;
                    c.noise.ucall = (unsigned int)uc;
                    if (nz->mode == CCSP_NOISE_INJECTED) {
                        if (!nz->uniform || uc < nz->ucall_base || uc - nz->ucall_base >= nz->n_uniform)
                            return fail("chain_run: injected uniform stream exhausted at call %llu", (unsigned long long)uc);
                        c.noise.uniform = nz->uniform + (size_t)(uc - nz->ucall_base) * N;
                    }
                    hipLaunchKernelGGL(k_hmc, hgrid, dim3(256), 0, s, c);
                }
                encode_at(g->x);          // pose embeddings of the state for the next timestep's p_sample
                continue;
            }
            // MALA reuse: from the second inner step on, the gradient evaluation at x is skipped on the device when the previous
            // accept step moved nothing (the kernels read g->mala_changed: reset by the propose step, += accepted nodes by accept)
            bool reuse = false;
            if constexpr (H == 256)
                reuse = sampler == CCSP_SAMPLER_MALA && m->mala_reuse && m->f16x2 && m->energy_bwd_h2 && m->WpTH && m->pe2_wTH &&
                        !m->valu_node_energy &&      // (k_node_energy<H> has no skip prologue)
                        !g->profile;        // (a profiled chain times every kernel at full work)
            // with a shard hook the kernels write the shard's own energies to Escal[2..3]; a copy of them goes through the hook
            // (Escal[0..1], reduced in place) every inner step, so a skipped evaluation leaves the LOCAL E(x) standing
            const bool hook = sampler == CCSP_SAMPLER_MALA && (m->energy_hook != nullptr || m->rccl_comm != nullptr);
            float* E_xl = hook ? g->Escal + 2 : E_x;
            float* E_hatl = hook ? g->Escal + 3 : E_hat;
            for (int e = 1; e <= S; ++e) {
                // the MALA-reuse flags: the accept step of inner step e counts the pose elements it moved in word e & 1 (reset by the same
                // step's update kernel), the gradient evaluation of step e + 1 reads it -- two words, so the update that runs in the
                // evaluation's last kernel resets a word no block of that kernel reads
                const int* skip_flag = (reuse && e >= 2) ? g->mala_changed + ((e - 1) & 1) : (const int*)nullptr;
                NodeArgs a = node_args(m, g);
                a.src = 1; a.eps_buf = g->eps; a.do_encode = 1; a.xhat = g->xhat;
                sched(a, t);
                if (noise_for(L, call0[t] + (uint64_t)e, a.noise)) return 1;
                if (sampler != CCSP_SAMPLER_MALA) {
                    a.step = STEP_ULA;
                    a.reset_mask = (e == S);
                    a.hist = e == S ? hist_at(L, T - t) : nullptr;
                } else {
                    a.step = STEP_MALA_PROPOSE;
                    a.changed = reuse ? g->mala_changed + (e & 1) : nullptr;
                }
                bool tail_done = false;
                if (before_rewrite()) return 1;
                if (launch_eval_energy<H>(m, g, t, g->x, true, E_xl, s, skip_flag, nullptr, 0, &a, &tail_done)) return 1;
                if (!tail_done) launch_node<H>(m, g, a, s);                   // (MALA: x_hat, and its pose embedding)
                if (sampler != CCSP_SAMPLER_MALA) continue;
                // without a shard hook the accept kernel sums the proposal's energy partials itself (no k_energy_sum launch)
                const bool fold_sum = !hook && g->plan.E_act > 0;
                if (launch_eval_energy<H>(m, g, t, g->xhat, false, fold_sum ? (float*)nullptr : E_hatl, s)) return 1;
                // global-batch mode: E(x), E(x_hat) of this shard -> sums over all shards (the reference's energies are
                // one scalar for the WHOLE batch, ddpm.py:1026-1038); the hook enqueues the reduction on the chain's stream
                if (hook) {
                    HIP_TRY(hipMemcpyAsync(g->Escal, g->Escal + 2, 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
                    if (m->rccl_comm) {      // {E(x), E(x_hat)} of this shard -> sums over the communicator's ranks, enqueued on the chain's own stream
                        RcclApi* ra = rccl_api();
                        const int rc = ra ? ra->all_reduce(g->Escal, g->Escal, 2, 7 /*ncclFloat32*/, 0 /*ncclSum*/, m->rccl_comm, s) : -1;
                        if (rc != 0) return fail("chain_run: ncclAllReduce of the batch energies failed: %s", rccl_err(ra, rc));
                    } else if (m->energy_hook(m->energy_hook_ctx, g->Escal, (void*)s)) return fail("chain_run: the energy hook failed");
                }
                NodeArgs b = node_args(m, g);
                b.src = 1; b.eps_buf = g->eps; b.do_encode = 1; b.xhat = g->xhat; b.step = STEP_MALA_ACCEPT;
                b.E_x = E_x; b.E_hat = E_hat; b.acc_count = g->acc_count + t;
                b.changed = reuse ? g->mala_changed + (e & 1) : nullptr;
                b.margin = margin_at(g, ucall0[t] + (uint64_t)(e - 1) - ucall0[t_first]);
                if (fold_sum) { b.E_hat_partial = g->partial; b.n_hat_partial = g->n_part_last; }
                if (cp) {                                // the other lane's energies: wait until its forward pass at the proposal is behind us
                    cp->n_hat[me] = g->n_part_last;
                    HIP_TRY(hipEventRecord(cp->evF[me], s));
                    cp->recF[me].store(kstep + 1, std::memory_order_release);
                    if (spin(cp->recF[other], kstep + 1)) return 1;
                    HIP_TRY(hipStreamWaitEvent(s, cp->evF[other], 0));
                    b.E_x2 = cp->E_x[other]; b.E_hat_partial2 = cp->hat_partial[other]; b.n_hat_partial2 = cp->n_hat[other];
                    b.couple_second = me;
                }
                b.reset_mask = (e == S);
                b.hist = e == S ? hist_at(L, T - t) : nullptr;
                sched(b, t);
                // (uniform draws are indexed by the GLOBAL node row: a lane's rows start at node0)
                b.noise.mode = nz->mode; b.noise.seed = nz->seed; b.noise.row_offset = nz->row_offset + (unsigned long long)L.node0;
                const uint64_t uc = ucall0[t] + (uint64_t)(e - 1);
                b.noise.ucall = (unsigned int)uc;
                if (nz->mode == CCSP_NOISE_INJECTED) {
                    if (!nz->uniform || uc < nz->ucall_base || uc - nz->ucall_base >= nz->n_uniform)
                        return fail("chain_run: injected uniform stream exhausted at call %llu", (unsigned long long)uc);
                    b.noise.uniform = nz->uniform + (size_t)(uc - nz->ucall_base) * N_total + (size_t)L.node0;
                }
                launch_node<H>(m, g, b, s);
                if (cp) {
                    HIP_TRY(hipEventRecord(cp->evA[me], s));
                    cp->recA[me].store(kstep + 1, std::memory_order_release);
                    owe_wait = true;
                }
                ++kstep;
            }
        }
        // (coupled lanes: the caller adds both lanes' counts -- ccsp_chain_run)
        if (accept && !cp) hipLaunchKernelGGL(k_accept_rates, dim3(nblk(T, 256)), dim3(256), 0, s, T, g->acc_count, g->acc_denom, accept);
#ifdef CCSP_EXPERIMENTS
    } else if (m->graph_mode && lanes.size() == 1 && lanes[0].g->N < 512 && m->bf16x3 && m->d.model_kind == CCSP_MODEL_DIFFUSION_CCSP &&
               !lanes[0].g->profile && lanes[0].g->plan.E_act > 0 && t_first >= t_last) {
        // hipGraph mode (opt-in): a small batch is three short dependent launches per evaluation.  One graph of
        // (1 + S) evaluations per distinct S is captured once per ccsp_graph and replayed for every timestep; what
        // differs between evaluations is in the device step table (StepEntry), filled here for this chain.
        const Lane& L = lanes[0];
        ccsp_graph* g = L.g;
        hipStream_t s = L.s;
        size_t n_ent = 0;
        for (int t = t_first; t >= t_last; --t) n_ent += 1 + (size_t)steps_at(m, sampler, t);
        HIP_TRY(hipStreamSynchronize(s));                       // a previous chain may still be reading the host copies
        if (n_ent > g->tab_cap) {
            if (dev_alloc(g->allocs, &g->d_tab, n_ent)) return 1;
            g->tab_cap = n_ent;
        }
        if (!g->d_hdr && (dev_alloc(g->allocs, &g->d_hdr, 1) || dev_alloc(g->allocs, &g->d_counter, 1))) return 1;
        g->h_tab.resize(n_ent);
        size_t k = 0;
        for (int t = t_first; t >= t_last; --t) {
            const int S = steps_at(m, sampler, t);
            for (int e = 0; e <= S; ++e) {
                NodeArgs a;
                sched(a, t);
                NoiseArg na;
                if (noise_for(L, call0[t] + (uint64_t)e, na)) return 1;          // (bounds check of an injected stream)
                StepEntry& en = g->h_tab[k++];
                en.t = t; en.step = e == 0 ? STEP_ANCESTRAL : STEP_ULA; en.reset_mask = (e == S); en.hist_slot = e == S ? T - t : -1;
                en.call = (unsigned int)(call0[t] + (uint64_t)e);
                en.a_t = a.a_t; en.b_t = a.b_t; en.c1 = a.c1; en.c2 = a.c2; en.sigma = a.sigma; en.kappa = a.kappa; en.ss = a.ss; en.std_ = a.std_;
            }
        }
        ChainHeader& hd = g->h_hdr;
        memset(&hd, 0, sizeof(hd));
        hd.seed = nz->seed; hd.row_offset = nz->row_offset + (unsigned long long)L.node0; hd.call_base = nz->call_base; hd.np_total = NP_total;
        hd.hist = history ? history + (size_t)L.node0 * P : nullptr;
        hd.normal = nz->mode == CCSP_NOISE_INJECTED ? nz->normal + (size_t)L.node0 * P : nullptr;
        hd.noise_mode = nz->mode;
        HIP_TRY(hipMemcpyAsync(g->d_tab, g->h_tab.data(), n_ent * sizeof(StepEntry), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(g->d_hdr, &hd, sizeof(hd), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(g->d_counter, 0, sizeof(int), s));
        for (int t = t_first; t >= t_last; --t) {
            const int S = steps_at(m, sampler, t);
            auto it = g->execs.find(S);
            if (it == g->execs.end()) {
                hipGraph_t graph = nullptr;
                hipGraphExec_t exec = nullptr;
                // captured on a stream of our own: the caller's may be the legacy default stream, which cannot capture
                if (!m->capture_stream) HIP_TRY(hipStreamCreateWithFlags(&m->capture_stream, hipStreamNonBlocking));
                hipStream_t cs = m->capture_stream;
                HIP_TRY(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
                int rc = 0;
                for (int e = 0; e <= S && !rc; ++e) {
                    rc = launch_eval<H>(m, g, 0, cs, true);
                    NodeArgs a = node_args(m, g);
                    a.src = 0; a.do_encode = 1; a.step = STEP_ULA;
                    a.tab = g->d_tab; a.counter = g->d_counter; a.hdr = g->d_hdr;
                    launch_node<H>(m, g, a, cs);
                }
                const hipError_t ce = hipStreamEndCapture(cs, &graph);
                if (rc || ce != hipSuccess) return rc ? 1 : fail("chain_run: hipStreamEndCapture failed: %s", hipGetErrorString(ce));
                const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                if (ie != hipSuccess) return fail("chain_run: hipGraphInstantiate failed: %s", hipGetErrorString(ie));
                g->evals -= 1 + S;                                  // (counted by launch_eval during the capture)
                it = g->execs.emplace(S, exec).first;
            }
            HIP_TRY(hipGraphLaunch(it->second, s));
            g->evals += 1 + S;
        }
#endif
    } else {
#ifdef CCSP_EXPERIMENTS
        // the node update rides in the edge kernel's tail when the f16x2 kernels run with 16- / 32-edge tiles (FuseArgs)
        for (const Lane& L : lanes) {
            ccsp_graph* g = L.g;
            bool can = false;
            bool can2 = false;
            if constexpr (H == 256) {
                can2 = m->fuse_node == 2 && m->f16x2 && m->bf16x3 && m->pe2_wH && !m->node_generic && m->d.model_kind == CCSP_MODEL_DIFFUSION_CCSP &&
                       !m->d.energy_wrapper && g->plan.E_act > 0 && !g->profile;
                can = m->fuse_node == 1 && m->f16x2 && m->bf16x3 && m->pe2_wH && !m->node_generic && m->d.model_kind == CCSP_MODEL_DIFFUSION_CCSP &&
                      !m->d.energy_wrapper && g->plan.E_act > 0 && edge_tile_edges(m, g->plan.E_act) <= 32 &&
                      nblk(g->plan.E_act, edge_tile_edges(m, g->plan.E_act)) <= 2 * m->ncu;
            }
            if (can2 && fuse2_prepare(m, g, L.s)) return 1;
            g->ng_use = can2 && g->ng_wgs > 0;
            if (can) {
                if (fuse_prepare(m, g, edge_tile_edges(m, g->plan.E_act), L.s)) return 1;
                HIP_TRY(hipMemsetAsync(g->fuse_count, 0, (size_t)g->fuse_blocks * sizeof(unsigned int), L.s));
                g->fuse_epoch = 0;
            } else {
                g->fuse_me = 0;
            }
        }
#endif
        // relay mode (Gate): lanes whose three grids fit their share of the chip's workgroup slots all at once
        std::vector<Relay> relay(lanes.size());
        if constexpr (H == 256)
            for (size_t li = 0; li < lanes.size(); ++li)
                if (relay_begin(m, lanes[li], &relay[li])) return 1;
        for (int t = t_first; t >= t_last; --t) {
            const int S = steps_at(m, sampler, t);
            for (int e = 0; e <= S; ++e) {
                for (size_t li = 0; li < lanes.size(); ++li) {
                    const Lane& L = lanes[li];
                    ccsp_graph* g = L.g;
                    NodeArgs a = node_args(m, g);
                    a.do_encode = 1;
                    a.step = e == 0 ? STEP_ANCESTRAL : STEP_ULA;
                    a.reset_mask = (e == S);
                    a.hist = e == S ? hist_at(L, T - t) : nullptr;
                    sched(a, t);
                    if (noise_for(L, call0[t] + (uint64_t)e, a.noise)) return 1;
                    if constexpr (H == 256) {
                        Relay& r = relay[li];
                        if (r.on) {
                            // row GEMM i+1 waits for node update i, edge kernel i for row GEMM i, node update i for edge kernel i; every
                            // buffer of an evaluation is dead before its next writer passes its gate (the waits form one cycle)
                            unsigned int* c = g->relay_ctr;
                            StepRef ref{nullptr, nullptr, nullptr, Gate{r.ev ? c + 2 : nullptr, r.ev * r.nN, c + 0, c + 3}};
                            const size_t tau_stride = (size_t)m->d.n_types * 2 * H;
                            launch_rowgemm_h2(m, g, m->tau + (size_t)t * tau_stride, ref, tau_stride, L.s);
                            EdgeEnergyArgs en{};
                            en.gate = Gate{c + 0, (r.ev + 1) * r.nR, c + 1, c + 3};
                            launch_edge_h2<false>(m, g, en, nullptr, r.sE, nullptr);
                            a.src = 0;
                            a.gate = Gate{c + 1, (r.ev + 1) * r.nE, c + 2, c + 3};
                            launch_node<H>(m, g, a, r.sN);
                            r.ev++;
                            g->evals++;
                            continue;
                        }
                    }
                    bool fused = false;
                    if (m->d.model_kind == CCSP_MODEL_STRUCT_DIFFUSION) {
                        if (launch_eval_sd<H>(m, g, t, L.s)) return 1;
                        a.src = 1; a.eps_buf = g->eps;
                    } else {
                        a.src = 0;
                        NoiseAhead na{};
                        if constexpr (H == 256) {
                            if (m->f16x2 && m->bf16x3 && nz->mode != CCSP_NOISE_INJECTED && g->plan.E_act > 0) {
                                if (!g->zbuf && dev_alloc(g->allocs, &g->zbuf, (size_t)g->N * P)) return 1;
                                na.z = g->zbuf; na.N = g->N; na.P = P; na.blocks = nblk((long)g->N * P, 256);
                                na.call = a.noise.call; na.seed = a.noise.seed; na.row_offset = a.noise.row_offset;
                                a.noise.mode = CCSP_NOISE_INJECTED;          // the node update reads the draws the row GEMM's extra workgroups wrote
                                a.noise.normal = g->zbuf;
                            }
                        }
                        if (launch_eval<H>(m, g, t, L.s, false, (g->fuse_me > 0 || g->ng_use) ? &a : nullptr, &fused, na.z ? &na : nullptr)) return 1;
                    }
                    if (!fused) launch_node<H>(m, g, a, L.s);
                }
            }
        }
        for (size_t li = 0; li < lanes.size(); ++li)
            if (relay[li].on && relay_end(m, lanes[li], relay[li])) return 1;
    }
    for (const Lane& L : lanes)
        HIP_TRY(hipMemcpyAsync(x_io + (size_t)L.node0 * P, L.g->x, (size_t)L.g->N * P * sizeof(float), hipMemcpyDeviceToDevice, L.s));
    HIP_TRY(hipGetLastError());
    return 0;
}

int graph_build(ccsp_model* m, int N, int E, int F, const float* x, const signed char* mask, std::vector<int64_t>&& ei,
                std::vector<float>&& ea, hipStream_t s, ccsp_graph** out);

// cut the batch into `want` contiguous node ranges that no edge crosses (graphs are independent
// units: collation is block-diagonal) and build one child graph per range
int sequences_build(ccsp_graph* g, int B, int b0, const std::vector<int>& cnt_all, const std::vector<int>& graph_of, const std::vector<int>& pos_of, hipStream_t s);

int ensure_children(ccsp_model* m, ccsp_graph* g, int want, hipStream_t s) {
    if (g->lanes_tried) return 0;
    g->lanes_tried = 1;
    const int N = g->N, E = g->E;
    if (want < 2 || N < 2 * want) return 0;
    const bool sd = m->d.model_kind == CCSP_MODEL_STRUCT_DIFFUSION;
    if (sd) {                                            // lanes are cut between graphs: the nodes of a graph must be contiguous, graphs ascending
        if (!g->seq_ready) return 0;
        for (int n = 1; n < N; ++n) if (g->h_seq_graph[n] < g->h_seq_graph[n - 1]) return 0;
    }
    std::vector<int> cross(N + 1, 0);                    // cross[i] > 0: some edge spans the boundary before node i
    for (int e = 0; e < E; ++e) {
        const int a = (int)g->h_ei[e], b = (int)g->h_ei[(size_t)E + e];
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        if (hi > lo) { cross[lo + 1]++; cross[hi + 1]--; }
    }
    std::vector<int> cuts;
    cuts.push_back(0);
    int run = 0;
    std::vector<char> ok(N + 1, 0);
    for (int i = 1; i < N; ++i) { run += cross[i]; ok[i] = run == 0 && (!sd || g->h_seq_graph[i] != g->h_seq_graph[i - 1]); }
    for (int k = 1; k < want; ++k) {
        const int target = (int)((long)N * k / want);
        int best = -1;
        for (int d = 0; d < N; ++d) {
            if (target - d > cuts.back() && target - d < N && ok[target - d]) { best = target - d; break; }
            if (target + d > cuts.back() && target + d < N && ok[target + d]) { best = target + d; break; }
        }
        if (best < 0) return 0;                          // no valid cut: run as one lane
        cuts.push_back(best);
    }
    cuts.push_back(N);
    for (size_t k = 0; k + 1 < cuts.size(); ++k) {
        const int n0 = cuts[k], n1 = cuts[k + 1];
        std::vector<int64_t> a_, b_;
        std::vector<float> ea;
        for (int e = 0; e < E; ++e) {
            const int64_t a = g->h_ei[e], b = g->h_ei[(size_t)E + e];
            if (a >= n0 && a < n1) { a_.push_back(a - n0); b_.push_back(b - n0); ea.push_back(g->h_ea[e]); }
        }
        std::vector<int64_t> ei(a_);
        ei.insert(ei.end(), b_.begin(), b_.end());
        ccsp_graph* c = nullptr;
        if (m->lane_streams.size() <= k) {
            hipStream_t cs = nullptr;
            hipEvent_t ce = nullptr;
            // CCSP_LANE_CUMASK (experiment, default off): give every lane its own share of the compute units instead of letting the
            // lanes' kernels interleave on all of them; 1 = contiguous ranges of the mask, 2 = every want-th bit
            const char* cm = exp_env("CCSP_LANE_CUMASK");
            const int cmode = cm ? atoi(cm) : 0;
            if (cmode == 1 || cmode == 2) {
                const int ncu = m->ncu > 0 ? m->ncu : 256;
                std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
                for (int i = 0; i < ncu; ++i) {
                    const bool mine = cmode == 1 ? (i * want / ncu == (int)k) : (i % want == (int)k);
                    if (mine) mask[i >> 5] |= 1u << (i & 31);
                }
                HIP_TRY(hipExtStreamCreateWithCUMask(&cs, (uint32_t)mask.size(), mask.data()));
            } else if (lane_stream_get(k, &cs)) return 1;
            HIP_TRY(hipEventCreateWithFlags(&ce, hipEventDisableTiming));
            m->lane_stream_owned.push_back((cmode == 1 || cmode == 2) ? 1 : 0);
            m->lane_streams.push_back(cs);
            m->lane_events.push_back(ce);
        }
        if (graph_build(m, n1 - n0, (int)ea.size(), g->F, g->xfeat + (size_t)n0 * g->F, g->mask + n0, std::move(ei), std::move(ea), s, &c)) return 1;
        g->children.push_back(c);
        g->child_node0.push_back(n0);
        if (sd) {
            const int b0 = g->h_seq_graph[n0], b1 = g->h_seq_graph[n1 - 1] + 1;
            std::vector<int> graph_of(n1 - n0), pos_of;
            for (int n = n0; n < n1; ++n) graph_of[n - n0] = g->h_seq_graph[n] - b0;
            if (!g->h_seq_pos.empty()) pos_of.assign(g->h_seq_pos.begin() + n0, g->h_seq_pos.begin() + n1);
            if (sequences_build(c, b1 - b0, b0, g->h_seq_cnt, graph_of, pos_of, s)) return 1;
        }
    }
    if (!m->fork_event) HIP_TRY(hipEventCreateWithFlags(&m->fork_event, hipEventDisableTiming));
    return 0;
}

