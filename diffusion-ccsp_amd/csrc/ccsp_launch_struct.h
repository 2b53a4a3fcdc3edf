// ccsp_launch_struct.h -- launch of one evaluation of the StructDiffusion transformer baseline (kernels: ccsp_struct.h).
// A fragment of the ONE translation unit csrc/ccsp_hip.hip (included there, at this position, inside its namespaces): not a standalone header.
// ---- StructDiffusion baseline ------------------------------------------------------------------
template <int EPI>
void sd_gemm(int M, int K, int N, const float* A, const float* W, const float* b, float* Cm, hipStream_t s) {
    const int rt = nblk(M, TILE_M);
    if (N % 128 == 0 && (long)rt * (N / 128) >= 512)
        hipLaunchKernelGGL((k_sd_gemm<2, EPI>), dim3(rt * (N / 128)), dim3(256), 0, s, M, K, N, A, W, b, Cm);
    else
        hipLaunchKernelGGL((k_sd_gemm<1, EPI>), dim3(rt * (N / 64)), dim3(256), 0, s, M, K, N, A, W, b, Cm);
}

constexpr int SD_KSPLIT = 4;       // most K slices of the c_proj GEMM (sdY holds that many partial products); used: 2 (r04 A/B: 453 us per evaluation against 473 with 4, 477 with 1)

// returns the number of K slices written (1: Cm is the result; > 1: partial products [slices][M][N], summed by the LayerNorm kernel that reads them)
template <int EPI>
int sd_gemm_h2(const ccsp_model* m, int M, int K, int N, const float* A, const unsigned int* amax, const unsigned short* WH, int w_exp, const float* b, float* Cm,
               unsigned int* cmax, hipStream_t s, bool may_split = false, bool split2 = false /*the consumer adds two K slices whatever the shape (in_proj -> k_sd_attn)*/) {
    // 64-column tiles when the 128-column tile list would not give every CU two workgroups (the N = Wd GEMMs of a 256-graph batch)
    static const int force_tn = exp_env("CCSP_SD_TN") ? atoi(exp_env("CCSP_SD_TN")) : 0;
    static const int force_ks = exp_env("CCSP_SD_KSPLIT") ? atoi(exp_env("CCSP_SD_KSPLIT")) : -1;
    const bool tn64 = force_tn ? force_tn == 64 : (long)nblk(M, 64) * (N / 128) < 2L * m->ncu;
    int ks = 1;
    // (chosen from K and N alone: the same batch run as one lane or as two adds the same partial products in the same order)
    if (may_split && EPI == SD_EPI_BIAS && cmax == nullptr && K % (64 * SD_KSPLIT) == 0 && K >= 4 * N) ks = 2;
    if (may_split && force_ks >= 1 && K % (64 * force_ks) == 0 && force_ks <= SD_KSPLIT) ks = force_ks;
    // (opt-in, CCSP_SD_INSPLIT=1: measured SLOWER, 54.6 against 57.0 samples/s in one call -- the two lanes' in_proj already give the chip
    // three workgroups per CU, and the attention kernel reads twice the bytes)
    static const bool insplit = exp_env("CCSP_SD_INSPLIT") && atoi(exp_env("CCSP_SD_INSPLIT")) == 1;
    if (split2 && insplit && EPI == SD_EPI_BIAS && cmax == nullptr && K % 128 == 0) ks = 2;
#ifdef CCSP_EXPERIMENTS
    // round 6: 128 x 128 tiles (k_sd_gemm_h2w) when the batch has the rows for them -- the tile list of a 2048-row batch is then 192 (in_proj),
    // 256 (c_fc), 4 x 64 (c_proj as four K slices) workgroups of four times the products per operand byte; out_proj (64 tiles) keeps the narrow form
    const int wide_env = m->sd_tile;              // (CCSP_SD_TILE=narrow | wide at model creation; -1: the rule below)
    // (chosen from K, N and a row count that a lane of a split batch always has -- lanes exist from 1024 token rows on, 512 per lane: the same batch
    //  run as one lane or as two takes the same kernels and adds the same K slices in the same order, test_struct_diffusion_lanes_are_bitwise_identical)
    if (N % 128 == 0 && K % 64 == 0 && M >= 384 && wide_env >= 1) {        // (opt-in: measured slower than the 64-row tiles, ccsp_struct.h; 2 = the eight-wave form)
        int ksw = 1;
        if (may_split && EPI == SD_EPI_BIAS && cmax == nullptr && K % (64 * SD_KSPLIT) == 0 && K >= 4 * N) ksw = SD_KSPLIT;
        if ((long)(N / 128) * ksw >= 12) {          // (12+ column tiles x K slices: in_proj, c_fc, c_proj; out_proj's four stay narrow)
            if (m->sd_tile == 2) hipLaunchKernelGGL((k_sd_gemm_h2x<EPI>), dim3(nblk(M, 128) * (N / 128), ksw), dim3(512), 0, s, M, K, N, A, amax, WH, (size_t)N * K, w_exp, b, Cm, cmax);
            else if (m->sd_pipe) hipLaunchKernelGGL((k_sd_gemm_h2w<EPI, true>), dim3(nblk(M, 128) * (N / 128), ksw), dim3(256), 0, s, M, K, N, A, amax, WH, (size_t)N * K, w_exp, b, Cm, cmax);
            else hipLaunchKernelGGL((k_sd_gemm_h2w<EPI, false>), dim3(nblk(M, 128) * (N / 128), ksw), dim3(256), 0, s, M, K, N, A, amax, WH, (size_t)N * K, w_exp, b, Cm, cmax);
            return ksw;
        }
    }
#endif
    // operands requested 2 chunks ahead; 4 (CCSP_SD_PD=4) when the slice is a multiple of 4 chunks
    const dim3 gr64(nblk(M, 64) * (N / 64), ks), gr128(nblk(M, 64) * (N / 128), ks);
#ifdef CCSP_EXPERIMENTS
    static const int kstep_env = getenv("CCSP_SD_KSTEP") ? atoi(getenv("CCSP_SD_KSTEP")) : 1;
    const bool ks2 = kstep_env == 2 && (K / ks) % 128 == 0;
    static const int force_pd = getenv("CCSP_SD_PD") ? atoi(getenv("CCSP_SD_PD")) : 0;
    const bool pd4 = (K / ks) % 128 == 0 && force_pd == 4;      // (r04 A/B at 2048 token rows: 4 ahead 437 us per evaluation, 2 ahead 428 -- the chunk is not waiting for loads)
    if (tn64 && pd4) hipLaunchKernelGGL((k_sd_gemm_h2<EPI, 64, 4>), gr64, dim3(256), 0, s, M, K, N, A, amax, WH, (size_t)N * K, w_exp, b, Cm, cmax);
    else if (pd4) hipLaunchKernelGGL((k_sd_gemm_h2<EPI, 128, 4>), gr128, dim3(256), 0, s, M, K, N, A, amax, WH, (size_t)N * K, w_exp, b, Cm, cmax);
    else
#endif
#ifdef CCSP_EXPERIMENTS
    // (round 6, CCSP_SD_KSTEP=2: two chunks per step of the K loop -- half the barriers and waits.  SLOWER in one call: in_proj / c_proj 23.9 -> 25.9 us,
    //  c_fc 26.6 -> 35.6 us (its 96 KB of stages leave one workgroup per CU), 56.3 / 58.1 -> 51.8 samples/s.  The loop is not a chain of exposed
    //  latencies: per chunk and CU the four operand requests of every wave (one address unit: ~25 cycles each) and 88 LDS cycles per wave (fragment
    //  reads of 32 x 32 wave tiles + the staging stores) each add up to the ~1040 cycles a chunk takes with three workgroups per CU; profiles/r06_findings.md)
    if (tn64 && ks2) hipLaunchKernelGGL((k_sd_gemm_h2<EPI, 64, 2, 2>), gr64, dim3(256), 0, s, M, K, N, A, amax, WH, (size_t)N * K, w_exp, b, Cm, cmax);
    else if (ks2) hipLaunchKernelGGL((k_sd_gemm_h2<EPI, 128, 2, 2>), gr128, dim3(256), 0, s, M, K, N, A, amax, WH, (size_t)N * K, w_exp, b, Cm, cmax);
    else
#endif
    if (tn64) hipLaunchKernelGGL((k_sd_gemm_h2<EPI, 64, 2>), gr64, dim3(256), 0, s, M, K, N, A, amax, WH, (size_t)N * K, w_exp, b, Cm, cmax);
    else hipLaunchKernelGGL((k_sd_gemm_h2<EPI, 128, 2>), gr128, dim3(256), 0, s, M, K, N, A, amax, WH, (size_t)N * K, w_exp, b, Cm, cmax);
    return ks;
}

// one evaluation of the transformer at the poses whose embeddings are in g->pemb; result -> g->eps
template <int H>
int launch_eval_sd(ccsp_model* m, ccsp_graph* g, int t, hipStream_t s) {
    if (!g->seq_ready) return fail("StructDiffusion: call ccsp_graph_set_sequences (batch.batch) before evaluating");
    const int M = g->sd_M, Wd = m->Wd, P = m->d.pose_dim;
    prof_mark(g, s, CCSP_K_SD_EVAL);
    unsigned int* const nomax = nullptr;
    unsigned int *mY = g->sdMax, *mA = g->sdMax + M, *mX = g->sdMax + 2 * (size_t)M, *mF = g->sdMax + 3 * (size_t)M;
    // (f16x2 path: ln_1 of the first block runs in the embedding kernel, the last block's ln_2 in the decoder kernel: 26 kernel launches per evaluation, 28 in round 4)
    static const bool sd_fuse_off = getenv("CCSP_SD_FUSE") && atoi(getenv("CCSP_SD_FUSE")) == 0;
    const bool fuse_ends = m->sd_h2 && !sd_fuse_off;
#define CCSP_SD_EMBED(NVV)                                                                                                                           \
        hipLaunchKernelGGL(k_sd_embed<NVV>, dim3(nblk(M, 4)), dim3(256), 0, s, M, H, Wd, m->d.grasp_dim > 0 ? 1 : 0, g->tok_node, g->tok_pos, g->gemb,   \
                           g->remb, g->pemb, m->temb + (size_t)t * H, m->sd_pe, m->lnpre_g, m->lnpre_b, g->sdX,                                        \
                           (const float*)m->sd[0].ln1_g, (const float*)m->sd[0].ln1_b, g->sdY, mY, mA, mX, mF)
    if (fuse_ends && Wd == 512) CCSP_SD_EMBED(8);                  // (hidden_dim 256: the width at compile time, no bounds tests next to the loads)
    else if (fuse_ends) CCSP_SD_EMBED(0);
#undef CCSP_SD_EMBED
    else
        hipLaunchKernelGGL(k_sd_embed<0>, dim3(nblk(M, 4)), dim3(256), 0, s, M, H, Wd, m->d.grasp_dim > 0 ? 1 : 0, g->tok_node, g->tok_pos, g->gemb,
                           g->remb, g->pemb, m->temb + (size_t)t * H, m->sd_pe, m->lnpre_g, m->lnpre_b, g->sdX,
                           (const float*)nullptr, (const float*)nullptr, (float*)nullptr, nomax, nomax, nomax, nomax);
    // LayerNorm kernels with the width at compile time (no bounds tests next to their loads) for the widths multiples of 128 give
    static const bool ln_generic = exp_env("CCSP_SD_LN") && !strcmp(exp_env("CCSP_SD_LN"), "generic");
    const int Wsel = ln_generic ? 0 : Wd;
    auto ln0 = [&](const float* X, const float* ga, const float* be, float* Y, unsigned int* ym, unsigned int* z0, unsigned int* z1, unsigned int* z2, int parts) {
        const dim3 gr(nblk(M, 4)), bl(256);
        switch (Wsel) {
            case 128: hipLaunchKernelGGL((k_sd_ln<0, 2>), gr, bl, 0, s, M, Wd, X, ga, be, Y, ym, z0, z1, z2, parts); break;
            case 256: hipLaunchKernelGGL((k_sd_ln<0, 4>), gr, bl, 0, s, M, Wd, X, ga, be, Y, ym, z0, z1, z2, parts); break;
            case 384: hipLaunchKernelGGL((k_sd_ln<0, 6>), gr, bl, 0, s, M, Wd, X, ga, be, Y, ym, z0, z1, z2, parts); break;
            case 512: hipLaunchKernelGGL((k_sd_ln<0, 8>), gr, bl, 0, s, M, Wd, X, ga, be, Y, ym, z0, z1, z2, parts); break;
            case 768: hipLaunchKernelGGL((k_sd_ln<0, 12>), gr, bl, 0, s, M, Wd, X, ga, be, Y, ym, z0, z1, z2, parts); break;
            default: hipLaunchKernelGGL((k_sd_ln<0, 0>), gr, bl, 0, s, M, Wd, X, ga, be, Y, ym, z0, z1, z2, parts);
        }
    };
    auto ln1 = [&](const float* X, const float* ga, const float* be, float* Y, int parts) {
        const dim3 gr(nblk(M, 4)), bl(256);
        switch (Wsel) {
            case 128: hipLaunchKernelGGL((k_sd_ln<1, 2>), gr, bl, 0, s, M, Wd, X, ga, be, Y, nomax, nomax, nomax, nomax, parts); break;
            case 256: hipLaunchKernelGGL((k_sd_ln<1, 4>), gr, bl, 0, s, M, Wd, X, ga, be, Y, nomax, nomax, nomax, nomax, parts); break;
            case 384: hipLaunchKernelGGL((k_sd_ln<1, 6>), gr, bl, 0, s, M, Wd, X, ga, be, Y, nomax, nomax, nomax, nomax, parts); break;
            case 512: hipLaunchKernelGGL((k_sd_ln<1, 8>), gr, bl, 0, s, M, Wd, X, ga, be, Y, nomax, nomax, nomax, nomax, parts); break;
            case 768: hipLaunchKernelGGL((k_sd_ln<1, 12>), gr, bl, 0, s, M, Wd, X, ga, be, Y, nomax, nomax, nomax, nomax, parts); break;
            default: hipLaunchKernelGGL((k_sd_ln<1, 0>), gr, bl, 0, s, M, Wd, X, ga, be, Y, nomax, nomax, nomax, nomax, parts);
        }
    };
    auto ln21 = [&](float* X, const float* g2, const float* b2, const float* g1, const float* b1, float* Y, unsigned int* ym, unsigned int* z0, unsigned int* z1,
                    unsigned int* z2, int parts) {
        const dim3 gr(nblk(M, 4)), bl(256);
        switch (Wsel) {
            case 128: hipLaunchKernelGGL((k_sd_ln2ln1<2>), gr, bl, 0, s, M, Wd, X, g2, b2, g1, b1, Y, ym, z0, z1, z2, parts); break;
            case 256: hipLaunchKernelGGL((k_sd_ln2ln1<4>), gr, bl, 0, s, M, Wd, X, g2, b2, g1, b1, Y, ym, z0, z1, z2, parts); break;
            case 384: hipLaunchKernelGGL((k_sd_ln2ln1<6>), gr, bl, 0, s, M, Wd, X, g2, b2, g1, b1, Y, ym, z0, z1, z2, parts); break;
            case 512: hipLaunchKernelGGL((k_sd_ln2ln1<8>), gr, bl, 0, s, M, Wd, X, g2, b2, g1, b1, Y, ym, z0, z1, z2, parts); break;
            case 768: hipLaunchKernelGGL((k_sd_ln2ln1<12>), gr, bl, 0, s, M, Wd, X, g2, b2, g1, b1, Y, ym, z0, z1, z2, parts); break;
            default: hipLaunchKernelGGL((k_sd_ln2ln1<0>), gr, bl, 0, s, M, Wd, X, g2, b2, g1, b1, Y, ym, z0, z1, z2, parts);
        }
    };
    int last_parts = 0;
    for (int l = 0; l < SD_LAYERS; ++l) {
        const ccsp_model::SdLayer& w = m->sd[l];
        if (m->sd_h2) {
            // row maxima travel with the activations: ln_1 stores those of its output and clears the three buffers this block accumulates
            // (from the second block on, ln_1 ran fused behind the previous block's ln_2: k_sd_ln2ln1)
            if (l == 0 && !fuse_ends) ln0(g->sdX, w.ln1_g, w.ln1_b, g->sdY, mY, mA, mX, mF, 1);
            // (CCSP_SD_INSPLIT=1: in_proj as two K slices -- twice the workgroups, each half the chain of chunks -- added by the attention kernel
            // while it loads them; measured slower, see sd_gemm_h2)
            const int qparts = sd_gemm_h2<SD_EPI_BIAS>(m, M, Wd, 3 * Wd, g->sdY, mY, w.in_wH, w.in_e, w.in_b, g->sdQKV, nomax, s, false, true);
            hipLaunchKernelGGL(k_sd_attn, dim3(g->sd_B * SD_HEADS), dim3(256), 0, s, Wd, g->sdQKV, g->mask_from, g->sdA, mA, qparts, (size_t)M * 3 * Wd);
            sd_gemm_h2<SD_EPI_RESID>(m, M, Wd, Wd, g->sdA, mA, w.out_wH, w.out_e, w.out_b, g->sdX, mX, s);
            sd_gemm_h2<SD_EPI_QGELU>(m, M, Wd, 4 * Wd, g->sdX, mX, w.fc_wH, w.fc_e, w.fc_b, g->sdF, mF, s);
            const int parts = sd_gemm_h2<SD_EPI_BIAS>(m, M, 4 * Wd, Wd, g->sdF, mF, w.proj_wH, w.proj_e, w.proj_b, g->sdY, nomax, s, true);
            static const bool no_ln21 = exp_env("CCSP_SD_LN21") && atoi(exp_env("CCSP_SD_LN21")) == 0;
            if (l + 1 < SD_LAYERS && no_ln21) {
                ln1(g->sdY, w.ln2_g, w.ln2_b, g->sdX, parts);
                ln0(g->sdX, m->sd[l + 1].ln1_g, m->sd[l + 1].ln1_b, g->sdY, mY, mA, mX, mF, 1);
            } else if (l + 1 < SD_LAYERS) ln21(g->sdX, w.ln2_g, w.ln2_b, m->sd[l + 1].ln1_g, m->sd[l + 1].ln1_b, g->sdY, mY, mA, mX, mF, parts);
            else if (fuse_ends) last_parts = parts;                 // (x + ln_2(y) of the last block: in k_sd_decode, for the rows it decodes)
            else ln1(g->sdY, w.ln2_g, w.ln2_b, g->sdX, parts);
            continue;
        }
        ln0(g->sdX, w.ln1_g, w.ln1_b, g->sdY, nomax, nomax, nomax, nomax, 1);
        sd_gemm<SD_EPI_BIAS>(M, Wd, 3 * Wd, g->sdY, w.in_w, w.in_b, g->sdQKV, s);
        hipLaunchKernelGGL(k_sd_attn, dim3(g->sd_B * SD_HEADS), dim3(256), 0, s, Wd, g->sdQKV, g->mask_from, g->sdA, nomax, 1, (size_t)0);
        sd_gemm<SD_EPI_RESID>(M, Wd, Wd, g->sdA, w.out_w, w.out_b, g->sdX, s);
        sd_gemm<SD_EPI_QGELU>(M, Wd, 4 * Wd, g->sdX, w.fc_w, w.fc_b, g->sdF, s);
        sd_gemm<SD_EPI_BIAS>(M, 4 * Wd, Wd, g->sdF, w.proj_w, w.proj_b, g->sdY, s);
        ln1(g->sdY, w.ln2_g, w.ln2_b, g->sdX, 1);
    }
#define CCSP_SD_DECODE(NVV)                                                                                                                          \
    hipLaunchKernelGGL((k_sd_decode<H, NVV>), dim3(nblk(g->N, 4)), dim3(256), 0, s, g->N, Wd, P, g->F, g->node_tok, g->sdX, m->lnpost_g, m->lnpost_b,   \
                       m->pd0_wT, m->pd0_b, m->pd2_w, m->pd2_b, g->xfeat, g->mask, g->eps,                                                           \
                       last_parts ? (const float*)g->sdY : (const float*)nullptr, last_parts, M,                                                      \
                       (const float*)m->sd[SD_LAYERS - 1].ln2_g, (const float*)m->sd[SD_LAYERS - 1].ln2_b)
    if (Wd == 512 && H == 256) CCSP_SD_DECODE(8); else CCSP_SD_DECODE(0);
#undef CCSP_SD_DECODE
    prof_mark(g, s, -1);
    g->evals++;
    return 0;
}

