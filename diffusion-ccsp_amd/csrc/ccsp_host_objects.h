// ccsp_host_objects.h -- host objects, part 2: ccsp_model and ccsp_graph (everything the library owns per model / per collated batch).
// A fragment of the ONE translation unit csrc/ccsp_hip.hip (included there, at this position, inside its namespaces): not a standalone header.
struct ccsp_model {
    ccsp_model_desc d;
    int K_in;
    int max_wgs;     // grid cap of the tile kernels (persistent loops); unlimited by default
    // device weights (library-owned copies)
    float *ge0_w, *ge0_b, *ge2_wT, *ge2_b;
    float *gr0_w, *gr0_b, *gr2_wT, *gr2_b;
    float *pe0_w, *pe0_b, *pe2_wT, *pe2_b, *pe2_wF;
    float *pd0_w, *pd0_b, *pd2_w, *pd2_b;
    float *pd0_wT;   // [H, H/2]  pose_decoder.0.weight transposed (k_edge_bwd)
    float *pe2_w;    // [H, H/2]  pose_encoder.2.weight as given (encoder backward)
    float* Wg;     // [C][2][2H][H]   geometry slices (slot 0 = node a, slot 1 = node b)
    float* Wr;     // [C][2][2H][H]   grasp slice in slot 0 (slot 1 unused) or nullptr
    float* Wp;     // [C][2][2H][H]   pose slices
    float* WpT;    // [C][2][H][2H]   their transposes (energy-mode backward)
    int lanes;     // concurrent sub-batch chains per ccsp_chain_run (direct mode), default 2
    int lane_min_edges;   // batches with fewer active edges run as one lane
    int sd_pipe = 1;      // (CCSP_SD_PIPE=0) k_sd_gemm_h2w without the staging interleaved between its MFMA pairs
    int sd_tile = -1;     // (CCSP_SD_TILE) StructDiffusion GEMM tiles: 0 = 64-row tiles everywhere (round 4), 1 = 128 x 128 wherever the shape allows, -1 = by shape
    int lane_min_tokens;  // StructDiffusion: batches with fewer token rows run as one lane
    std::vector<hipStream_t> lane_streams;   // taken from the process-wide pool (lane_stream_get): new HIP streams are expensive to create
    std::vector<char> lane_stream_owned;     // (1: created for this model alone -- the CU-mask experiment -- and destroyed with it)
    std::vector<hipEvent_t> lane_events;     // (hundreds of ms for the first few), graphs come and go
    hipEvent_t fork_event = nullptr;
    hipStream_t capture_stream = nullptr;    // hipGraph captures (the caller's stream may be the legacy default stream)
    int graph_mode;    // CCSP_GRAPH=1: small batches replay captured hipGraphs; default 0 -- measured no faster
    int edge_kernel;   // 2: k_edge_bf2 (default, H = 256); 1: k_edge_bf (CCSP_EDGE_KERNEL=1)
    int row_tile;  // 128: k_rowgemm_bf2 (default); 64: k_rowgemm_bf (CCSP_ROW_TILE=64)
    int bf16x3;    // 1: direct-mode GEMMs on the bf16 matrix cores with 3-way split operands (ccsp_bf16x3.h)
    unsigned short* WpS;    // [3][C][2][2H][H] bf16 planes of Wp
    unsigned short* Wd1S;   // [3][H/2][H]      bf16 planes of pose_decoder.0.weight
    unsigned short* Wd1TS;  // [3][H][H/2]      planes of its transpose (k_edge_bwd_bf)
    unsigned short* WpTS;   // [3][C][2][H][2H] planes of WpT (transpose row GEMM of the energy backward)
    int f16x2 = 0;          // 1: evaluation GEMMs on the f16 matrix cores with 2-way split, exactly scaled operands (ccsp_f16x2.h; H = 256)
    unsigned short* WpH = nullptr;    // [2][C][2][2H][H] fp16 planes of Wp * 2^wp_exp
    unsigned short* WpHI = nullptr;   // the same planes as [C][2][2H][H / 32][2][32]: the forward row GEMM's operand (k_interleave_planes)
    unsigned short* Wd1H = nullptr;   // [2][H/2][H]      fp16 planes of pose_decoder.0.weight * 2^wd_exp
    unsigned short* Wd1HI = nullptr;  // [H/2][H/32][2][32] the same planes chunk-interleaved: the edge kernels' B operand
    int wp_exp = 0, wd_exp = 0;
    unsigned short* WpF = nullptr;    // the planes of WpH in MFMA fragment order (k_pack_wp_frag): k_eval_fused reads them straight into registers
    unsigned short* Wd1F = nullptr;   // likewise pose_decoder.0.weight (k_pack_wd1_frag)
    int eval_fused = 0;               // CCSP_EVAL=fused: direct-mode evaluations as ONE launch with U kept in LDS (ccsp_fused.h: 1 = k_eval_fused4, two
                                      // 256-thread workgroups per CU; 2 = CCSP_EVAL=fused8, the persistent 512-thread form); split: two launches
    unsigned short* WpTH = nullptr;   // [2][C][2][H][2H] fp16 planes of WpT * 2^wp_exp (energy backward; energy_wrapper models only)
    unsigned short* Wd1TH = nullptr;  // [2][H][H/2]      fp16 planes of pose_decoder.0.weight^T * 2^wd_exp
    unsigned short *WpTHI = nullptr, *Wd1THI = nullptr;      // the two above chunk-interleaved ([row][K / 32][2][32]): what the backward kernels read
    float wd2_absmax = 0.0f;          // max |pose_decoder.2.weight| (row-exponent bound of k_edge_bwd_h2)
    float bwd_bound_c = 0.0f;         // 1.21 max|Wd2| max_n sum_j |Wd1[j, n]|: |g_z[k, s H + n]| <= bwd_bound_c sum_p |go[k, s, p]| (k_edge_bwd_h2<true>)
    int bwd_rowsum_fused = 1;         // (CCSP_ENERGY_ROWSUM=kernel turns it off) row sums of g_z inside the decoder backward, transpose GEMM on partial rows
    int bwd_generic_p = 0;            // (CCSP_ENERGY_BWD_P=generic) k_edge_bwd_h2 with the run-time pose_dim even where it is 4 (A/B runs)
    int mala_lanes = 1;               // (CCSP_MALA_LANES=2 turns it on) MALA on batches of >= lane_min_edges active edges as two coupled lanes (MalaCouple, ccsp_chain.h).
                                      // Built in round 6, correct (test_mala_two_coupled_lanes), and SLOWER in one call: C4 recomputing 164.5 -> 143.2 samples/s, with reuse
                                      // 317.7 -> 237.6 (profiles/r06_ab_c4_mala_lanes.txt): two cross-stream event waits per inner step cost more than the kernel tails they overlap
    hipEvent_t mala_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int edge_fb = 2;                  // (CCSP_EDGE_FB) energy mode's decoder: 2 = forward + backward in one kernel (k_edge_fb_h2, ccsp_edge_fb.h), 1 = the backward alone
                                      // on that kernel's 32-edge tiles (k_edge_bwd2_h2), 0 = round 4's k_edge_bwd_h2 on 64-edge blocks (four workgroups per block)
    int node_energy_fused = 1;        // (CCSP_ENERGY_NODE=split turns it off) k_node_energy_h2_update: the update that consumes the gradient in the same launch
    unsigned short* pe2_wH = nullptr; // pose_encoder.2.weight * 2^pe2_exp, fp16 planes in fragment order (k_pack_enc_frag_h2); CCSP_ENC=f32 leaves it null
    unsigned short* pe2_wTH = nullptr;    // the same tensor transposed, for the energy backward (k_pack_enc_frag_h2t; energy_wrapper models)
    int pe2_exp = 0;
    float pe0_c1 = 0.0f, pe0_c2 = 0.0f;   // bound of the pose encoder's layer-1 pre-activation: c1 max|x| + c2
    int energy_bwd_h2 = 1;            // CCSP_ENERGY_BWD=bf16x3 keeps the backward GEMMs on the six-product bf16 kernels
    int fuse_node = 0;                // CCSP_FUSE_NODE=1: fold the node update into the edge kernel's tail (FuseArgs).  Measured slower than
                                      // the separate launch (C2 467 -> 383, C5 250 -> 182 samples/s, profiles/r03_findings.md), so off by default
    int relay = 0;                    // CCSP_RELAY=1: relay mode for small batches (Gate)
    int node_generic = 0;             // CCSP_NODE=generic: k_node instead of k_node_direct in direct-mode chains (A/B runs)
    int node_stream = 0;              // CCSP_NODE=stream: k_node_direct_s
    int valu_node_energy = 0;         // CCSP_NODE_ENERGY_VALU: the pre-MFMA node-energy kernel (A/B runs; never combined with the reuse below)
    int mala_reuse = 1;               // (CCSP_MALA_REUSE=0 turns it off) an inner step that accepted NO node leaves x where it was, so the next step's
                                      // E(x) and gradient are the ones already computed; their kernels return at once (bitwise the
                                      // same chain: every kernel is deterministic).  f16x2 energy kernels.
    int ncu = 256;          // compute units of the device (residency-based kernel selection)
    ccsp_energy_hook energy_hook = nullptr;   // MALA global-batch mode (ccsp_model_set_energy_hook)
    void* energy_hook_ctx = nullptr;
    void* rccl_comm = nullptr;                // ccsp_model_set_energy_allreduce: the pair is all-reduced by ncclAllReduce on the chain's stream
    int row_mode = -1, edge_mt = -1;  // CCSP_ROW_MODE / CCSP_EDGE_MT: force a variant of the f16x2 kernels (-1: by tile count)
    int edge_small = -1;              // CCSP_EDGE_SMALL=1 / 0: always / never the 16-edge-tile kernel k_edge_h2s (-1: by tile count)
    // StructDiffusion baseline (model_kind 1): transformer weights as given ([out, in] row-major)
    struct SdLayer {
        float *in_w, *in_b, *out_w, *out_b, *ln1_g, *ln1_b, *fc_w, *fc_b, *proj_w, *proj_b, *ln2_g, *ln2_b;
        unsigned short *in_wH = nullptr, *out_wH = nullptr, *fc_wH = nullptr, *proj_wH = nullptr;    // fp16 planes [N][K / 32][2][32] * 2^exp (k_sd_gemm_h2)
        int in_e = 0, out_e = 0, fc_e = 0, proj_e = 0;
    };
    int sd_h2 = 0;         // 1: the transformer's GEMMs on the f16 pipe (f16x2; Wd a multiple of 128, CCSP_MMA unset or f16x2)
    int Wd = 0;            // transformer width: 2H, or 3H with a grasp group
    float *lnpre_g = nullptr, *lnpre_b = nullptr, *lnpost_g = nullptr, *lnpost_b = nullptr;
    float* sd_pe = nullptr;   // [8][Wd] positional-encoding rows (transformer.py:22-28)
    SdLayer sd[4];
    float *tm1_w = nullptr, *tm1_b = nullptr, *tm3_w = nullptr, *tm3_b = nullptr;   // time_mlp.{1,3} copies (operator API: float t)
    float* Wt = nullptr;    // [C][2H][H]  time slices of the type MLPs, and their biases bt [C][2H] (operator API)
    float* bt = nullptr;
    float* temb;   // [T][H]
    float* tau;    // [T][C][2H]      W_t . temb(t) + b_i
    std::vector<float> betas, ac, acp, sqrt_recip_ac, sqrt_recipm1_ac, post_lv, post_var, coef1, coef2, kappa, step;
    std::vector<float> sqrt_ac, sqrt_1m_ac, log_1m_ac;      // q_sample buffers (ddpm.py:210-212): checkpoint round trips only
    std::vector<int32_t> sps;
    std::vector<void*> allocs;
    // every live graph handle built on this model (children of lane splits included): ccsp_model_destroy
    // orphans them, so a graph destroyed after its model never touches the freed model or its streams
    std::vector<ccsp_graph*> graphs;
};

struct ccsp_graph {
    ccsp_model* m;
    int N, E, F;
    ccsp::Plan plan;
    int n_tiles;
    // device
    float* xfeat;
    signed char* mask;
    int *e_type, *e_u0, *e_u1, *e_orig, *urow_node, *tile_row0, *tile_nrows, *tile_ts, *node_ptr, *node_ent, *ent_pos;
    float *base, *U, *O, *pemb, *x, *eps;
    unsigned short* pembS = nullptr;   // [3][N][H] bf16 planes of pemb (bf16x3 mode)
    unsigned short* pembH = nullptr;   // [2][N][H] fp16 planes of pemb rows scaled by 2^pexp[n] (f16x2 mode)
    int* pexp = nullptr;               // [N]
    float* umax = nullptr;             // [R][8] max |U| per row and 64-column piece (k_rowgemm_h2 / _h3 -> k_edge_h2)
    int *t2_row0 = nullptr, *t2_nrows = nullptr, *t2_ts = nullptr;   // 128-row tiles of k_rowgemm_bf2 (pairs of plan tiles)
    int n_tiles2 = 0;
    // node update folded into the edge kernel's tail (FuseArgs): lists for edge tiles of fuse_me edges, arrival counters
    int *fuse_ptr = nullptr, *fuse_list = nullptr, *fuse_expect = nullptr;
    int *fuse_u0 = nullptr, *fuse_u1 = nullptr, *fuse_pos = nullptr;      // e_u0 / e_u1 / ent_pos in the fused kernel's edge order
    unsigned int* fuse_count = nullptr;
    int fuse_me = 0, fuse_blocks = 0;
    // node-grouped edge tiles (CCSP_FUSE_NODE=2, fuse2_prepare): -1 = not possible for this graph (a node with more than 64 entries)
    int ng_wgs = 0;
    bool ng_use = false;                      // this chain runs them
    int4* ng_desc = nullptr;
    int *ng_off0 = nullptr, *ng_off1 = nullptr;
    std::vector<int> h_ng;                    // kept alive for the async upload
    unsigned int fuse_epoch = 0;
    std::vector<int> h_fuse;                  // kept alive for the async upload
    // fused tiles of k_eval_fused (ccsp::FusedPlan)
    int4* ft_tiles = nullptr;
    int* ft_rows = nullptr;
    unsigned short* ft_elu = nullptr;
    int* ft_order = nullptr;                  // work list of the persistent launch: 2 tile + half, most expensive first
    std::vector<int> h_forder;
    int n_ftiles = 0;
    ccsp::FusedPlan fplan;                    // kept alive for the async upload
    int *tr64 = nullptr, *tr128 = nullptr;    // urow_node per tile row, padded per tile (StepRef::tile_rows)
    int4 *td64 = nullptr, *td128 = nullptr;   // the same tile lists as {row0, nrows, 2 type + slot, 0} records (k_rowgemm_h2: one scalar load per tile)
    std::vector<int> h_tr;                    // (kept alive for the asynchronous upload, like h_td)
    std::vector<int4> h_td;                   // kept alive for the async upload
    int* urow_ts;
    // energy mode (allocated on first use)
    bool energy_ready = false;
    int *e_a = nullptr, *e_b = nullptr, *row_ptr = nullptr, *row_edge = nullptr, *nrow_ptr = nullptr, *nrow_idx = nullptr;
    int *tileb_row0 = nullptr, *tileb_nrows = nullptr, *tileb_ts = nullptr;
    unsigned short* GZRS = nullptr;    // [3][R][2H] bf16 planes of GZR (energy backward on the bf16 pipe)
    unsigned short* GZRH = nullptr;    // [2][R][2H] fp16 planes of GZR rows scaled by 2^gexp[r] (energy backward on the f16 pipe)
    int* gexp = nullptr;               // [R]
    // row sums inside the decoder backward (ccsp::BwdSumPlan): partial rows instead of U rows downstream of it
    ccsp::BwdSumPlan bsplan;           // kept alive for the async upload
    bool bs_ready = false;
    int bs_fb = 0;                     // the decoder form the partial-row plan was built for (ccsp_model::edge_fb at energy_prepare)
    int *bs_blocks = nullptr, *bs_nrow_ptr = nullptr, *bs_nrow_idx = nullptr, *bs_gexp = nullptr;
    unsigned short* GZPH = nullptr;    // [2][NP][2H] fp16 planes of the partial rows scaled by 2^bs_gexp
    float* GPP = nullptr;              // [NP][H]
    int4 *bs_td64 = nullptr, *bs_td128 = nullptr;
    std::vector<int4> h_bstd;
    int bs_tiles = 0, bs_tiles2 = 0;
    float *Q = nullptr, *GZ = nullptr, *GZR = nullptr, *GP = nullptr, *xhat = nullptr, *partial = nullptr, *Escal = nullptr;
    int *acc_count = nullptr, *acc_denom = nullptr;
    int* mala_changed = nullptr;       // MALA reuse: nodes accepted by the last accept step
    float* zbuf = nullptr;             // [N, P] normal draws of the evaluation in flight (NoiseAhead)
    unsigned int* relay_ctr = nullptr; // relay mode: {row GEMM, edge, node} workgroups done since the chain began, fault flag
    hipEvent_t relay_ev[3] = {nullptr, nullptr, nullptr};
    int64_t relay_chains = 0;          // chains of this graph that ran in relay mode (ccsp_graph_variant)
    float* margin_buf = nullptr;       // ccsp_chain_margins: caller-owned [accept steps of a call][N] buffer, or null
    int64_t margin_cap = 0;            // its size in floats
    float *hmc_vk = nullptr, *hmc_vp = nullptr, *hmc_vl = nullptr;   // HMC momenta (allocated on first use)
    std::vector<int> h_denom;      // host copy kept alive for the async upload
    std::vector<int> h_t2;         // (row0 | nrows | ts) of the 128-row tiles, kept alive for the async upload
    int n_edge_blocks = 0;
    int n_part_last = 0;           // energy partials written by the most recent edge kernel
    std::vector<void*> allocs;
    // concurrent lanes: the batch cut into independent sub-batches (children), each a complete graph
    // object with its own stream, whose chains are enqueued interleaved (see ccsp_chain_run)
    std::vector<int64_t> h_ei;     // host copy of edge_index [2,E]
    std::vector<float> h_ea;       // host copy of edge_attr [E]
    std::vector<ccsp_graph*> children;
    std::vector<int> child_node0;
    int lanes_tried = 0;
    // StructDiffusion: token layout (ccsp_graph_set_sequences) and activations
    bool seq_ready = false;
    int sd_B = 0, sd_M = 0;
    std::vector<int> h_seq_graph, h_seq_pos, h_seq_cnt;   // host copies for the lanes: graph of node n, its position, nodes per graph (whole batch)
    int *tok_node = nullptr, *tok_pos = nullptr, *node_tok = nullptr, *mask_from = nullptr;
    float *gemb = nullptr, *remb = nullptr;
    float *sdX = nullptr, *sdY = nullptr, *sdQKV = nullptr, *sdA = nullptr, *sdF = nullptr;
    unsigned int* sdMax = nullptr;     // [4][M] bits of the row maxima of sdY (ln_1 output), sdA, sdX (after out_proj), sdF: the f16x2 GEMMs' row exponents
    // hipGraph mode (small batches): step table, header, counter and the instantiated per-S graphs
    StepEntry* d_tab = nullptr;
    ChainHeader* d_hdr = nullptr;
    int* d_counter = nullptr;
    size_t tab_cap = 0;
    std::vector<StepEntry> h_tab;
    ChainHeader h_hdr;
    std::map<int, hipGraphExec_t> execs;       // inner steps S -> graph of (1 + S) evaluations
    // profiling
    int profile = 0;
    int64_t evals = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool have_events = false;
    // profiling (ccsp_profile_enable): one event before every launch of the evaluation / update kernels, tagged with the
    // kernel about to run (CCSP_K_*), and one closing mark (-1) per evaluation; a kernel's duration is the elapsed time
    // to the next mark on the same stream (it includes the gap to the next launch)
    std::vector<hipEvent_t> kev;
    std::vector<int> kev_id;
    size_t kev_used = 0;
    int lanes_last = 0;                // concurrent lanes of the last ccsp_chain_run on this graph (ccsp_chain_lanes)
};

