/* ccsp.h -- C ABI of the MI355X-native Diffusion-CCSP sampling path (libccsp_hip.so).
 *
 * The reference (zt-yang/diffusion-ccsp) has no FFI; its seam for this path is duck-typed Python
 * (SURVEY.md 8b).  Each entry point below names the reference interface it replaces
 * (file:line in the reference repository).  INTEGRATION.md shows the ctypes binding a reference
 * maintainer would add; diffusion-ccsp_amd/_lib.py is that binding.
 *
 * Conventions
 *  - every array pointer is a DEVICE pointer owned by the caller (e.g. torch tensor.data_ptr()),
 *    fp32 / int64 / int8 row-major contiguous, unless the parameter is documented "host";
 *  - every call enqueues its work on the caller-supplied hipStream_t (`stream`, passed as
 *    void*; NULL = the default stream).  ccsp_model_create / ccsp_graph_create /
 *    ccsp_graph_set_sequences (one-time set-up: they read small arrays back to build index tables),
 *    ccsp_chain_stats and the ccsp_*_get calls synchronise.  ccsp_denoise / ccsp_energy_grad /
 *    ccsp_edge_outputs never do.  ccsp_chain_run does not wait for the chain it enqueues; it waits
 *    for EARLIER work on the stream only where a host table of a previous chain on the same graph is
 *    about to be rewritten (energy-mode samplers);
 *  - streams and threads the library owns: a direct-mode ccsp_chain_run on a batch of >= 6144 active edges cuts it into two
 *    independent sub-batches ("lanes") that run on two streams created once per process and device (shared by every model), forked from and joined to the caller's
 *    stream by events; the two lanes are enqueued by two std::thread workers that live for the duration of the call (a
 *    chain is 33 000 launches per lane: one thread alternating between streams is launch-bound).  The call returns when both
 *    have enqueued everything, i.e. it blocks for the enqueue time but not for the chain.  Host cost measured on an MI355X
 *    box (profiles/r03_findings.md): pinned to two cores (`taskset -c 0-1`) the C2 chain runs at the unpinned rate; with one
 *    lane (CCSP_LANES=1: no threads) pinned to ONE core likewise.  Round 6 measured what a rank costs its host: 2.67 busy cores with two
 *    lanes, 1.9 with one (profiles/r06_host_budget.txt; bench.py budgets that for N ranks).  Energy mode and profiled runs use the
 *    caller's stream only (MALA can run as two coupled lanes, CCSP_MALA_LANES=2: measured slower, off by default); the transformer
 *    baseline is cut into lanes from 1024 token rows on.  The lane streams are shared by every model of the process (one pool per device): chains of DIFFERENT
 *    models -- or of one model enqueued from different caller streams -- serialise per lane on them, and their fork / join events couple the
 *    caller streams involved (each waits for the other's lane work in front of its own).  ccsp_model_destroy waits for the destroyed model's
 *    own last chains (its join events), not for other models' work on the shared streams;
 *  - return value 0 = ok, non-zero = error; ccsp_last_error() gives the thread-local message;
 *  - no exceptions cross the boundary; handles are not thread-safe (one per device & stream);
 *  - NaN is data, not an error (isolated nodes give 0/0 exactly like the reference,
 *    networks/denoise_fn.py:523-524).
 */
#ifndef CCSP_H
#define CCSP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ccsp_version() = 1000 MAJOR + MINOR.  MAJOR changes whenever the signature or the meaning of an existing entry point changes (a binding
 * built against another MAJOR must refuse the library: diffusion-ccsp_amd/_lib.py does, INTEGRATION.md section 2 shows the check); MINOR
 * counts additions.  1.0 = round 4's 0.7 with ccsp_compose_chain_run's `accept` argument (added in round 4 WITHOUT a bump: the reason for
 * the rule) + ccsp_rccl_comm_count / ccsp_rccl_allreduce_sum_f32 + ccsp_chain_margins; ccsp_plan_fused_host moved behind CCSP_EXPERIMENTS.
 * 1.1 (round 6) adds ccsp_chain_lanes, ccsp_plan_bwdsum_blocks_host and the kernel selector CCSP_K_EDGE_FB. */
#define CCSP_VERSION_MAJOR 1
#define CCSP_VERSION_MINOR 1
#define CCSP_MAX_SAMPLES_PER_STEP 100000

typedef struct ccsp_model ccsp_model;   /* ConstraintDiffuser weights + GaussianDiffusion schedule */
typedef struct ccsp_graph ccsp_graph;   /* one collated batch of constraint graphs              */

/* Static description of a ConstraintDiffuser (networks/denoise_fn.py:184-291) wrapped in a
 * GaussianDiffusion (networks/ddpm.py:168-228).  `dims` is the reference's tuple of
 * (length, begin, end) per variable group (train_utils.py:266-278). */
typedef struct {
    int32_t hidden_dim;     /* H: -hidden_dim (train_utils.py:107); supported: multiples of 64 up to 512 */
    int32_t pose_dim;       /* P = dims[-1][0] (4 or 5)                                          */
    int32_t pose_begin;     /* dims[-1][1]: ground-truth pose columns x[:, pose_begin:+P]        */
    int32_t geom_dim;       /* dims[0][0]: geometry columns x[:, 0:geom_dim]                     */
    int32_t grasp_dim;      /* 0, or dims[1][0] when 'robot' in input_mode                       */
    int32_t grasp_begin;    /* dims[1][1]                                                        */
    int32_t n_types;        /* len(constraint_sets) (denoise_fn.py:207-214)                      */
    int32_t timesteps;      /* -timesteps                                                        */
    int32_t normalize;      /* -normalize (denoise_fn.py:523)                                    */
    int32_t energy_wrapper; /* 1 when wrapped in ComposedEBMDenoiseFn (denoise_fn.py:57-83)      */
    int32_t ebm_per_steps;  /* denoise_fn.ebm_per_steps (denoise_fn.py:284; ddpm.py:330)         */
    int32_t model_kind;     /* CCSP_MODEL_*: ConstraintDiffuser(model=...) (denoise_fn.py:205,267-282) */
} ccsp_model_desc;

/* 'Diffusion-CCSP': per-constraint-type MLPs over the edges.  'StructDiffusion': the transformer
 * baseline over each graph's object sequence (denoise_fn.py:391-451, transformer.py). */
enum { CCSP_MODEL_DIFFUSION_CCSP = 0, CCSP_MODEL_STRUCT_DIFFUSION = 1 };

enum { CCSP_SAMPLER_NONE = 0, CCSP_SAMPLER_ULA = 1, CCSP_SAMPLER_ULA_PLUS = 2, CCSP_SAMPLER_MALA = 3,
       CCSP_SAMPLER_HMC = 4 /* AnnealedMUHASampler, ddpm.py:1050-1128: S = 4, 2 leapfrogs, mass 9 betas */ };
enum { CCSP_NOISE_PHILOX = 0, CCSP_NOISE_INJECTED = 1 };

/* Where the chain's torch.randn / torch.rand draws come from (ddpm.py:255,273,292,1037).
 * PHILOX: the stateless counter-based stream of diffusion-ccsp_amd/noise.py, regenerated on
 * device.  INJECTED: the caller supplies the draws (parity runs against a recorded stream). */
typedef struct {
    int32_t mode;
    int32_t _pad;
    uint64_t seed;          /* PHILOX                                                            */
    uint64_t row_offset;    /* PHILOX: global row index of this batch's row 0 (sharded batches)  */
    const float* normal;    /* INJECTED: device [n_normal, N, P]; entry k = randn call call_base+k */
    uint64_t n_normal;
    const float* uniform;   /* INJECTED (MALA): device [n_uniform, N]                            */
    uint64_t n_uniform;
    uint64_t call_base;
    uint64_t ucall_base;
} ccsp_noise;

const char* ccsp_last_error(void);
int32_t ccsp_version(void);                       /* major * 1000 + minor                        */
/* name, compute units, HBM bytes of the current HIP device; 0 on success                       */
int ccsp_device_info(char* name, int32_t name_len, int32_t* compute_units, uint64_t* hbm_bytes);

/* Replaces ConstraintDiffuser.__init__ + load_state_dict (denoise_fn.py:184-308,
 * ddpm.py:503-514).  params: host array of 2*n_linear DEVICE pointers, weight then bias, in the
 * reference state_dict order: geom_encoder.{0,2}, [grasp_encoder.{0,2}], pose_encoder.{0,2},
 * pose_decoder.{0,2}, time_mlp.{1,3}, mlps.i.0 (i < n_types).  For CCSP_MODEL_STRUCT_DIFFUSION the
 * mlps are replaced by (weight, bias) pairs of: ln_pre; per block l < 4: attn.in_proj, attn.out_proj,
 * ln_1, mlp.c_fc, mlp.c_proj, ln_2; ln_post (transformer.py:43-57).  Weights are copied (and re-laid
 * out) into library-owned HBM; the cosine schedule (ddpm.py:152-162) and the per-type time-term
 * table are built here. */
int ccsp_model_create(const ccsp_model_desc* desc, const float* const* params, void* stream,
                      ccsp_model** out);
/* Graph handles built on the model stay valid to DESTROY afterwards (any other call on them fails). */
void ccsp_model_destroy(ccsp_model* model);

/* Replaces GaussianDiffusion.__init__'s betas / step_sizes / samples_per_step arguments
 * (ddpm.py:169-228).  n: length of every non-NULL array, must equal the model's timesteps (the
 * reference takes T from betas.shape, ddpm.py:189).  betas: HOST double[n] in [0, 1) or NULL
 * (cosine); step_sizes: HOST float[n] or NULL ('2*self.betas'); samples_per_step: HOST int32[n]
 * in [0, CCSP_MAX_SAMPLES_PER_STEP] or NULL (default_samples for every t). */
int ccsp_schedule_set(ccsp_model* model, int32_t n, const double* betas, const float* step_sizes,
                      const int32_t* samples_per_step, int32_t default_samples);
/* Copies one schedule buffer to HOST float[T].  which: 0 betas, 1 alphas_cumprod,
 * 2 alphas_cumprod_prev, 3 sqrt_recip_alphas_cumprod, 4 sqrt_recipm1_alphas_cumprod,
 * 5 posterior_log_variance_clipped, 6 posterior_mean_coef1, 7 posterior_mean_coef2,
 * 8 _sqrt_recipm1_alphas_cumprod_custom, 9 step_sizes, 10 posterior_variance,
 * 11 sqrt_alphas_cumprod, 12 sqrt_one_minus_alphas_cumprod, 13 log_one_minus_alphas_cumprod
 * (ddpm.py:200-228: the twelve registered buffers of a checkpoint, plus 8 and 9). */
int ccsp_schedule_get(const ccsp_model* model, int32_t which, float* out_host);
/* time_mlp(t) (denoise_fn.py:259-264) for one t -> DEVICE float[H] (visualize_energy.py:402-450
 * reads denoise_fn.time_mlp) */
int ccsp_time_embedding(ccsp_model* model, int32_t t, float* out, void* stream);

/* Operator-level entry points: visualize_energy.py:402-450 calls the denoiser's sub-modules on tensors of its own
 * (a grid of poses for one constraint type).  All arrays DEVICE fp32 row-major; not on the sampling path.
 *   ccsp_encode              denoise_fn.geom_encoder / .pose_encoder / .grasp_encoder (denoise_fn.py:227-250):
 *                            in [n, dims[group][0]] -> out [n, H]
 *   ccsp_time_mlp            denoise_fn.time_mlp on arbitrary (float) timestep values (denoise_fn.py:38-50,259-264):
 *                            t_values [n] -> out [n, H]
 *   ccsp_process_constraint  ConstraintDiffuser._process_constraint(type, input_dict) (denoise_fn.py:341-371):
 *                            geoms_emb [n, 2, H], poses_emb [n, 2, H], time_emb [n, H], grasp_emb [n, H] ('robot'
 *                            models only, else NULL) -> out [n, 2, P], the decoded pose pair of every row */
enum { CCSP_ENC_GEOM = 0, CCSP_ENC_POSE = 1, CCSP_ENC_GRASP = 2 };
int ccsp_encode(ccsp_model* model, int32_t which, int32_t n, const float* in, float* out, void* stream);
int ccsp_time_mlp(ccsp_model* model, int32_t n, const float* t_values, float* out, void* stream);
int ccsp_process_constraint(ccsp_model* model, int32_t type, int32_t n, const float* geoms_emb, const float* poses_emb,
                            const float* time_emb, const float* grasp_emb, float* out, void* stream);

/* Replaces the per-evaluation graph handling of ConstraintDiffuser.forward
 * (denoise_fn.py:313-339,466-485,508): one-time type sort of the edges, node->edge CSR in the
 * reference's accumulation order, geometry/grasp embeddings and the chain-constant part of every
 * edge's pre-activation.  x [N,F] fp32, edge_index [2,E] int64, edge_attr [E] fp32 (integer
 * ids; edges matching no type are ignored like the reference does), mask [N] int8. */
int ccsp_graph_create(ccsp_model* model, int32_t N, int32_t E, int32_t F, const float* x,
                      const int64_t* edge_index, const float* edge_attr, const int8_t* mask,
                      void* stream, ccsp_graph** out);
void ccsp_graph_destroy(ccsp_graph* graph);
/* StructDiffusion only (denoise_fn.py:408-423): batch [N] int64 DEVICE = graph id of every node
 * (PyG batch.batch; a graph's nodes in order are its tokens, at most 8), shuffled [N] int64 DEVICE or
 * NULL = batch.shuffled, the positional-encoding row of every token.  Once per graph handle. */
int ccsp_graph_set_sequences(ccsp_graph* graph, const int64_t* batch, const int64_t* shuffled,
                             void* stream);

/* ConstraintDiffuser.forward(poses_in, batch, t, eval=True), direct mode
 * (denoise_fn.py:453-537): poses_in [N,P] -> out [N,P]. */
int ccsp_denoise(ccsp_model* model, ccsp_graph* graph, const float* poses_in, int32_t t,
                 float* out, void* stream);
/* Energy mode (denoise_fn.py:373-375,518-519,527-529,539-548; ComposedEBMDenoiseFn.forward /
 * .neg_logp_unnorm :70-83): grad [N,P] = dE/dposes_in, energy [1] = E summed over the batch. */
int ccsp_energy_grad(ccsp_model* model, ccsp_graph* graph, const float* poses_in, int32_t t,
                     float* grad, float* energy, void* stream);
/* _process_constraint outputs (denoise_fn.py:341-371; visualize_energy.py:450): [E,2,P] in the
 * caller's edge order, NaN rows for ignored edges. */
int ccsp_edge_outputs(ccsp_model* model, ccsp_graph* graph, const float* poses_in, int32_t t,
                      float* out, void* stream);

/* GaussianDiffusion.p_sample_loop (ddpm.py:260-340) with p_sample (:245-258) and
 * AnnealedULASampler / AnnealedMALASampler.sample_step (:940-966, :999-1047): runs timesteps
 * t_first, t_first-1, ..., t_last.  init != 0: draw the initial state first (randn call 0,
 * ddpm.py:273-274) -- a whole chain is (init=1, t_first=T-1, t_last=0); init == 0: x [N,P]
 * holds the state on entry.  x receives the final state.  history: NULL or [(T+1),N,P]; entry k
 * = state after k timesteps.  accept: NULL or [T] mean MALA acceptance per timestep.
 * Asynchronous: the caller synchronises the stream before reading x. */
int ccsp_chain_run(ccsp_model* model, ccsp_graph* graph, int32_t sampler, const ccsp_noise* noise,
                   float* x, int32_t init, int32_t t_first, int32_t t_last, float* history,
                   float* accept, void* stream);

/* Composition of two constraint domains on one set of nodes -- the reference's input_mode 'robot_qualitative'
 * (networks/denoise_fn.py:287-291 pose_encoder_2 / geom_encoder_2 / pose_decoder_2 / time_mlp_2 and composing_weight,
 * :310-311 the types >= n_types(first) use the second set, :341-371 their decoded outputs get a zero column and both
 * domains their weight, :487-503 the second domain's inputs).  `first` / `graph_first` hold the first domain (all P pose
 * columns, the edges of its types); `second` / `graph_second` the second domain: pose_dim P - 1, the SAME N nodes, the
 * edges of the second domain with their types renumbered from 0, and node features whose pose columns are placeholders
 * (the library fills poses_2 = [poses[:, :2] | x[:, -(P-3):]] itself, x = graph_first's features, denoise_fn.py:499).
 * One composed evaluation = one ordinary evaluation per domain + one elementwise kernel:
 *     out = (w1 * sum1 + w2 * widen(sum2)) / sqrt(count1 + count2);  out[mask] = x[:, -P:][mask]
 * Both models must be Diffusion-CCSP models with the same `timesteps`, direct-mode (energy_wrapper 0) for ccsp_compose_denoise
 * and ccsp_compose_chain_run; their own `normalize` flags are not used.  The chain form runs samplers NONE / ULA / ULA+ with the
 * schedule of `first`; when both are energy_wrapper models (every evaluation = the composed energy gradient) also MALA and HMC
 * (ddpm.py:999-1047 / :1050-1128 on gradient_function / energy_function of the composed model, :280-289; `accept` [timesteps] = mean
 * acceptance per timestep, or NULL). */
typedef struct ccsp_compose {
    int32_t zero_col;       /* column of the P-wide pose the second domain does not produce (2: z) */
    float weight_first;     /* composing_weight[0] */
    float weight_second;    /* composing_weight[1] */
    int32_t normalize;      /* ConstraintDiffuser.normalize */
} ccsp_compose;
int ccsp_compose_denoise(ccsp_model* first, ccsp_graph* graph_first, ccsp_model* second, ccsp_graph* graph_second,
                         const ccsp_compose* compose, const float* poses_in, int32_t t, float* out, void* stream);
/* energy mode of the composed model (both models energy_wrapper = 1; composing weights (1, 1)): energy [1] = sum over the
 * edges of both domains of |outputs - poses_in[args]|^2 with the second domain's outputs widened by the zero column
 * (denoise_fn.py:373-375 on :364-370), grad [N,P] = d energy / d poses_in (the second domain's encoder sees poses_in[:, :2]
 * only, denoise_fn.py:499) */
int ccsp_compose_energy_grad(ccsp_model* first, ccsp_graph* graph_first, ccsp_model* second, ccsp_graph* graph_second,
                             const ccsp_compose* compose, const float* poses_in, int32_t t, float* grad, float* energy,
                             void* stream);
int ccsp_compose_chain_run(ccsp_model* first, ccsp_graph* graph_first, ccsp_model* second, ccsp_graph* graph_second,
                           const ccsp_compose* compose, int32_t sampler, const ccsp_noise* noise, float* x, int32_t init,
                           int32_t t_first, int32_t t_last, float* history, float* accept, void* stream);

/* MALA across shards.  The reference's accept test uses ONE energy for the whole batch
 * (logp_x, logp_x_hat of shape [1], ddpm.py:1026-1038), so a batch cut into per-GPU shards only reproduces the
 * unsharded chain if the shards' energies are summed.  With a hook installed, every MALA inner step calls
 * hook(ctx, pair, stream) between the energy evaluation at the proposal and the accept step: pair = DEVICE float[2]
 * {E(x), E(x_hat)} of this shard; the hook replaces both by their sums over all shards with work enqueued on `stream`
 * (e.g. an RCCL all_reduce) and returns 0.  NULL removes the hook (replica semantics: each shard is its own batch). */
typedef int (*ccsp_energy_hook)(void* ctx, float* pair, void* stream);
int ccsp_model_set_energy_hook(ccsp_model* model, ccsp_energy_hook hook, void* ctx);
/* The same reduction inside the library: with a communicator installed, every MALA inner step enqueues
 * ncclAllReduce(pair, pair, 2, ncclFloat32, ncclSum, comm, stream) on the chain's own stream between the energy evaluation at the
 * proposal and the accept step -- no callback, no host round trip (the Python trampoline of round 3 cost 10 % of a C4 chain with
 * ONE rank).  RCCL is bound at run time (CCSP_RCCL_LIB, else the already-loaded "librccl.so.1" by RTLD_NOLOAD, else dlopen): a process that already carries an RCCL
 * (PyTorch-ROCm) gets that instance.  comm = an ncclComm_t; NULL removes it; a communicator takes precedence over a hook.
 * ccsp_rccl_unique_id / ccsp_rccl_comm_create / ccsp_rccl_comm_destroy wrap ncclGetUniqueId / ncclCommInitRank (on the calling
 * thread's current device; collective over the n_ranks callers) / ncclCommDestroy for hosts that have no communicator of their
 * own: rank 0 draws the 128-byte id, sends it to the other ranks by any means, every rank creates its communicator. */
#define CCSP_RCCL_ID_BYTES 128
int ccsp_model_set_energy_allreduce(ccsp_model* model, void* comm);
int ccsp_rccl_unique_id(void* id /* host, CCSP_RCCL_ID_BYTES */);
int ccsp_rccl_comm_create(int32_t n_ranks, int32_t rank, const void* id, void** comm);
int ccsp_rccl_comm_destroy(void* comm);
/* What the communicator itself says (ncclCommCount) and the RCCL it is bound to (ncclGetVersion's code; `version` may be NULL):
 * bench.py prints both in its N > 1 line as the proof that RCCL joined N ranks.  ccsp_rccl_allreduce_sum_f32 = ncclAllReduce(buf, buf, n,
 * ncclFloat32, ncclSum) on `stream` over that communicator (the same call the MALA reduction enqueues; buf = device floats). */
int ccsp_rccl_comm_count(void* comm, int32_t* n_ranks, int32_t* version);
int ccsp_rccl_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream);

/* Kernel-level timing of the most recent ccsp_chain_run on this graph, measured with HIP events
 * on the chain's stream (bench.py's roofline block).  evals = network evaluations executed,
 * ms_total = event time of the whole chain.  After ccsp_profile_enable(graph, 1) the launches of a chain
 * are additionally bracketed one by one (ccsp_kernel_stats): ms_ugemm / ms_edge = mean duration of the forward
 * row GEMM / edge kernel (0 when profiling is off).  Synchronises on the chain's end. */
int ccsp_profile_enable(ccsp_graph* graph, int32_t on);
int ccsp_chain_stats(ccsp_graph* graph, int64_t* evals, float* ms_total, float* ms_ugemm, float* ms_edge);

/* Debugging aid for the Metropolis samplers (AnnealedMALASampler / AnnealedMUHASampler accept tests, ddpm.py:1026-1041,1104-1121): with a
 * buffer installed, every accept step of the following ccsp_chain_run / ccsp_compose_chain_run calls on this graph writes, per node row,
 * margin = log(acceptance ratio) - log(u) -- the row is accepted iff margin > 0 -- to margins[k * 2 N + n] and scale = the sum of the absolute
 * values of the four terms the ratio is made of (the two batch log-probabilities, the two proposal / momentum log-densities) to
 * margins[k * 2 N + N + n], k = index of the accept step within the call in chain order (timesteps t_first..t_last, inner steps ascending).  An accept test is a discrete decision: two correct fp32
 * implementations decide a row with |margin| <~ 1e-6 scale either way, and through the batch-scalar energy every later decision follows.  The parity
 * tests use this to assert that every decision that differs from the reference's is such a near-tie.  DEVICE buffer owned by the caller,
 * n_floats its size (steps beyond it are not recorded); NULL removes it.  Costs one store per node row and accept step. */
int ccsp_chain_margins(ccsp_graph* graph, float* margins, int64_t n_floats);
/* MALA chains (ddpm.py:1012-1047; the gradient and E(x) of ddpm.py:1019,1026): an inner step that accepted NO node leaves x where it was, so E(x) and dE/dx of the next
 * inner step are the values already computed; unless CCSP_MALA_REUSE=0 the kernels of that gradient evaluation then return
 * at once (decided on the device from the accept kernel's count; every kernel is deterministic, so the chain is bitwise the
 * one that recomputes -- tests/test_hip_parity.py::test_mala_rejected_step_reuse_is_bitwise_identical).  `evals` of
 * ccsp_chain_stats counts the evaluations ENQUEUED; this returns how many of them the last chain skipped.  Synchronises. */
int ccsp_chain_skipped(ccsp_graph* graph, int64_t* evaluations_skipped);
/* How many concurrent lanes the last ccsp_chain_run on this graph was cut into (1 = the caller's stream only, no library threads; 2 = the
 * default above 6144 active edges in direct mode: two pooled streams and two enqueueing std::thread workers, i.e. two busy host cores for
 * the duration of the call -- what a launcher that places several ranks on one host has to budget, bench.py `host_budget`).  The reference
 * is one process and one stream (networks/ddpm.py:342-351); this is the port's own degree of freedom (CCSP_LANES).  Does not synchronise. */
int ccsp_chain_lanes(ccsp_graph* graph, int32_t* lanes);
/* Per-kernel timing of a profiled chain (ccsp_profile_enable): while profiling, an event is recorded before every
 * launch of the kernels below (the first CCSP_PROFILE_MARKS marks of a chain); which = CCSP_K_*; calls = launches
 * seen, ms_mean = their mean duration, launch to next mark on the stream; name = a short label (may be NULL). */
enum { CCSP_K_ROWGEMM = 0, CCSP_K_EDGE = 1, CCSP_K_NODE = 2, CCSP_K_EDGE_BWD = 3, CCSP_K_ROWSUM = 4, CCSP_K_ROWGEMM_T = 5,
       CCSP_K_NODE_ENERGY = 6, CCSP_K_ENERGY_SUM = 7, CCSP_K_HMC = 8, CCSP_K_SD_EVAL = 9, CCSP_K_EVAL_FUSED = 10, CCSP_K_EDGE_FB = 11 /* 1.1 */,
       CCSP_K_COUNT = 12 };
#define CCSP_PROFILE_MARKS 16384
int ccsp_kernel_stats(ccsp_graph* graph, int32_t which, int64_t* calls, float* ms_mean, char* name, int32_t name_len);
/* Which variants of the f16x2 evaluation kernels a ONE-lane launch on this graph runs (they are chosen by tile count, see
 * csrc/ccsp_f16x2.h): row_mode = MODE of k_rowgemm_h2 (0..4), edge_tile = edges per workgroup of the edge kernel (16: k_edge_h2s,
 * 32 / 64: k_edge_h2<., 1 | 2, .>); -1 each when the model does not run the f16x2 kernels.  For reports that pair a timing with
 * counters taken in another process (bench.py). */
int ccsp_graph_variant(ccsp_graph* graph, int32_t* row_mode, int32_t* edge_tile);

/* Host-only planning entry (needs no device): the one-time index tables ccsp_graph_create builds
 * from the edge lists -- type-sorted edges, the distinct (type, slot, node) rows, their row tiles
 * and the node->(edge,slot) CSR in the reference's scatter_add_ order (denoise_fn.py:377-389,
 * :512-521).  All pointers are HOST pointers sized by the caller: per-edge arrays [E],
 * urow_* [2E], tile_* [2E + 2C], node_ptr [N+1], node_ent [2E]; counts = {E_act, R, n_tiles}.
 * Output arrays may be NULL. */
int ccsp_plan_host(int32_t N, int32_t E, int32_t C, const int64_t* edge_index, const float* edge_attr,
                   int32_t* counts, int32_t* e_orig, int32_t* e_type, int32_t* e_u0, int32_t* e_u1,
                   int32_t* urow_node, int32_t* urow_ts, int32_t* tile_row0, int32_t* tile_nrows,
                   int32_t* tile_ts, int32_t* node_ptr, int32_t* node_ent);

#ifdef CCSP_EXPERIMENTS   /* exported by the experiments build only (libccsp_hip_exp.so; _lib.build(experiments=True)) */
/* Host-only: the FUSED TILES of the one-launch evaluation kernels (csrc/ccsp_fused.h; the same loop over constraint types and
 * their edges, denoise_fn.py:313-371).  Every type's sorted edges are cut, in order, into runs whose distinct U rows are at most
 * rows_per_slot (<= 32; the kernels use 28 or 32) per slot and whose length is at most max_edges (<= 128; 112 or 128); a workgroup
 * owns (tile, output half), computes the tile's U rows into LDS and decodes the tile's edges from there.  n_tiles <= E_act;
 * tiles [n_tiles][4] = {type, first sorted edge, edges, rows slot 0 | rows slot 1 << 16}; rows [n_tiles][128]: entries 0..63 =
 * node of A row i (slot 0 at 0.., slot 1 at 32..), entries 64..127 = U row of tile row j (slot 0 at 0.., slot 1 at
 * rows_per_slot..); unused entries repeat the slot's first row; e_lu [E_act] = tile row of operand 0 | tile row of operand 1 << 8.
 * Caller-sized HOST arrays (tiles [4 E], rows [128 E], e_lu [E]); any of them may be NULL. */
int ccsp_plan_fused_host(int32_t N, int32_t E, int32_t C, const int64_t* edge_index, const float* edge_attr, int32_t rows_per_slot,
                         int32_t max_edges, int32_t* n_tiles, int32_t* tiles, int32_t* rows, uint16_t* e_lu);
#endif

/* Host-only: the PARTIAL ROWS of the energy backward (csrc/ccsp_plan.h build_bwdsum_plan).  The reference's autograd adds, for every
 * constraint type, the gradient of an edge's MLP input back onto the two nodes' embeddings (the backward of the gathers at
 * denoise_fn.py:341-371); here the decoder-backward kernel works on blocks of 64 consecutive sorted edges and adds up, per block, the
 * gradients of the edges that share a U row -- one partial row per (block, U row), numbered by (U row, block) ascending.
 * blocks [n_blocks][513]: [0] = partial rows of the block, [1 + p] = global partial row, [129 + p] = first << 16 | end (exclusive) of
 * p's PAIRS of entries in the block's reference list, [257 + 2 q], [258 + 2 q] = pair q, each entry 528 x (block-local edge) -- the
 * byte offset of that edge's row in the kernel's LDS tile -- an odd count padded with 528 x 64 (an all-zero row); prow_urow
 * [n_partial] = U row of a partial row; nrow_ptr [N + 1] / nrow_idx [n_partial] = partial rows of every node, ascending.
 * Caller-sized HOST arrays (blocks [513 (E / 64 + 1)], prow_urow and nrow_idx [2 E], nrow_ptr [N + 1]); any of them may be NULL. */
int ccsp_plan_bwdsum_host(int32_t N, int32_t E, int32_t C, const int64_t* edge_index, const float* edge_attr, int32_t* n_blocks,
                          int32_t* n_partial, int32_t* blocks, int32_t* prow_urow, int32_t* nrow_ptr, int32_t* nrow_idx);
/* (1.1) The same for blocks of `block_edges` sorted edges with at most `max_parts` partial rows each (max_parts >= 2 block_edges): what the fused
 * decoder kernel of round 6 runs on is (32, 64) -- its tile is 32 edges x both halves.  blocks [n_blocks][1 + 4 max_parts]: [0] = partial rows,
 * [1 + p] = global partial row, [1 + max_parts + p] = first << 16 | end of p's pairs, [1 + 2 max_parts + 2 q], [.. + 1] = pair q (528 x block-local
 * edge; padding entry 528 x 64 in every geometry).  (64, 128) is ccsp_plan_bwdsum_host.  Caller-sized host arrays: blocks
 * [(1 + 4 max_parts) (E / block_edges + 1)], the rest as above. */
int ccsp_plan_bwdsum_blocks_host(int32_t N, int32_t E, int32_t C, const int64_t* edge_index, const float* edge_attr, int32_t block_edges,
                                 int32_t max_parts, int32_t* n_blocks, int32_t* n_partial, int32_t* blocks, int32_t* prow_urow, int32_t* nrow_ptr,
                                 int32_t* nrow_idx);

#ifdef __cplusplus
}
#endif
#endif
