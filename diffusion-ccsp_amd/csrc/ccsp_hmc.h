// AnnealedMUHASampler.sample_step (reference networks/ddpm.py:1050-1128) + leapfrog_step (:917-937):
// the elementwise parts of one HMC timestep on the pose rows [N,P].  The network evaluations between
// them are the energy-mode launches of ccsp_hip.hip.  Included by ccsp_hip.hip.
#pragma once

enum { HMC_MOMENTUM = 0, HMC_REFRESH = 1, HMC_LEAP_A = 2, HMC_LEAP_B = 3, HMC_ACCEPT = 4 };

struct HmcArgs {
    int N, P, F, mode;
    float* x;             // chain state x_k
    float* xl;            // leapfrog position (x_k_next)
    float *vk, *vp, *vl;  // momentum v_k, refreshed v_k', leapfrog momentum
    const float* eps;     // model output at xl and inner index i: gradient_function = -eps * kappa_i
    float m_t;            // mass_diag_sqrt[t] = 9 betas[t]: momentum scale and v_dist scale (real timestep)
    float ss_i, md_i, kap_i;   // step size, mass_diag = (9 betas[i])^2 and kappa of the INNER index i (ddpm.py:1076-1084)
    float kappa_t;
    const float* E_x;     // batch energies at x_k and x_k_next, timestep t (device scalars)
    const float* E_hat;
    int* acc_count;
    int reset_mask;       // last inner step: x[mask] = gt[mask], history slot
    const signed char* mask;
    const float* xfeat;
    int pose_begin;
    float* hist;
    float* margin;        // HMC_ACCEPT, debugging aid (ccsp_chain_margins): [N] log acceptance ratio - log u, or null
    NoiseArg noise;
};

__global__ __launch_bounds__(256) void k_hmc(HmcArgs a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)a.N * a.P) return;
    const int n = (int)(idx / a.P), p = (int)(idx % a.P);
    const size_t i = (size_t)idx;
    if (a.mode == HMC_MOMENTUM || a.mode == HMC_REFRESH) {
        float z;
        if (a.noise.mode == CCSP_NOISE_INJECTED) z = a.noise.normal[i];
        else z = ccsp::philox_normal(a.noise.seed, a.noise.row_offset + (unsigned long long)n, a.noise.call, p);
        if (a.mode == HMC_MOMENTUM) { a.vk[i] = z * a.m_t; return; }            // ddpm.py:1090
        const float v = a.vk[i] * 0.0f + (1.0f * z) * a.m_t;                     // damping 0 (:1099): a non-finite v_k stays non-finite
        a.vp[i] = v; a.vl[i] = v; a.xl[i] = a.x[i];
        return;
    }
    if (a.mode == HMC_LEAP_A || a.mode == HMC_LEAP_B) {                          // ddpm.py:930-934
        const float half = 0.5f * a.ss_i;
        const float v = a.vl[i] + half * ((-a.eps[i]) * a.kap_i);
        a.vl[i] = v;
        if (a.mode == HMC_LEAP_A) a.xl[i] = a.xl[i] + a.ss_i * v / a.md_i;
        return;
    }
    // HMC_ACCEPT (ddpm.py:1104-1121): one decision per node row from batch-scalar energies
    const size_t r0 = (size_t)n * a.P;
    const float var = a.m_t * a.m_t, log_scale = logf(a.m_t), lc = 0.918938533204672742f;
    float lvp = 0.0f, lv = 0.0f;
    for (int c = 0; c < a.P; ++c) {
        const float vp = a.vp[r0 + c], vn = a.vl[r0 + c];
        lvp += -(vp * vp) / (2.0f * var) - log_scale - lc;
        lv += -(vn * vn) / (2.0f * var) - log_scale - lc;
    }
    const float logp_x = (-a.E_x[0]) * a.kappa_t, logp_h = (-a.E_hat[0]) * a.kappa_t;
    const float la = (logp_h + lv) - (logp_x + lvp);
    float u;
    if (a.noise.mode == CCSP_NOISE_INJECTED) u = a.noise.uniform[n];
    else u = ccsp::philox_uniform(a.noise.seed, a.noise.row_offset + (unsigned long long)n, a.noise.ucall);
    const float acc = (u < expf(la)) ? 1.0f : 0.0f;
    if (p == 0 && acc != 0.0f && a.acc_count) atomicAdd(a.acc_count, 1);
    if (p == 0 && a.margin) {
        a.margin[n] = la - logf(u);
        a.margin[a.N + n] = fabsf(logp_h) + fabsf(lv) + fabsf(logp_x) + fabsf(lvp);
    }
    float xv = acc * a.xl[i] + (1.0f - acc) * a.x[i];
    const float vv = acc * a.vl[i] + (1.0f - acc) * a.vp[i];
    if (a.reset_mask && a.mask[n]) xv = a.xfeat[(size_t)n * a.F + a.pose_begin + p];
    // every thread of the row has read x / xl / vl / vp of the WHOLE row above only through vp/vl (momenta);
    // x and vk are written element-wise, and vp / vl are not modified here, so no intra-row hazard
    a.x[i] = xv;
    a.vk[i] = vv;
    if (a.hist) a.hist[i] = xv;
}
