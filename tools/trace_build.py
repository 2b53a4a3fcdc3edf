"""Builds tools/abl_trace.so: the shipped kernels with s_memtime stamps at their phase boundaries (k_rowgemm_h2 ring form,
k_edge_h2, k_node with the f16 encoder), for tools/trace_run.py.  The stamps are inserted by exact-text patches of a COPY of the
sources (nothing under diffusion-ccsp_amd/ is modified); a patch that no longer matches fails loudly -- update it with the kernel.
Used for the phase tables in profiles/r02_findings.md.   usage: python tools/trace_build.py && gpurun -- python tools/trace_run.py c5"""
import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'diffusion-ccsp_amd', 'csrc')


def rep(s, a, b, cnt=1):
    assert s.count(a) == cnt, 'patch does not match (%d occurrences): %s' % (s.count(a), a[:100])
    return s.replace(a, b)


def main():
    tmp = tempfile.mkdtemp(prefix='ccsp_trace_')
    for f in os.listdir(SRC):
        if f.endswith(('.h', '.hip')):
            shutil.copy(os.path.join(SRC, f), tmp)
    hip = open(os.path.join(tmp, 'ccsp_hip.hip')).read()
    h2 = open(os.path.join(tmp, 'ccsp_f16x2.h')).read()
    hip = rep(hip, 'namespace {\n\nthread_local char g_err[512] = "";',
              '__device__ unsigned long long g_trace[3 * 256 * 32];\n#define TRK(kern, k) do { if (threadIdx.x == 0 && (blockIdx.x & 7) == 0 && blockIdx.x < 2048) '
              'g_trace[((kern) * 256 + (blockIdx.x >> 3)) * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)\nnamespace {\n\nthread_local char g_err[512] = "";')
    hip = rep(hip, 'int ccsp_profile_enable(ccsp_graph* g, int32_t on) {',
              'int ccsp_debug_trace(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * 3 * 256 * 32) '
              '== hipSuccess ? 0 : 1; }\nint ccsp_profile_enable(ccsp_graph* g, int32_t on) {')
    # k_node<256, true>
    hip = rep(hip, '    __builtin_amdgcn_s_setprio(3);\n    if (a.tab) {', '    TRK(2, 0);\n    __builtin_amdgcn_s_setprio(3);\n    if (a.tab) {')
    hip = rep(hip, '    if (!a.do_encode) return;\n    __syncthreads();\n', '    if (!a.do_encode) return;\n    __syncthreads();\n    TRK(2, 1);\n')
    hip = rep(hip, '    else encode_tile_mfma<H>(w, pf, xs, reinterpret_cast<float (*)[H / 2 + 1]>(s1raw), smax, node0, a.N, eo);\n}',
              '    else encode_tile_mfma<H>(w, pf, xs, reinterpret_cast<float (*)[H / 2 + 1]>(s1raw), smax, node0, a.N, eo);\n    TRK(2, 5);\n}')
    hip = rep(hip, '            s1h[(NODE_TILE + n) * LD + j] = h2;\n        }\n    }\n    __syncthreads();\n',
              '            s1h[(NODE_TILE + n) * LD + j] = h2;\n        }\n    }\n    __syncthreads();\n    TRK(2, 2);\n')
    hip = rep(hip, '    const int eu = -(sexp[lane & 15] + w.w2_exp);', '    TRK(2, 3);\n    const int eu = -(sexp[lane & 15] + w.w2_exp);')
    hip = rep(hip, '        if (lane < NODE_TILE) smax[wave][lane] = m;\n        __syncthreads();\n',
              '        if (lane < NODE_TILE) smax[wave][lane] = m;\n        __syncthreads();\n        TRK(2, 4);\n')
    # k_rowgemm_h2<256, 512, 3 | 4>
    h2 = rep(h2, '    if (ref.tab) tau_t += (size_t)ref.tab[*ref.counter].t * tau_stride;      // hipGraph mode: timestep from the device table\n',
             '    if (KD == 256) TRK(0, 0);\n    if (ref.tab) tau_t += (size_t)ref.tab[*ref.counter].t * tau_stride;\n')
    h2 = rep(h2, '            h2_epilogue_prefetch<ND>(bs0, 0, wr0, nrows, row0, col0 + wn * 64, base);\n#pragma unroll\n            for (int c = 0; c < D; ++c) glds(c, c);',
             '            if (KD == 256) TRK(0, 1);\n            h2_epilogue_prefetch<ND>(bs0, 0, wr0, nrows, row0, col0 + wn * 64, base);\n#pragma unroll\n'
             '            for (int c = 0; c < D; ++c) glds(c, c);')
    h2 = rep(h2, '                __builtin_amdgcn_s_barrier();                         // chunk c has landed for every wave; stage (c-1) % NST is free\n'
             '                __builtin_amdgcn_sched_barrier(0);\n',
             '                __builtin_amdgcn_s_barrier();\n                __builtin_amdgcn_sched_barrier(0);\n                if (KD == 256) TRK(0, 2 + c);\n')
    h2 = rep(h2, '            __builtin_amdgcn_s_barrier();                             // every wave is done reading the stages\n            __builtin_amdgcn_sched_barrier(0);\n',
             '            __builtin_amdgcn_s_barrier();\n            __builtin_amdgcn_sched_barrier(0);\n            if (KD == 256) TRK(0, 10);\n')
    h2 = rep(h2, '                         (tau_t && (ts & 1) == 0) ? tau_t + (size_t)(ts >> 1) * ND : nullptr, U, umax, 2 * NCT, 2 * ct + wn);\n}',
             '                         (tau_t && (ts & 1) == 0) ? tau_t + (size_t)(ts >> 1) * ND : nullptr, U, umax, 2 * NCT, 2 * ct + wn);\n    if (KD == 256) TRK(0, 12);\n}')
    # k_rowgemm_h2<256, 512, 0 | 2>: the forms C2-sized batches run
    h2 = rep(h2, '        glds(0, 0);\n        __syncthreads();\n#pragma unroll\n        for (int c = 0; c < NCH; ++c) {\n            if (c + 1 < NCH) glds(c + 1, (c + 1) & 1);',
             '        if (KD == 256) TRK(0, 1);\n        glds(0, 0);\n        __syncthreads();\n#pragma unroll\n        for (int c = 0; c < NCH; ++c) {\n            if (c + 1 < NCH) glds(c + 1, (c + 1) & 1);')
    h2 = rep(h2, '            __syncthreads();                                      // (drains the LDS-DMA of chunk c+1 as well)\n',
             '            __syncthreads();\n            if (KD == 256) TRK(0, 2 + c);\n')
    h2 = rep(h2, '        gload(0, 0);\n        lstore(0, 0);\n        gload(1, 0);\n        __syncthreads();\n        for (int c = 0; c < NCH; ++c) {',
             '        if (KD == 256) TRK(0, 1);\n        gload(0, 0);\n        lstore(0, 0);\n        gload(1, 0);\n        __syncthreads();\n        for (int c = 0; c < NCH; ++c) {')
    h2 = rep(h2, '            __syncthreads();                                      // every wave is done reading the stage\n',
             '            __syncthreads();\n            if (KD == 256) TRK(0, 2 + c);\n')
    h2 = rep(h2, '        asm volatile("" ::: "memory");                            // (compiler ordering only: the LDS runs one wave\'s operations in order)\n',
             '        asm volatile("" ::: "memory");\n        if (ND == 512) TRK(0, 13 + 2 * i);\n')
    h2 = rep(h2, '        asm volatile("" ::: "memory");\n    }\n}\n\n// s_waitcnt vmcnt(n) lgkmcnt(0) with n known',
             '        asm volatile("" ::: "memory");\n        if (ND == 512) TRK(0, 14 + 2 * i);\n    }\n    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    if (ND == 512) TRK(0, 17);\n}\n\n// s_waitcnt vmcnt(n) lgkmcnt(0) with n known')
    # k_edge_h2
    h2 = rep(h2, '    constexpr int H = 256, BN = 128, NCH = H / H2_BK;\n    constexpr int ME = 32 * MT, ROWS = 2 * ME;',
             '    TRK(1, 0);\n    constexpr int H = 256, BN = 128, NCH = H / H2_BK;\n    constexpr int ME = 32 * MT, ROWS = 2 * ME;')
    h2 = rep(h2, '    gload_a(0, 0);\n    gload_b(0);\n    gload_a(1, 1);\n    floatx16 acc[MT][2];', '    TRK(1, 1);\n    gload_a(0, 0);\n    gload_b(0);\n    gload_a(1, 1);\n    floatx16 acc[MT][2];')
    h2 = rep(h2, '    gload_b(1);\n    gload_a(2, 0);\n    __syncthreads();\n#pragma unroll\n    for (int c = 0; c < NCH; ++c) {                               // fully unrolled: the register-set index is a constant',
             '    gload_b(1);\n    gload_a(2, 0);\n    __syncthreads();\n    TRK(1, 2);\n#pragma unroll\n    for (int c = 0; c < NCH; ++c) {')
    h2 = rep(h2, '        if (c + 3 < NCH) gload_a(c + 3, nx);\n        __syncthreads();\n    }\n    // epilogue, 64 rows per pass',
             '        if (c + 3 < NCH) gload_a(c + 3, nx);\n        __syncthreads();\n        TRK(1, 3 + c);\n    }\n    // epilogue, 64 rows per pass')
    h2 = rep(h2, '        __syncthreads();\n        if constexpr (L2 == 1) {                                               // (see h2_decoder_l2)',
             '        __syncthreads();\n        TRK(1, 11);\n        if constexpr (L2 == 1) {')
    h2 = rep(h2, '            else h2_decoder_l2<0>(S1, S1_LD, Wd2, P, RED, wave, lane);\n            __syncthreads();\n',
             '            else h2_decoder_l2<0>(S1, S1_LD, Wd2, P, RED, wave, lane);\n            __syncthreads();\n            TRK(1, 12);\n')
    h2 = rep(h2, '    if constexpr (ENERGY) {\n        const float tot = block_sum_256(e2, reinterpret_cast<float*>(smem));\n        if (tid == 0) en.partial[blockIdx.x] = tot;\n    }\n}',
             '    TRK(1, 13);\n    if constexpr (ENERGY) {\n        const float tot = block_sum_256(e2, reinterpret_cast<float*>(smem));\n        if (tid == 0) en.partial[blockIdx.x] = tot;\n    }\n}')
    hip = hip.replace('#include "../../include/ccsp.h"', '#include "%s"' % os.path.join(ROOT, 'include', 'ccsp.h'))
    open(os.path.join(tmp, 'ccsp_hip.hip'), 'w').write(hip)
    open(os.path.join(tmp, 'ccsp_f16x2.h'), 'w').write(h2)
    out = os.path.join(ROOT, 'tools', 'abl_trace.so')
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-pthread', '-I', os.path.join(ROOT, 'include'),
                           '-I', SRC, '-o', out, os.path.join(tmp, 'ccsp_hip.hip')])
    shutil.rmtree(tmp, ignore_errors=True)
    print('built', out)


if __name__ == '__main__':
    main()
