"""Builds tools/abl_<name>.so from a patched COPY of the kernel sources (nothing under diffusion-ccsp_amd/ is modified): timing-only
ablations and candidate variants for same-box A/B runs (tools/ab.sh, tools/abl_run.sh).  A patch that no longer matches fails loudly.
usage: python tools/mkvariant.py <name> [--trace]  with the patch list for <name> in VARIANTS below."""
import os, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'diffusion-ccsp_amd', 'csrc')

# name -> (patches [(file, old, new)], defines [-D...])
VARIANTS = {
    'base': ([], []),                                            # the sources as they are (build it from a stash / an older checkout for a same-call A/B)
    'gate_nofence': ([], ['CCSP_GATE_NOFENCE']),                 # relay gates without the cache-wide release / acquire (TIMING ONLY: results are stale)
    'act_scalar': ([], ['CCSP_ACT_SCALAR']),
    'try_mode2': ([], ['CCSP_TRY_MODE2']),                       # the product build + row GEMM MODEs 2 / 9 selectable by CCSP_ROW_MODE (direct-to-LDS staging and the software-pipelined form on the product's interleaved A planes)                     # edge kernels' activation producer without the packed fp32 instructions
    'cb22': ([], ['CCSP_H2_CB0=2', 'CCSP_H2_CB1=2']),          # row GEMM MODE 2: base of both row tiles requested under chunk NCH - 2
    'cb88': ([], ['CCSP_H2_CB0=8', 'CCSP_H2_CB1=8']),          # ... under chunk 0
    'cb11': ([], ['CCSP_H2_CB0=1', 'CCSP_H2_CB1=1']),          # ... under the last chunk
    'cb42': ([], ['CCSP_H2_CB0=4', 'CCSP_H2_CB1=2']),
    # bisecting the one-ulp difference between k_node and k_node_direct
    'ni_philox': ([('ccsp_philox.h', '__device__ __forceinline__ float philox_normal(', '__device__ __attribute__((noinline)) float philox_normal(')], []),
    'pin_norm': ([('ccsp_kernels_node.h', '    if (a.normalize) acc = acc / sqrtf((float)csr_cnt);                   // 0/0 -> NaN like the reference', '    if (a.normalize) acc = __fdiv_rn(acc, __fsqrt_rn((float)csr_cnt));'),
                  ('ccsp_kernels_node.h', '                if (a.normalize) acc = acc / sqrtf((float)csr_cnt);            // 0/0 -> NaN like the reference', '                if (a.normalize) acc = __fdiv_rn(acc, __fsqrt_rn((float)csr_cnt));')], []),
    # finer stamps inside the epilogue's store loop (trace builds)
    'epi_stamps': ([('ccsp_f16x2.h', '                if constexpr (FWD) m = h2_max8(m);', '                CCSP_TRK(0, 18 + 4 * i + st);\n                if constexpr (FWD) m = h2_max8(m);'),
                    ('ccsp_f16x2.h', '        float4 cv[4][2];                                          // the lane', '        CCSP_TRK(0, 26 + i);\n        float4 cv[4][2];                                          // the lane')], []),
    'nostore': ([('ccsp_f16x2.h', '                    *reinterpret_cast<float4*>(up) = o[0];\n                    *reinterpret_cast<float4*>(up + 32) = o[1];',
                  '                    if (o[0].x == 123.456f) { *reinterpret_cast<float4*>(up) = o[0];\n                    *reinterpret_cast<float4*>(up + 32) = o[1]; }')], []),
}


def build(name, spec, trace=False):
    patches, defines = spec
    tmp = tempfile.mkdtemp(prefix='ccsp_var_')
    for f in os.listdir(SRC):
        if f.endswith(('.h', '.hip')):
            shutil.copy(os.path.join(SRC, f), tmp)
    for fn, a, b in patches:
        p = os.path.join(tmp, fn)
        s = open(p).read()
        assert s.count(a) == 1, 'patch does not match (%d occurrences): %s' % (s.count(a), a[:100])
        open(p, 'w').write(s.replace(a, b))
    hip = open(os.path.join(tmp, 'ccsp_hip.hip')).read()
    hip = hip.replace('#include "../../include/ccsp.h"', '#include "%s"' % os.path.join(ROOT, 'include', 'ccsp.h'))
    open(os.path.join(tmp, 'ccsp_hip.hip'), 'w').write(hip)
    out = os.path.join(ROOT, 'tools', 'abl_%s%s.so' % (name, '_trace' if trace else ''))
    # (the product build's flags, diffusion-ccsp_amd/_lib.py: -save-temps changes the pipeline the device code goes through, and an A/B must not
    # measure that)
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-pthread', '-save-temps', '--offload-compress'] +
                          (['-DCCSP_TRACE'] if trace else []) + ['-D' + d for d in defines] +
                          ['-I', os.path.join(ROOT, 'include'), '-I', SRC, '-o', out, os.path.join(tmp, 'ccsp_hip.hip'), '-ldl'], cwd=tmp)
    shutil.rmtree(tmp, ignore_errors=True)
    print('built', out)


if __name__ == '__main__':
    names = [a for a in sys.argv[1:] if not a.startswith('--')]
    for n in names:
        build(n, VARIANTS[n], '--trace' in sys.argv)
