"""Builds tools/abl_trace.so: the shipped kernels with their s_memtime phase stamps compiled in (-DCCSP_TRACE: the CCSP_TRK marks of
ccsp_hip.hip / ccsp_f16x2.h; the product build has none), for tools/trace_run.py.
usage: python tools/trace_build.py && gpurun -- python tools/trace_run.py 256"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'diffusion-ccsp_amd', 'csrc')


def build(src_dir=SRC, out=os.path.join(ROOT, 'tools', 'abl_trace.so'), defines=('CCSP_TRACE',)):
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-pthread'] + ['-D' + d for d in defines] +
                          ['-I', os.path.join(ROOT, 'include'), '-o', out, os.path.join(src_dir, 'ccsp_hip.hip')])
    print('built', out)


if __name__ == '__main__':
    build()
