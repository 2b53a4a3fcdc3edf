"""Builds tools/abl_trace2.so: the product kernels with -DCCSP_TRACE2 (ccsp_hip.hip: per-workgroup phase stamps of k_rowgemm_h2 through LDS, the
hardware slot of every workgroup), at the product residency.  usage: python tools/trace2_build.py && gpurun -- python tools/trace2_run.py 256"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trace_build import ROOT, build

if __name__ == '__main__':
    build(out=os.path.join(ROOT, 'tools', 'abl_trace2.so'), defines=('CCSP_TRACE2',) + tuple(sys.argv[1:]))
