"""A/B helper (tools only): run bench.py's measurement against an alternative build of the library.
usage: CCSP_SO=/path/to/variant.so python tools/bench_so.py [bench args]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diffusion_ccsp_amd  # noqa: E402
from diffusion_ccsp_amd import _lib  # noqa: E402

if os.environ.get('CCSP_SO'):
    _lib.SO = os.environ['CCSP_SO']
    _lib._stale = lambda *a: False
import bench  # noqa: E402

bench.main()
