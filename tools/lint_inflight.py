#!/usr/bin/env python3
"""command-line front of diffusion-ccsp_amd/_asmlint.py (uses of registers whose asm loads are still in flight); see its header"""
import importlib.util
import os
import sys

spec = importlib.util.spec_from_file_location('_asmlint', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'diffusion-ccsp_amd', '_asmlint.py'))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
sys.exit(mod.main())
