"""the build's lint of registers whose inline-asm loads are in flight (diffusion-ccsp_amd/_asmlint.py): it must see the hazard that
broke k_sd_gemm_h2 in round 4 (a wait inside a branch -> v_mov copies in front of it) and accept the shapes the kernels use"""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('_asmlint', os.path.join(ROOT, 'diffusion-ccsp_amd', '_asmlint.py'))
lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lint)

HEAD = '_ZN12_GLOBAL__N_112k_sd_gemm_h2ILi1ELi64ELi2EEEv:\n'
TAIL = '\ts_endpgm\n.Lfunc_end0:\n'

# the round-4 miscompile, condensed: the `younger == 0` arm copies the set before its wait
BROKEN = HEAD + '''
	global_load_dwordx4 v[18:21], v[60:61], off
	global_load_dwordx4 v[22:25], v[62:63], off
	s_cmp_gt_i32 s26, s22
	s_cbranch_scc1 .LBB0_2
	v_mov_b64_e32 v[50:51], v[18:19]
	v_mov_b64_e32 v[52:53], v[20:21]
	s_waitcnt vmcnt(0)
	s_branch .LBB0_3
.LBB0_2:
	s_waitcnt vmcnt(1)
	v_mov_b64_e32 v[50:51], v[18:19]
	v_mov_b64_e32 v[52:53], v[20:21]
.LBB0_3:
	v_ldexp_f32 v20, v50, v74
''' + TAIL

# a software-pipelined loop: two sets, each waited for with the other one still in flight; stores share the counter
CLEAN = HEAD + '''
	global_load_dwordx4 v[18:21], v[60:61], off
	global_load_dwordx4 v[22:25], v[62:63], off
.LBB0_1:
	s_waitcnt vmcnt(1)
	v_ldexp_f32 v30, v18, v74
	ds_write_b128 v82, v[18:21]
	global_load_dwordx4 v[18:21], v[60:61], off
	s_waitcnt vmcnt(1)
	v_ldexp_f32 v31, v22, v74
	global_load_dwordx4 v[22:25], v[62:63], off
	global_load_lds_dwordx4 v[60:61], off
	s_cbranch_scc1 .LBB0_1
	s_waitcnt vmcnt(0)
	v_add_f32_e32 v1, v18, v22
	global_store_dwordx4 v[4:5], v[10:13], off
''' + TAIL


def test_lint_sees_a_copy_in_front_of_the_wait():
    checked, found = lint.lint_text(BROKEN)
    assert checked == 1 and len(found) == 1
    hits = list(found.values())[0]
    assert len(hits) == 2 and all('v_mov_b64' in v[0] for v in hits.values())
    assert sorted(r for v in hits.values() for r in v[1]) == [18, 19, 20, 21]


def test_lint_accepts_counted_waits_in_a_loop():
    checked, found = lint.lint_text(CLEAN)
    assert checked == 1 and not found


def test_lint_counts_stores_and_lds_dma_as_slots():
    # the store is younger than the load: vmcnt(1) covers the load; with vmcnt(2) it would not
    ok = HEAD + '\tglobal_load_dwordx4 v[18:21], v[60:61], off\n\tglobal_store_dwordx4 v[4:5], v[10:13], off\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32_e32 v1, v18\n' + TAIL
    bad = ok.replace('vmcnt(1)', 'vmcnt(2)')
    assert not lint.lint_text(ok)[1] and lint.lint_text(bad)[1]
    dma = HEAD + '\tglobal_load_lds_dwordx4 v[18:19], off\n\tv_lshl_add_u64 v[18:19], v[16:17], 0, 64\n' + TAIL     # (the operand is an address)
    assert not lint.lint_text(dma)[1]


def test_lint_only_looks_at_guarded_kernels():
    other = BROKEN.replace('k_sd_gemm_h2', 'k_something')
    assert lint.lint_text(other) == (0, {})
