#!/usr/bin/env python3
"""Lint of the gfx950 assembly of the kernels that prefetch with inline-asm loads and hand-counted s_waitcnt
(diffusion-ccsp_amd/_lib.py GUARDED_KERNELS): reports every instruction that reads or writes a vector register which is the
destination of a global load still in flight -- not yet covered by an `s_waitcnt vmcnt(N)` on the path that reaches it.  hipcc
does not see asm loads: it may place a copy (v_mov), a spill or a reuse of such a register in front of the wait (round 4: NaN in
every transformer parity test from `v_mov` copies that merged two control-flow paths).

Method: basic blocks from labels and s_branch / s_cbranch_*; forward may-analysis to a fixpoint, state = for every register with a
load in flight the fewest vector-memory operations issued after it on any path (paths that disagree keep the smaller count: what a
hand-counted wait cannot rely on).  Loads and stores share the counter on gfx950 and retire in issue order; a store takes a slot
and has no destination; LDS-DMA loads (`... lds`) take a slot and write no register.

_lib.build() runs it on the device assembly of every build (-save-temps) and refuses a library with findings; by hand:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o ccsp.s diffusion-ccsp_amd/csrc/ccsp_hip.hip
    python tools/lint_inflight.py ccsp.s [kernel-name-substring ...]        exit status 1 when anything is found

The analysis does not evaluate branch conditions.  EXEMPT lists the one kernel form it misjudges for that reason: MODE 5 of
k_rowgemm_h2 (an experiment behind CCSP_ROW_MODE=5) writes `v = have_table ? asm_load(table) : row + offset`, whose two arms reach
the same label; the arm that loads never falls into the arm that computes.
"""
import re
import sys

GUARDED = ('k_rowgemm_h2', 'k_edge_h2', 'k_edge_bwd_h2', 'k_node_direct', 'k_node_energy_h2', 'k_eval_fused', 'k_sd_gemm_h2')
REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')
VMEM = ('global_load', 'global_store', 'buffer_load', 'buffer_store', 'flat_load', 'flat_store', 'scratch_load', 'scratch_store', 'global_atomic', 'buffer_atomic', 'flat_atomic')
EXEMPT = ('k_rowgemm_h2ILi256ELi512ELi5E', 'k_rowgemm_h2ILi512ELi256ELi5E')
MAX_STATES = 200000


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return frozenset(out)


def vmcnt_of(ins):
    m = re.match(r's_waitcnt.*vmcnt\((\d+)\)', ins)
    if m:
        return int(m.group(1))
    m = re.fullmatch(r's_waitcnt\s+(0x[0-9a-fA-F]+|\d+)', ins)
    if m:                                                   # raw immediate: vmcnt = bits 3:0 and 15:14
        v = int(m.group(1), 0)
        return (v & 0xF) | (((v >> 14) & 0x3) << 4)
    return None


def parse_blocks(lines):
    """-> (blocks, label -> block index); a block: list of (line number, op, operand text, kind, payload)"""
    blocks, labels, cur = [[]], {}, 0
    for ln, raw in lines:
        ins = raw.split(';')[0].strip()
        if not ins or ins.startswith('.') and not ins.endswith(':'):
            continue
        if ins.endswith(':'):
            if blocks[-1]:
                blocks.append([])
            labels[ins[:-1]] = len(blocks) - 1
            continue
        op = ins.split()[0]
        rest = ins[len(op):]
        if op == 's_branch' or op.startswith('s_cbranch'):
            blocks[-1].append((ln, op, rest, 'branch', rest.strip()))
            blocks.append([])
            continue
        if op == 's_endpgm':
            blocks[-1].append((ln, op, rest, 'end', None))
            blocks.append([])
            continue
        n = vmcnt_of(ins) if op == 's_waitcnt' else None
        if op == 's_waitcnt':
            blocks[-1].append((ln, op, rest, 'wait', n))
            continue
        if op.startswith(VMEM):
            dest = frozenset()
            if 'load' in op and '_lds_' not in op and not re.search(r'\blds\b', rest):
                dest = regs(rest.split(',')[0])
            elif 'atomic' in op and 'sc0' in rest:           # returning atomic
                dest = regs(rest.split(',')[0])
            blocks[-1].append((ln, op, rest, 'vmem', (dest, regs(rest))))
            continue
        blocks[-1].append((ln, op, rest, 'alu', regs(rest)))
    return blocks, labels


def lint_kernel(lines):
    """may-analysis to a fixpoint: state = {register: fewest operations issued after its load on any path here}; an
    `s_waitcnt vmcnt(N)` completes the loads with at least N younger operations"""
    blocks, labels = parse_blocks(lines)
    nb = len(blocks)
    succ = []
    for bi, blk in enumerate(blocks):
        nxt = [bi + 1] if bi + 1 < nb else []
        if blk:
            ln, op, rest, kind, pay = blk[-1]
            if kind == 'end':
                nxt = []
            elif kind == 'branch':
                nxt = [] if op == 's_branch' else nxt
                if pay in labels:
                    nxt = nxt + [labels[pay]]
        succ.append(nxt)

    def run(bi, state, found):
        st = dict(state)
        for ln, op, rest, kind, pay in blocks[bi]:
            if kind == 'wait':
                if pay is not None:
                    st = {r: y for r, y in st.items() if y < pay}
                continue
            if kind in ('end', 'branch'):
                break
            used = pay[1] if kind == 'vmem' else pay
            if found is not None and st:
                hit = [r for r in used if r in st]
                if hit and ln not in found:
                    found[ln] = (op + rest, sorted(hit))
            if kind == 'vmem':
                st = {r: min(y + 1, 64) for r, y in st.items()}
                for r in pay[0]:
                    st[r] = 0
        return st

    inp = [None] * nb
    inp[0] = {}
    work = [0]
    steps = 0
    while work:
        bi = work.pop()
        steps += 1
        if steps > MAX_STATES:
            return {}, False
        out = run(bi, inp[bi], None)
        for n in succ[bi]:
            if inp[n] is None:
                inp[n] = dict(out)
                work.append(n)
                continue
            changed = False
            for r, y in out.items():
                if r not in inp[n] or inp[n][r] > y:
                    inp[n][r] = y
                    changed = True
            if changed:
                work.append(n)
    found = {}
    for bi in range(nb):
        if inp[bi] is not None:
            run(bi, inp[bi], found)
    return found, True


def split_kernels(text):
    cur, kernels = None, {}
    for i, l in enumerate(text.split('\n'), 1):
        m = re.match(r'^(_Z\S+):', l)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur:
            if l.startswith('.Lfunc_end'):
                cur = None
                continue
            kernels[cur].append((i, l))
    return kernels


def lint_text(text, want=GUARDED, exempt=EXEMPT):
    """-> (number of kernels checked, {kernel: {line: (instruction, registers)}}) for the kernels whose name contains one of `want`"""
    out, checked = {}, 0
    for name, lines in split_kernels(text).items():
        if not any(w in name for w in want) or any(e in name for e in exempt):
            continue
        checked += 1
        found, complete = lint_kernel(lines)
        if not complete:
            found = dict(found)
            found[0] = ('analysis did not converge', [])
        if found:
            out[name] = found
    return checked, out


def main():
    want = tuple(sys.argv[2:]) or GUARDED
    checked, bad = lint_text(open(sys.argv[1]).read(), want, () if sys.argv[2:] else EXEMPT)
    total = 0
    for name, found in bad.items():
        total += len(found)
        print('%s: %d uses of in-flight registers' % (name[:110], len(found)))
        for ln in sorted(found)[:8]:
            print('    line %d: %-72s in flight: %s' % (ln, found[ln][0][:72], found[ln][1][:8]))
    print('lint_inflight: %d findings in %d kernels' % (total, checked))
    return 1 if total else 0


if __name__ == '__main__':
    sys.exit(main())
