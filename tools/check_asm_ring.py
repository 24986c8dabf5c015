"""Audit of the asm weight-fragment ring in a compiled .s (cdna_hip_programming.md 5.7 item 1):
between an asm `global_load_dwordx4 vA:B` and the asm `s_waitcnt vmcnt(N)` that retires it, no
compiler instruction may read or write vA..vB (a v_mov / spill there would copy garbage).

    python tools/check_asm_ring.py file.s

Linear scan per kernel: outstanding asm loads are retired in order by every asm wait (all but the
N youngest).  Branch targets are not followed; loop bodies are checked in their textual order,
which covers the steady state because the ring state at the loop end equals the one at its head.
"""
import re
import sys

VREG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')


def regs_of(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check(path):
    bad = 0
    kernel = None
    in_asm = False
    outstanding = []          # list of (line_no, set(regs))
    n_loads = n_waits = 0
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        if s.endswith(':') and s.startswith('_Z'):
            kernel, outstanding, in_asm = s[:-1], [], False
            continue
        if s.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if s.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not s or s.startswith(';') or s.startswith('.'):
            continue
        if 's_endpgm' in s:
            if outstanding:
                print(f'{kernel}: {len(outstanding)} asm loads never retired before s_endpgm (line {ln})')
                bad += 1
            outstanding = []
            continue
        if in_asm:
            if s.startswith('global_load_dwordx4'):
                dst = regs_of(s.split(',')[0])
                outstanding.append((ln, dst))
                n_loads += 1
            elif s.startswith('s_waitcnt'):
                m = re.search(r'vmcnt\((\d+)\)', s)
                if m:
                    keep = int(m.group(1))
                    outstanding = outstanding[len(outstanding) - keep:] if keep else []
                    n_waits += 1
            continue
        # compiler instruction: may not touch registers of outstanding asm loads.
        # (its own s_waitcnt vmcnt(0) retires everything)
        if s.startswith('s_waitcnt') and 'vmcnt(0)' in s:
            outstanding = []
            continue
        touched = regs_of(s)
        for lno, dst in outstanding:
            if touched & dst:
                print(f'{kernel}: line {ln}: `{s}` touches v{sorted(touched & dst)} of the asm load at line {lno}')
                bad += 1
    print(f'{path}: {n_loads} asm loads, {n_waits} asm waits, {bad} violations')
    return bad


if __name__ == '__main__':
    sys.exit(1 if sum(check(p) for p in sys.argv[1:]) else 0)
