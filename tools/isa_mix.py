#!/usr/bin/env python3
"""Instruction mix of the kernels in a gfx950 assembly file (hipcc -S --cuda-device-only): per kernel, and for its hottest
loop (the longest stretch between a label and the backward branch to it), counts of MFMA / VALU by opcode / LDS / VMEM /
s_nop / s_waitcnt, and the MFMA-VALU interleave pattern of that loop (M = MFMA, v = VALU, n = s_nop, w = s_waitcnt,
d = LDS, g = global, | = s_barrier).   usage: isa_mix.py file.s [kernel-name-substring] [--pattern]"""
import collections
import re
import sys


def kernels(text):
    lines = text.splitlines()
    starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
    for (i, name) in starts:
        end = next((j for j in range(i, len(lines)) if lines[j].startswith('.Lfunc_end')), len(lines))
        yield name, lines[i + 1:end]


def classify(op):
    if op.startswith('v_mfma'): return 'M'
    if op.startswith('v_'): return 'v'
    if op.startswith('ds_'): return 'd'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'g'
    if op == 's_nop': return 'n'
    if op == 's_waitcnt': return 'w'
    if op == 's_barrier': return '|'
    return ''


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else ''
    pattern = '--pattern' in sys.argv
    for name, body in kernels(text):
        if want not in name:
            continue
        ins = []                      # (index, op) ; labels: name -> index
        labels = {}
        for l in body:
            l = l.strip()
            if not l or l.startswith((';', '//')):
                continue
            l = l.split(';')[0].strip()
            if not l:
                continue
            if l.endswith(':'):
                labels[l[:-1]] = len(ins)
                continue
            if l.startswith('.'):
                continue
            ins.append(l)
        # hottest loop: backward branch with the longest body
        best = (0, 0)
        for i, l in enumerate(ins):
            m = re.match(r's_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)', l)
            if m:
                tgt = labels.get(m.group(1) or m.group(2))
                if tgt is not None and tgt < i and i - tgt > best[1] - best[0]:
                    best = (tgt, i)
        for title, (a, b) in (('whole kernel', (0, len(ins))), ('longest loop', best)):
            c = collections.Counter(l.split()[0] for l in ins[a:b])
            valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
            print(f"{name[:70]}  [{title}: {b - a} instructions]  MFMA {sum(v for k, v in c.items() if k.startswith('v_mfma'))}  VALU {valu}  "
                  f"LDS {sum(v for k, v in c.items() if k.startswith('ds_'))}  VMEM {sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'scratch_')))}  "
                  f"s_nop {c['s_nop']}  s_waitcnt {c['s_waitcnt']}  s_barrier {c['s_barrier']}")
            print('    VALU by opcode: ' + ', '.join(f'{k} {v}' for v, k in sorted(((v, k) for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma')), reverse=True)[:16]))
        if pattern:
            a, b = best
            pat = ''.join(classify(l.split()[0]) for l in ins[a:b])
            for i in range(0, len(pat), 160):
                print('    ' + pat[i:i + 160])


main()
