#!/usr/bin/env python3
"""Approximate VGPR/AGPR liveness over a kernel's ISA (hipcc -S output): where is register pressure highest?

    python tools/isa_liveness.py file.s KERNEL_SUBSTRING [--top 15]

Straight-line backward liveness per basic block with a fixpoint over the CFG (labels / s_cbranch / s_branch).
First operand of an instruction = definition (stores, ds_write*, global_store*, buffer_store*, s_*, v_cmp*, exp: no
definition); an MFMA / v_mac / v_fmac / v_dot*c / v_pk_fmac whose destination is also a source keeps it live.
Partial writes of a tuple (v_mov to one element) are treated per 32-bit register, so they are exact."""
import re
import sys

REG = re.compile(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b')
NODEF = ('global_store', 'buffer_store', 'ds_write', 'ds_store', 'scratch_store', 'flat_store', 's_', 'v_cmp', 'v_cmpx',
         'global_load_lds', 'buffer_load_lds', 'exp', 'v_nop', 'buffer_wbl2', 'buffer_inv', 'v_writelane')
ACCUM = ('v_mfma', 'v_mac', 'v_fmac', 'v_dot2c', 'v_dot4c', 'v_dot8c', 'v_pk_fmac', 'v_smfmac')


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def parse(lines):
    ins = []
    for ln in lines:
        s = ln.split(';')[0].strip()
        if not s or s.startswith('.') and not s.endswith(':'):
            continue
        if s.endswith(':'):
            ins.append(('label', s[:-1], set(), set(), ln))
            continue
        parts = s.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
        # re-join "v[1:2]" pieces split on commas inside brackets (none: ranges use ':')
        d, u = set(), set()
        if op.startswith(NODEF) and not op.startswith('v_writelane'):
            for o in ops:
                u |= regs(o)
        elif op.startswith('v_writelane'):
            d |= regs(ops[0]); u |= regs(ops[0])
        else:
            if ops:
                d |= regs(ops[0])
            for o in ops[1:]:
                u |= regs(o)
            if op.startswith(ACCUM) and len(ops) >= 3 and not op.startswith('v_mfma'):
                u |= regs(ops[0])
            if op.startswith('v_readlane') or op.startswith('v_readfirstlane'):
                d = set()
        tgt = ops[0] if op.startswith(('s_cbranch', 's_branch')) and ops else None
        ins.append((op, tgt, d, u, ln))
    return ins


def main():
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 15
    txt = open(path).read().split('\n')
    start = next(i for i, l in enumerate(txt) if key in l and l.split(';')[0].strip().endswith(':') and not l.startswith('.'))
    end = next(i for i in range(start, len(txt)) if txt[i].startswith('.Lfunc_end'))     # (a kernel may hold several s_endpgm)
    ins = parse(txt[start + 1:end + 1])
    n = len(ins)
    labels = {t: i for i, (op, t, _, _, _) in enumerate(ins) if op == 'label'}
    succ = []
    for i, (op, t, _, _, _) in enumerate(ins):
        s = []
        if op.startswith('s_branch'):
            s = [labels[t]] if t in labels else []
        elif op.startswith('s_endpgm'):
            s = []
        else:
            if i + 1 < n:
                s.append(i + 1)
            if op.startswith('s_cbranch') and t in labels:
                s.append(labels[t])
        succ.append(s)
    live_in = [set() for _ in range(n)]
    changed = True
    it = 0
    while changed and it < 50:
        changed = False
        it += 1
        for i in range(n - 1, -1, -1):
            out = set()
            for s in succ[i]:
                out |= live_in[s]
            new = (out - ins[i][2]) | ins[i][3]
            if new != live_in[i]:
                live_in[i] = new
                changed = True
    cnt = [len(x) for x in live_in]
    print(f"{n} instructions, peak live = {max(cnt)} (v {max(len([r for r in x if r[0]=='v']) for x in live_in)}, "
          f"a {max(len([r for r in x if r[0]=='a']) for x in live_in)})")
    # profile: pressure at every MFMA-count milestone and barrier
    nm = 0
    print("pressure at barriers / every 48th MFMA:")
    for i, (op, t, d, u, ln) in enumerate(ins):
        if op.startswith('v_mfma'):
            nm += 1
            if nm % 48 == 1:
                print(f"   mfma #{nm:4d}  live {cnt[i]:4d}")
        if op == 's_barrier':
            print(f"   s_barrier (after mfma #{nm})  live {cnt[i]:4d}")
    order = sorted(range(n), key=lambda i: -cnt[i])[:top]
    print("highest-pressure instructions:")
    for i in sorted(order):
        print(f"   [{i:5d}] live {cnt[i]:4d}  {ins[i][4].strip()[:110]}")


if __name__ == '__main__':
    main()
