#!/usr/bin/env python3
"""Static per-basic-block instruction census of one kernel in a hipcc -save-temps .s file.

    python tools/isa_blocks.py file.s 'step_kernelILi4ELi0E'

Prints one line per basic block: label, VALU (f64 / other), SALU, LDS, VMEM, waits, branch target.
Issue-cycle estimate per block uses MI355X_MICROARCH.md: f64 VALU 4 cycles, other VALU 2 cycles (>= 2 waves/SIMD)."""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + re.escape(pat) + r'\w*:', l))
    blocks, cur = [], {'label': 'entry', 'ins': []}
    for l in lines[start + 1:]:
        if l.startswith('.Lfunc_end'):
            break
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            blocks.append(cur)
            cur = {'label': m.group(1), 'ins': []}
            continue
        s = l.strip()
        if not s or s.startswith(';') or s.startswith('.'):
            continue
        cur['ins'].append(s.split(';')[0].strip())
    blocks.append(cur)
    tot = dict(f64=0, valu=0, salu=0, lds=0, vmem=0, wait=0)
    print(f'{"block":12s} {"f64":>4s} {"valu":>5s} {"salu":>5s} {"lds":>4s} {"vmem":>4s} {"wait":>4s} {"cyc":>6s}  branch')
    for b in blocks:
        c = dict(f64=0, valu=0, salu=0, lds=0, vmem=0, wait=0)
        br = ''
        for ins in b['ins']:
            op = ins.split()[0]
            if op.startswith('v_'):
                if '_f64' in op:
                    c['f64'] += 1
                else:
                    c['valu'] += 1
            elif op.startswith('ds_'):
                c['lds'] += 1
            elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
                c['vmem'] += 1
            elif op.startswith('s_waitcnt'):
                c['wait'] += 1
            elif op.startswith('s_'):
                c['salu'] += 1
                if op.startswith(('s_cbranch', 's_branch')):
                    br += ' ' + op.replace('s_cbranch_', '').replace('s_branch', 'jmp') + '->' + ins.split()[-1]
        for k in tot:
            tot[k] += c[k]
        cyc = 4 * c['f64'] + 2 * c['valu']
        print(f'{b["label"]:12s} {c["f64"]:4d} {c["valu"]:5d} {c["salu"]:5d} {c["lds"]:4d} {c["vmem"]:4d} {c["wait"]:4d} {cyc:6d} {br}')
    print('total', tot, 'valu issue cycles (static)', 4 * tot['f64'] + 2 * tot['valu'])


if __name__ == '__main__':
    main()
