#!/usr/bin/env python3
"""Static instruction census of kernels in a built library's gfx950 code object (no GPU needed).

    python tools/isa_census.py gridworld_amd/libigw_hip.so 'step_kernelILi4ELi0ELb0ELb0' ['step_kernelILi4ELi1ELb0ELb1' ...]

Per kernel: instructions by class (VALU, f64 VALU, DPP, SDWA, SALU, LDS, global loads / LDS-DMA / stores / atomics,
waits, scratch) -- what tests/test_code_object.py cannot see: the length of the instruction stream."""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'


def disassemble(lib):
    d = tempfile.mkdtemp()
    fat, co = os.path.join(d, 'fat.bin'), os.path.join(d, 'dev.co')
    subprocess.check_call([f'{LLVM}/llvm-objcopy', '--dump-section', '.hip_fatbin=' + fat, lib])
    subprocess.check_call([f'{LLVM}/clang-offload-bundler', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                           '--input=' + fat, '--output=' + co, '--unbundle'])
    return subprocess.check_output([f'{LLVM}/llvm-objdump', '-d', co], text=True)


def census(text, pat):
    m = re.search(r'<(_Z\w*%s\w*)>:\n(.*?)(?=\n\n|\Z)' % re.escape(pat), text, re.S)
    if not m:
        return None
    c = collections.Counter()
    for line in m.group(2).splitlines():
        parts = line.strip().split()
        if not parts:
            continue
        o = parts[0]
        c['total'] += 1
        if o.startswith('global_store'):
            c['gstore'] += 1
        elif o.startswith('global_load_lds'):
            c['glds'] += 1
        elif o.startswith('global_load'):
            c['gload'] += 1
        elif o.startswith('global_atomic'):
            c['gatomic'] += 1
        elif o.startswith('scratch'):
            c['scratch'] += 1
        elif o.startswith('ds_'):
            c['lds'] += 1
        elif o.startswith('v_'):
            c['valu'] += 1
            if '_f64' in o:
                c['f64'] += 1
        elif o.startswith('s_waitcnt'):
            c['wait'] += 1
        elif o.startswith('s_'):
            c['salu'] += 1
        if 'dpp' in o:
            c['dpp'] += 1
        if 'sdwa' in o:
            c['sdwa'] += 1
    return m.group(1), c


if __name__ == '__main__':
    text = disassemble(sys.argv[1])
    for pat in sys.argv[2:]:
        r = census(text, pat)
        if r is None:
            print(pat, 'not found')
            continue
        print(pat, ' '.join(f'{k}={v}' for k, v in sorted(r[1].items())))
