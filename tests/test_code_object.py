"""The resource contract of the built HIP library, read from its gfx950 code object (no GPU needed): the step kernels
keep their env state in registers -- no scratch (private segment) memory -- and the four-lanes-per-env instantiations
fit four blocks per CU (<= 128 VGPRs, <= 40 KB LDS), which is what makes the 65,536-env batch ONE round of co-resident
wavefronts.  A `cond ? e.x : e.z` on struct FIELDS once put the env struct into scratch memory
and made the launch 74 % longer with every parity test still green; this test is the guard."""
import os
import re
import shutil
import subprocess

import pytest

LLVM = '/opt/rocm/lib/llvm/bin'


def _kernels(tmp_path):
    from gridworld_amd import build
    lib = build.build()
    tools = [os.path.join(LLVM, t) for t in ('llvm-objcopy', 'clang-offload-bundler', 'llvm-readelf')]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip('LLVM binutils of the ROCm toolchain not found')
    fat, co = str(tmp_path / 'fat.bin'), str(tmp_path / 'dev.co')
    subprocess.check_call([tools[0], '--dump-section', '.hip_fatbin=' + fat, lib])
    subprocess.check_call([tools[1], '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--input=' + fat,
                           '--output=' + co, '--unbundle'])
    notes = subprocess.check_output([tools[2], '--notes', co], text=True)
    out = {}
    for block in notes.split('- .agpr_count:')[1:]:
        name = re.search(r'\.name:\s+(\S+)', block).group(1)
        out[name] = {k: int(re.search(r'\.%s:\s+(\d+)' % k, block).group(1))
                     for k in ('private_segment_fixed_size', 'vgpr_count', 'sgpr_count', 'group_segment_fixed_size')}
    return out


def test_step_kernels_use_no_scratch_and_fit_four_blocks_per_cu(tmp_path):
    if shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'):
        pytest.skip('no hipcc')
    ks = _kernels(tmp_path)
    step = {n: v for n, v in ks.items() if 'step_kernel' in n}
    assert len(step) >= 42, sorted(step)
    for n, v in step.items():
        assert v['private_segment_fixed_size'] == 0, (n, v)
    for n, v in ks.items():   # nothing on the path spills: rollout, reset, task preparation, task evaluation
        if re.search(r'rollout_kernel|reset_kernel|prepare_tasks|task_eval|fill_actions', n):
            assert v['private_segment_fixed_size'] == 0, (n, v)
    four = {n: v for n, v in step.items() if 'step_kernelILi4E' in n}
    assert len(four) == 7, sorted(four)   # 3 action spaces x {plain, extras} + flying's whole-blocks variant
    for n, v in four.items():
        assert v['vgpr_count'] <= 128, (n, v)                   # 4 waves per SIMD
        assert v['group_segment_fixed_size'] <= 40 * 1024, (n, v)   # 4 blocks of the 160 KB of a CU
