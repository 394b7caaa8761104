#!/usr/bin/env python3
"""Condenses rocprofv3 output (tools/profile_gpu.sh) into gpurun_out/profiles_<tag>/<tag>_*.{csv,json}:
kernel stats, HBM traffic (FETCH_SIZE / WRITE_SIZE with the guide's gfx950 corrections) and the SQ
issue-side counters of the step kernel (per launch, per wave, per env-step)."""
import collections
import csv
import glob
import json
import os
import sys

CLOCK_GHZ = 2.4      # MI355X max shader clock (MI355X_MICROARCH.md); SQ cycle counters are quad-cycles
SIMDS = 256 * 4


def find(d, pat):
    r = glob.glob(os.path.join(d, '**', pat), recursive=True)
    return r[0] if r else None


def counters(path, want='step_kernel'):
    """{counter: mean value per dispatch} and dispatch count of the kernels whose name contains `want`."""
    acc = collections.defaultdict(lambda: [0, 0.0])
    name = None
    for r in csv.DictReader(open(path)):
        if want not in r['Kernel_Name']:
            continue
        name = r['Kernel_Name']
        a = acc[r['Counter_Name']]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
    return {k: s / n for k, (n, s) in acc.items()}, (max(n for n, _ in acc.values()) if acc else 0), name


def bench_line(out, name):
    try:
        return json.loads(open(os.path.join(out, name + '_bench.json')).read().strip().splitlines()[-1])
    except Exception as e:
        return {'error': str(e)}


def provenance():
    """What the profile was taken on: the build id of the library file that ran (a hash of csrc/ + include/igw.h + the
    compiler flags, gridworld_amd/build.py) and the git commit of the tree (the GPU box has no .git: profile_gpu.sh
    passes it in IGW_GIT_COMMIT).  bench.py compares `build_id` with igw_build_id() of the library it times."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    try:
        from gridworld_amd import build as B
        return {'build_id': B.built_id(), 'source_hash': B.source_hash(), 'git_commit': os.environ.get('IGW_GIT_COMMIT') or None}
    except Exception as e:  # noqa: BLE001
        return {'build_id': None, 'source_hash': None, 'git_commit': os.environ.get('IGW_GIT_COMMIT') or None, 'error': str(e)}


def main():
    out, tag = sys.argv[1], sys.argv[2]
    dst = os.path.join('gpurun_out', 'profiles_' + tag)
    os.makedirs(dst, exist_ok=True)
    prov = provenance()
    summary = {'tag': tag, **prov}
    ks = find(os.path.join(out, 'kt'), '*kernel_stats.csv')
    kernel_ns = None
    if ks:
        rows = list(csv.DictReader(open(ks)))
        with open(os.path.join(dst, f'{tag}_kernel_stats.csv'), 'w') as f:
            f.write(open(ks).read())
        summary['kernel_stats'] = rows[:12]
        for r in rows:
            if 'step_kernel' in r.get('Name', ''):
                kernel_ns = float(r['AverageNs'])
                break
    summary['bench_under_kernel_trace'] = bench_line(out, 'kt')
    for name, counter in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
        cc = find(os.path.join(out, name), '*counter_collection.csv')
        if cc:
            c, n, k = counters(cc)
            if counter in c:
                summary[counter] = {'kernel': k, 'dispatches': n, 'mean': c[counter]}
    # HBM bytes per launch of the step kernel, with the guide's gfx950 correction:
    # FETCH_SIZE is in KiB and reports 1/2 of wide coalesced reads -> x2; WRITE_SIZE in KiB as is.
    summary['envs'] = ((summary['bench_under_kernel_trace'].get('config') or {}).get('envs_per_gpu'))
    summary['kernel_avg_ns'] = kernel_ns
    if 'FETCH_SIZE' in summary and 'WRITE_SIZE' in summary:
        fetch_kib, write_kib = summary['FETCH_SIZE']['mean'], summary['WRITE_SIZE']['mean']
        summary['hbm_bytes_per_launch'] = fetch_kib * 1024 * 2 + write_kib * 1024
        summary['hbm_read_bytes_per_launch_corrected'] = fetch_kib * 1024 * 2
        summary['hbm_read_bytes_per_launch_raw'] = fetch_kib * 1024
        summary['hbm_write_bytes_per_launch'] = write_kib * 1024
        summary['correction'] = ('FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read under-count, MI355X_MICROARCH.md HBM '
                                 'section) + WRITE_SIZE KiB x 1024')
    with open(os.path.join(dst, f'{tag}_traffic.json'), 'w') as f:
        json.dump(summary, f, indent=1)

    # ---- issue side: SQ counters of the step kernel
    raw, kname, disp = {}, None, 0
    for name in ('sq_insts', 'sq_cycles', 'sq_misc'):
        cc = find(os.path.join(out, name), '*counter_collection.csv')
        if cc:
            c, n, k = counters(cc)
            raw.update(c)
            kname, disp = k or kname, max(disp, n)
    issue = {'tag': tag, **prov, 'kernel': kname, 'dispatches_averaged': disp, 'raw': raw}
    b = bench_line(out, 'sq_insts')
    n_envs = (b.get('config') or {}).get('envs_per_gpu')
    waves = raw.get('SQ_WAVES')
    if waves and n_envs:
        per_wave = lambda k: raw[k] / waves if k in raw else None  # noqa: E731
        issue.update({
            'envs': n_envs, 'waves_per_launch': waves, 'waves_per_simd': waves / SIMDS,
            'valu_insts_per_wave': per_wave('SQ_INSTS_VALU'), 'salu_insts_per_wave': per_wave('SQ_INSTS_SALU'),
            'lds_insts_per_wave': per_wave('SQ_INSTS_LDS'),
            'vmem_rd_insts_per_wave': per_wave('SQ_INSTS_VMEM_RD'), 'vmem_wr_insts_per_wave': per_wave('SQ_INSTS_VMEM_WR'),
            'valu_insts_per_env_step': raw.get('SQ_INSTS_VALU', 0) / n_envs,
            'salu_insts_per_env_step': raw.get('SQ_INSTS_SALU', 0) / n_envs,
        })
        if 'SQ_WAVE_CYCLES' in raw:  # quad-cycles -> cycles
            wc = raw['SQ_WAVE_CYCLES'] * 4
            issue['wave_cycles_mean'] = wc / waves
            for k, nm in (('SQ_WAIT_ANY', 'frac_wave_time_parked_on_waitcnt'),
                          ('SQ_WAIT_INST_ANY', 'frac_wave_time_issue_stalled'),
                          ('SQ_ACTIVE_INST_ANY', 'frac_wave_time_issuing'),
                          ('SQ_ACTIVE_INST_VALU', 'frac_wave_time_issuing_valu'),
                          ('SQ_ACTIVE_INST_LDS', 'frac_wave_time_issuing_lds')):
                if k in raw:
                    issue[nm] = raw[k] * 4 / wc
        if kernel_ns:
            cyc = kernel_ns * CLOCK_GHZ
            issue['kernel_avg_ns'] = kernel_ns
            issue['kernel_cycles_at_2p4ghz'] = cyc
            if 'SQ_ACTIVE_INST_VALU' in raw:
                # SIMD VALU-issue occupancy: VALU-active wave cycles per SIMD over the launch's cycles
                issue['valu_issue_util_per_simd'] = raw['SQ_ACTIVE_INST_VALU'] * 4 / SIMDS / cyc
            issue['valu_insts_per_s'] = raw.get('SQ_INSTS_VALU', 0) / (kernel_ns * 1e-9)
            # peak: one VALU wave-instruction per 2 cycles per SIMD with >= 2 waves (MI355X_MICROARCH.md, SIMD-32)
            issue['valu_insts_per_s_peak'] = SIMDS * CLOCK_GHZ * 1e9 / 2
            issue['valu_issue_frac_of_peak'] = issue['valu_insts_per_s'] / issue['valu_insts_per_s_peak']
    with open(os.path.join(dst, f'{tag}_issue.json'), 'w') as f:
        json.dump(issue, f, indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k not in ('kernel_stats',)}, indent=1)[:2500])
    print(json.dumps({k: v for k, v in issue.items() if k != 'raw'}, indent=1))
    print(json.dumps(raw, indent=1))
    if ks:
        print(open(ks).read()[:2000])


if __name__ == '__main__':
    main()
