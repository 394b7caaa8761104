#!/usr/bin/env python3
"""Condenses rocprofv3 output (tools/profile_gpu.sh) into profiles/<tag>_*.{csv,json,md}."""
import csv
import glob
import json
import os
import sys


def find(d, pat):
    r = glob.glob(os.path.join(d, '**', pat), recursive=True)
    return r[0] if r else None


def main():
    out, tag = sys.argv[1], sys.argv[2]
    dst = os.path.join('gpurun_out', 'profiles_' + tag)
    os.makedirs(dst, exist_ok=True)
    summary = {'tag': tag}
    ks = find(os.path.join(out, 'kt'), '*kernel_stats.csv')
    if ks:
        rows = list(csv.DictReader(open(ks)))
        with open(os.path.join(dst, f'{tag}_kernel_stats.csv'), 'w') as f:
            f.write(open(ks).read())
        summary['kernel_stats'] = rows[:12]
    try:
        summary['bench_under_kernel_trace'] = json.loads(open(os.path.join(out, 'kt_bench.json')).read().strip().splitlines()[-1])
    except Exception as e:
        summary['bench_under_kernel_trace'] = str(e)
    for name, counter in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
        cc = find(os.path.join(out, name), '*counter_collection.csv')
        if not cc:
            continue
        per_kernel = {}
        for r in csv.DictReader(open(cc)):
            if r.get('Counter_Name') != counter:
                continue
            k = r['Kernel_Name']
            v = float(r['Counter_Value'])
            a = per_kernel.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += v
        summary[counter] = {k[:120]: {'dispatches': n, 'sum': s, 'mean': s / n} for k, (n, s) in per_kernel.items()}
    # HBM bytes per launch of the step kernel, with the guide's gfx950 correction:
    # FETCH_SIZE is in KiB and reports 1/2 of wide coalesced reads -> x2; WRITE_SIZE in KiB as is.
    step = [k for k in summary.get('FETCH_SIZE', {}) if 'step_kernel' in k]
    if step and 'WRITE_SIZE' in summary:
        k = step[0]
        fetch_kib = summary['FETCH_SIZE'][k]['mean']
        write_kib = summary['WRITE_SIZE'].get(k, {'mean': 0.0})['mean']
        summary['hbm_bytes_per_launch'] = fetch_kib * 1024 * 2 + write_kib * 1024
        summary['hbm_read_bytes_per_launch_corrected'] = fetch_kib * 1024 * 2
        summary['hbm_write_bytes_per_launch'] = write_kib * 1024
        summary['correction'] = 'FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read under-count, MI355X_MICROARCH.md HBM section) + WRITE_SIZE KiB x 1024'
    with open(os.path.join(dst, f'{tag}_traffic.json'), 'w') as f:
        json.dump(summary, f, indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k not in ('kernel_stats',)}, indent=1)[:3000])
    if ks:
        print(open(ks).read()[:2500])


if __name__ == '__main__':
    main()
