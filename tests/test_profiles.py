"""The measurement contract's provenance chain, on the CPU: bench.py's `roofline.frac` is PMC bytes of a committed
profile over the kernel time of the run, so the profile must be of the kernels being timed.  Every profile summary
carries the build id of the library it was taken with (tools/summarize_profile.py); the library carries the hash of
its sources (igw_build_id); bench.py marks a profile of another build `stale`."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _write(d, name, **kw):
    with open(os.path.join(d, name), 'w') as f:
        json.dump(kw, f)


def test_bench_marks_a_profile_of_another_build_stale(tmp_path):
    import bench
    d = str(tmp_path)
    _write(d, 'r07_traffic.json', build_id='aaaa', hbm_bytes_per_launch=1e6, envs=65536)
    _write(d, 'r08_traffic.json', build_id='bbbb', hbm_bytes_per_launch=2e6, envs=65536)
    _write(d, 'r08_flying_traffic.json', build_id='cccc', hbm_bytes_per_launch=3e6, envs=65536)
    _write(d, 'r06_cdm_traffic.json', hbm_bytes_per_launch=4e6, envs=65536)     # a summary without a build id
    t = bench.load_profile('traffic.json', '', 'bbbb', d)
    assert t['hbm_bytes_per_launch'] == 2e6 and t['stale'] is False      # the latest round's, same build
    assert bench.load_profile('traffic.json', '', 'aaaa', d)['stale'] is True        # the library moved on
    assert bench.load_profile('traffic.json', '', None, d)['stale'] is True
    f = bench.load_profile('traffic.json', 'flying', 'cccc', d)
    assert f['hbm_bytes_per_launch'] == 3e6 and f['stale'] is False      # kinds do not mix
    assert bench.load_profile('traffic.json', 'cdm', 'cccc', d)['stale'] is True   # no build id: stale by definition
    assert bench.load_profile('issue.json', '', 'bbbb', d) is None


def test_library_build_id_is_the_hash_of_the_sources_on_disk():
    from gridworld_amd import _lib, build
    assert _lib.build_id() == build.source_hash() == build.built_id()
    assert not build.is_stale()


def test_committed_profiles_are_of_the_committed_kernels():
    """The newest committed traffic / issue summaries (walking and flying) were taken with a library built from the
    sources in this tree: a kernel change without a new profile fails here, not silently in the bench line.
    (Summaries of rounds 1-4 carry no build id; bench.py treats them as stale.)"""
    import bench
    from gridworld_amd import build
    here = build.source_hash()
    for kind in ('', 'flying'):
        for suffix in ('traffic.json', 'issue.json'):
            p = bench.load_profile(suffix, kind, here)
            assert p is not None, (kind, suffix)
            if p.get('build_id') is None:
                assert p['stale'] is True
                continue
            assert p['build_id'] == here and p['stale'] is False, \
                f"{p['_file']} was taken with build {p['build_id']}, the sources hash to {here}: run tools/profile_gpu.sh"
    # every stamped summary names its commit
    for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')):
        s = json.load(open(f))
        if s.get('build_id'):
            assert s.get('git_commit'), f


def test_roofline_object_names_the_bound_and_keeps_the_hbm_side(tmp_path):
    """bench.roofline_of on the CPU (fake measurements, fake profile summaries): with an SQ profile of the timed build the
    top level says `issue` with achieved = VALU instructions per wavefront x wavefronts per SIMD x 4 cycles over the run's
    kernel time, the PMC traffic fraction sits under `hbm` with its Infinity-Cache note, and the 2,097,152-env batch
    (`kind='large'`) takes its own profile unscaled; without an SQ profile the HBM object stands alone."""
    import types
    import bench
    from gridworld_amd import _lib
    d = str(tmp_path)
    bid = _lib.build_id()
    _write(d, 'r09_traffic.json', build_id=bid, git_commit='abc', hbm_bytes_per_launch=40.0e6, envs=65536, kernel_avg_ns=10000.0)
    _write(d, 'r09_issue.json', build_id=bid, valu_insts_per_wave=960.0, salu_insts_per_wave=300.0, kernel_avg_ns=10000.0)
    _write(d, 'r09_large_traffic.json', build_id=bid, git_commit='abc', hbm_bytes_per_launch=1.28e9, envs=2097152, kernel_avg_ns=280000.0)
    m = dict(kernel_ms=0.010, p_changed=0.0625, p_cell_changed=0.0625, resets_in_window=262 * 20)
    r = types.SimpleNamespace(N=65536, K=20, flying=False, workload='rt20')
    roof, iss = bench.roofline_of(r, m, 4, profiles_dir=d)
    assert roof['bound'] == 'issue' and roof['unit'].startswith('VALU issue cycles') and roof['stale'] is False
    assert abs(roof['achieved'] - 960.0 * 4 * 4 / (10e-6 * 2.4e9)) < 1e-9 and roof['frac'] == roof['achieved'] and roof['peak'] == 1.0
    hbm = roof['hbm']
    assert hbm['bound'] == 'hbm' and hbm['peak'] == 8000.0 and roof['traffic'] == hbm['traffic'] == 40.0e6
    assert abs(hbm['frac'] - 40.0e6 / 10e-6 / 1e9 / 8000.0) < 1e-12 and 'Infinity Cache' in hbm['note'] and 'fits' in hbm['note']
    assert roof['issue'] is iss and iss['stale'] is False
    # the large batch: its own profile (not the 65,536-env one scaled), DRAM regime in the note, no SQ profile of that kind
    rl = types.SimpleNamespace(N=2097152, K=20, flying=False, workload='rt20')
    ml = dict(kernel_ms=0.270, p_changed=0.0625, p_cell_changed=0.0625, resets_in_window=8400 * 20)
    big, iss_l = bench.roofline_of(rl, ml, 4, profiles_dir=d, kind='large')
    assert iss_l is None and big['bound'] == 'hbm' and big['traffic'] == 1.28e9 and 'exceeds' in big['note']
    assert abs(big['frac'] - 1.28e9 / 270e-6 / 1e9 / 8000.0) < 1e-12
    slim = bench.slim_roofline(roof)
    assert slim['bound'] == 'issue' and slim['hbm']['bound'] == 'hbm' and 'convention' not in slim['hbm']
