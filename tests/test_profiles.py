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
