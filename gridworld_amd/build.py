"""Builds gridworld_amd/libigw_hip.so (HIP kernels + C ABI) for gfx950 with hipcc.

    python -m gridworld_amd.build            # build if stale
    python -m gridworld_amd.build --force

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the reference computes
in IEEE binary64 with one rounding per operation; never add fast-math flags.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libigw_hip.so')
LIB_DIAG = os.path.join(HERE, 'libigw_hip_diag.so')  # -DIGW_DIAG: phase stamps + ablation switches (tools/ only)
SOURCES = ['igw_kernels.hip']
HEADERS = ['igw_device.h', 'igw_trig.h', 'igw_trig_lut.h', 'igw_trig_tables.h',
           os.path.join('..', '..', 'include', 'igw.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
         '-fno-fast-math', '-Wall', '-Wno-unused-variable', '-Wno-bitwise-instead-of-logical',
         # the step kernel's leading scalar arguments (the pointers of its input burst) arrive preloaded in SGPRs
         '-mllvm', '-amdgpu-kernarg-preload-count=7',
         # the counters are added by ONE lane per wavefront already (a ballot + popcount): the compiler's own wave-level
         # reduction in front of every atomic only adds a dozen instructions
         '-mllvm', '-amdgpu-atomic-optimizer-strategy=None']


def hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found')


def source_hash(diag=False):
    """Identity of a kernel build: sha256 over the kernel sources, include/igw.h and the compiler flags (16 hex
    digits).  It is compiled into the library (-DIGW_BUILD_ID, exported as igw_build_id()) and stamped on every
    profile summary under profiles/ (tools/summarize_profile.py), so a bench line can tell whether the PMC bytes it
    quotes were measured on the kernels it timed."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        h.update(os.path.basename(f).encode() + b'\0')
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(fh.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()[:16] + ('-diag' if diag else '')


def built_id(lib=LIB):
    """igw_build_id() of an existing library file, read without loading it (the id string follows a marker)."""
    try:
        with open(lib, 'rb') as f:
            data = f.read()
        i = data.find(b'igw-build-id:')
        if i < 0:
            return None
        j = data.index(b'\0', i)
        return data[i + len(b'igw-build-id:'):j].decode()
    except (OSError, ValueError):
        return None


def is_stale(lib=LIB, diag=False):
    """Missing, older than a source (mtime), or built from other sources than the ones on disk (build id)."""
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps) or built_id(lib) != source_hash(diag)


def build(force=False, extra_flags=(), verbose=False, diag=False):
    """Compiles under an exclusive file lock and installs the result with an atomic rename, so several
    ranks starting at once (torchrun) never see or write a half-built library.  diag=True builds the
    diagnostic variant (libigw_hip_diag.so, -DIGW_DIAG) next to the production library."""
    lib = LIB_DIAG if diag else LIB
    if diag:
        extra_flags = tuple(extra_flags) + ('-DIGW_DIAG',)
    extra_flags = tuple(extra_flags) + ('-DIGW_BUILD_ID="igw-build-id:%s"' % source_hash(diag),)
    if not force and not is_stale(lib, diag):
        return lib
    import fcntl
    with open(lib + '.lock', 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale(lib, diag):  # another process built it while we waited
                return lib
            tmp = f'{lib}.tmp.{os.getpid()}'
            cmd = [hipcc()] + FLAGS + list(extra_flags) + ['-o', tmp] + [os.path.join(CSRC, s) for s in SOURCES]
            if verbose:
                print(' '.join(cmd))
            subprocess.run(cmd, check=True)
            os.replace(tmp, lib)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return lib


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True, diag='--diag' in sys.argv))
