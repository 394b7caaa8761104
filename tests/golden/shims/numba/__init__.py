"""Stand-in for numba (absent in this image): `jit` is the identity.

The reference decorates only `normalize()` (gridworld/utils.py:57-73), whose
body is `int(round(v))` x3, i.e. Python-3 round-half-to-even.
"""


def jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


njit = jit
