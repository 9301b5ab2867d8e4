"""Process-wide generator bit-compatible with Python's ``random`` module (CPython 3.10 Mersenne Twister), living in the
native batch builder so that the C++ collate / sample collection consume it without the GIL.  ``pyrandom.seed(s)`` stands
where the reference calls ``random.seed(s)`` (main.py): dataset shuffles, query choice and history subsampling then draw
the same numbers in the same order as the reference's single-process loader."""
from . import _lib

_handle = None


def handle():
    global _handle
    if _handle is None:
        seed(None)
    return _handle


def seed(s=None):
    global _handle
    lib = _lib.load_data()
    if s is None:
        import os
        s = int.from_bytes(os.urandom(8), 'little')
    if _handle is None:
        _handle = lib.ps_rng_create(int(s) & (2 ** 64 - 1))
    else:
        lib.ps_rng_seed(_handle, int(s) & (2 ** 64 - 1))     # in place: holders of the handle stay valid


def randbelow(n):
    return int(_lib.load_data().ps_rng_randbelow(handle(), int(n)))


def random():
    return float(_lib.load_data().ps_rng_random(handle()))


def shuffle(x):
    """In-place ``random.shuffle`` of a Python list (host-side uses such as candidate shuffling)."""
    for i in range(len(x) - 1, 0, -1):
        j = randbelow(i + 1)
        x[i], x[j] = x[j], x[i]
