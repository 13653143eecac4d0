"""pyahocorasick_amd — MI355X-native batch Aho-Corasick scan engine behind the
`ahocorasick.Automaton` API of WojciechMula/pyahocorasick (bytes build).

    from pyahocorasick_amd import Automaton, STORE_INTS
    A = Automaton(STORE_INTS); A.add_word(b"he", 1); A.make_automaton()
    list(A.iter(b"she"))                      # GPU scan, same tuples as the reference
    A.iter_batch([b"she", b"hers"])           # batch entry (new)

Product code path: Python (this package) -> ctypes -> libacx.so (C-ABI, include/acx.h)
-> HIP kernels (csrc/acx_kernels.hip).  No CPU search fallback exists.
"""
from .automaton import (  # noqa: F401
    Automaton, AutomatonSearchIter, AutomatonSearchIterLong, BatchResult,
    EMPTY, TRIE, AHOCORASICK, STORE_INTS, STORE_LENGTH, STORE_ANY,
    KEY_STRING, KEY_SEQUENCE, MATCH_EXACT_LENGTH, MATCH_AT_MOST_PREFIX, MATCH_AT_LEAST_PREFIX,
    unicode, load,
)
from ._lib import AcxError, AcxNoDevice, ACX_SCAN_ALL, ACX_SCAN_LONG, device_count  # noqa: F401
from ._lib import ACX_FLATTEN_NO_PPM, ACX_FLATTEN_WIDE, ACX_FLATTEN_NO_ITOP, ACX_FLATTEN_TABLE_HOST, ACX_FLATTEN_TABLE_DEVICE, ACX_FLATTEN_HOT12  # noqa: F401

__version__ = "0.1.0"
