"""Shared helpers for the test-suite (test infrastructure; may use oracle/)."""
import json
import os
import random

import numpy as np

import pyahocorasick_amd as acx
from oracle import orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STORE = {"ints": acx.STORE_INTS, "ints_default": acx.STORE_INTS, "length": acx.STORE_LENGTH, "any": acx.STORE_ANY}


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def i32(v):
    """what Py_BuildValue('i') makes of a pointer-width integer"""
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >= 1 << 31 else v


def build_pair(keys, values=None, store="ints"):
    """-> (product Automaton, Oracle) holding the same dictionary.
    values None => key ordinal; store 'ints_default' => reference default (1-based count)."""
    A = acx.Automaton(STORE[store])
    O = orc.Oracle()
    for i, k in enumerate(keys):
        if store == "length":
            A.add_word(k)
            O.add_word(k, len(k))
        elif store == "ints_default" or (values is not None and values[i] is None):
            default = len(O) + 1
            A.add_word(k)
            O.add_word(k, default)
        else:
            v = i if values is None else values[i]
            A.add_word(k, v)
            O.add_word(k, v)
    A.make_automaton()
    O.make_automaton()
    return A, O


def expected_pairs(pairs):
    return [(int(i), int(v)) for i, v in pairs]


from pyahocorasick_amd.workloads import dna_workload, dna_keys, dna_reads  # noqa: E402,F401
