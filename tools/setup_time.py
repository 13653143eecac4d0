#!/usr/bin/env python3
"""Host-side set-up cost of a dictionary, phase by phase (no GPU needed): key generation, add_word,
make_automaton (failure links), flatten (image + position-parallel section; ACX_PPM_TIMING=1 prints
the builder's own laps).  usage: tools/setup_time.py [n_keys] [dna|snort|text]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("ACX_PPM_TIMING", "1")
import pyahocorasick_amd as acx
from pyahocorasick_amd import workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "snort"
t = time.perf_counter()
if kind == "dna":
    keys = W.dna_keys(n, seed=0)
elif kind == "text":
    keys = W.text_keys(W.text_vocab(max(10 * n, 1000), seed=2), n, seed=3)
else:
    keys = W.snort_signatures(n, seed=5)
print("generate keys   %.2f s" % (time.perf_counter() - t), flush=True)
A = acx.Automaton(acx.STORE_INTS)
t = time.perf_counter()
if "--loop" in sys.argv:                  # the reference's way: one Python call per key
    for i, k in enumerate(keys):
        A.add_word(k, i)
    print("add_word loop   %.2f s" % (time.perf_counter() - t), flush=True)
else:
    A.add_words(keys, range(len(keys)))
    print("add_words       %.2f s" % (time.perf_counter() - t), flush=True)
t = time.perf_counter()
A.make_automaton()
print("make_automaton  %.2f s" % (time.perf_counter() - t), flush=True)
t = time.perf_counter()
b = A.flat_image_bytes()
print("flatten (+ppm)  %.2f s   image %.2f GB   cores %d" % (time.perf_counter() - t, len(b) / 1e9, os.cpu_count()), flush=True)
