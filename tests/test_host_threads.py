"""make_automaton / flatten cut their passes over the trie into ranges for several host threads
(acx_trie.cpp: parallel_range).  The image must not depend on the thread count: same BFS numbering
(src/Automaton.c:582-637 visits the nodes in that order), same failure links, same output lists."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import hashlib, sys
sys.path.insert(0, %r)
import pyahocorasick_amd as acx
from pyahocorasick_amd import workloads as W
keys = W.snort_signatures(40000, seed=11)
A = acx.Automaton(acx.STORE_INTS)
for i, k in enumerate(keys):
    A.add_word(k, i)
A.make_automaton()
b = A.flat_image_bytes()
print(len(b), hashlib.sha1(b).hexdigest())
""" % ROOT


def _image_digest(threads):
    env = dict(os.environ, ACX_HOST_THREADS=str(threads))
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout.strip()


def test_image_is_the_same_for_any_number_of_host_threads():
    serial = _image_digest(1)
    assert serial.split()[0].isdigit() and int(serial.split()[0]) > 1 << 20
    assert _image_digest(4) == serial
    assert _image_digest(7) == serial
