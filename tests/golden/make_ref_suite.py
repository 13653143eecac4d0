#!/usr/bin/env python3
"""Copy the reference's own test-suite (/root/reference/tests: test_*.py + pytestingutils.py) into
tests/golden/ref_suite/ — TEST INFRASTRUCTURE, committed so that it travels to the GPU box, where
/root/reference does not exist.  tests/test_gpu_ref_suite.py runs it against the drop-in module
(dropin/ and dropin/unicode/).  Nothing in the product imports these files.

    python tests/golden/make_ref_suite.py [/root/reference]
"""
import os
import shutil
import sys

SRC = os.path.join(sys.argv[1] if len(sys.argv) > 1 else "/root/reference", "tests")
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_suite")
os.makedirs(DST, exist_ok=True)
n = 0
for name in sorted(os.listdir(SRC)):
    if (name.startswith("test_") and name.endswith(".py")) or name == "pytestingutils.py":
        shutil.copyfile(os.path.join(SRC, name), os.path.join(DST, name))
        n += 1
with open(os.path.join(DST, "README.md"), "w") as f:
    f.write("Verbatim copies of the reference's tests (pyahocorasick v2.2.0, `tests/`), made by\n"
            "`tests/golden/make_ref_suite.py`.  Test infrastructure only: run by `tests/test_gpu_ref_suite.py`\n"
            "against the drop-in `ahocorasick` module.  pytest does not collect this directory on its own\n"
            "(`conftest.py` ignores it).\n")
# tests/test_issue_9.py reads the first 2 KiB of <suite>/../README.rst as its haystack: a stand-in of the same kind of
# text (the reference's own README is not copied)
with open(os.path.join(os.path.dirname(DST), "README.rst"), "w") as f:
    para = ("pyahocorasick_amd test fixture: plain English text that the reference's test_issue_9 slices into two thousand "
            "start offsets. The automaton in that test holds the single key SSSSS, which this text does not contain. ")
    f.write("Stand-in for README.rst\n=======================\n\n" + para * 12 + "\n")
print("copied %d files to %s" % (n, DST))
