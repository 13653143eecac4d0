"""Build libacx.so (host trie + flattener + HIP kernels + C-ABI) for gfx950, in-tree.

    python -m pyahocorasick_amd.build            # or: from pyahocorasick_amd.build import build_libacx

hipcc cross-compiles without a GPU.  The .so lands next to this file so it travels with
the repository snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["acx_trie.cpp", "acx_ppm.cpp", "acx_persist.cpp", "acx_items.cpp", "acx_long.cpp", "acx_hostwalk.cpp", "acx_kernels.hip", "acx_ppm_kernels.hip", "acx_ppm_stream4.hip", "acx_build.hip", "acx_ws.hip", "acx_long.hip", "acx_capi.hip"]
HEADERS = [os.path.join(ROOT, "include", "acx.h"), os.path.join(ROOT, "include", "acx_blob.h"),
           os.path.join(CSRC, "acx_internal.h"), os.path.join(CSRC, "acx_kernels.h"), os.path.join(CSRC, "acx_trie_impl.h"),
           os.path.join(CSRC, "acx_ppm_layout.h"), os.path.join(CSRC, "acx_ppm_device.h"), os.path.join(CSRC, "acx_long.h")]
LIB = os.path.join(HERE, "libacx.so")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


FLAGS_STAMP = LIB + ".flags"


def needs_build():
    if not os.path.exists(LIB):
        return True
    want = os.environ.get("ACX_EXTRA_CFLAGS", "").strip()
    have = open(FLAGS_STAMP).read().strip() if os.path.exists(FLAGS_STAMP) else ""
    if want != have:                      # the library in place was built with other flags (e.g. a development build)
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build_libacx(force=False, verbose=True):
    """One object per source under build/obj (only what changed is compiled again: the kernels take a minute and a
    half, the rest seconds), then one link.  ACX_EXTRA_CFLAGS adds flags (a change of flags rebuilds everything)."""
    if not force and not needs_build():
        return LIB
    extra = os.environ.get("ACX_EXTRA_CFLAGS", "").split()
    objdir = os.path.join(ROOT, "build", "obj" + ("_" + "_".join(x.strip("-").replace("=", "_") for x in extra) if extra else ""))
    os.makedirs(objdir, exist_ok=True)
    base = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fvisibility-inlines-hidden", "-Wall", "-Wno-unused-function", *extra,
            # code object v5 loads on every ROCm >= 5.x runtime, including the HIP runtime that
            # PyTorch wheels bundle (a process must only ever hold ONE HIP runtime: see _lib.py)
            "-mcode-object-version=5",
            "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    newest_header = max(os.path.getmtime(h) for h in HEADERS)
    objs, procs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        ob = os.path.join(objdir, src + ".o")
        objs.append(ob)
        if force or not os.path.exists(ob) or os.path.getmtime(ob) < max(os.path.getmtime(sp), newest_header):
            cmd = base + ["-c", sp, "-o", ob]
            if verbose:
                print("[pyahocorasick_amd.build]", " ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, "hipcc -c " + src)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-Wl,--version-script=" + os.path.join(CSRC, "libacx.map"), "-o", LIB] + objs
    if verbose:
        print("[pyahocorasick_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(FLAGS_STAMP, "w") as f:
        f.write(os.environ.get("ACX_EXTRA_CFLAGS", "").strip())
    return LIB


DROPIN_DIR = os.path.join(ROOT, "dropin")


def dropin_path(unicode=False):
    import sysconfig
    d = os.path.join(DROPIN_DIR, "unicode") if unicode else DROPIN_DIR
    return os.path.join(d, "ahocorasick" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_dropin(force=False, verbose=True, unicode=False):
    """Build the CPython extension `ahocorasick`: the drop-in host side that exports
    PyInit_ahocorasick and calls libacx through the C-ABI.
      dropin/ahocorasick.cpython-*.so           bytes build  (keys and haystacks are bytes)
      dropin/unicode/ahocorasick.cpython-*.so   unicode build (str; -DACX_UNICODE_BUILD=1)
    Use it with  sys.path.insert(0, "<repo>/dropin")  (or ".../dropin/unicode"); import ahocorasick"""
    import sysconfig
    src = os.path.join(CSRC, "ahocorasick_module.cpp")
    out = dropin_path(unicode)
    build_libacx(force=False, verbose=verbose)
    if (not force and os.path.exists(out) and os.path.getmtime(out) > os.path.getmtime(src)
            and os.path.getmtime(out) > os.path.getmtime(os.path.join(ROOT, "include", "acx.h"))):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    rpath = "$ORIGIN/../../pyahocorasick_amd" if unicode else "$ORIGIN/../pyahocorasick_amd"
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-DACX_UNICODE_BUILD=%d" % (1 if unicode else 0),
           "-I" + sysconfig.get_paths()["include"], "-I" + os.path.join(ROOT, "include"),
           src, "-o", out, "-L" + HERE, "-l:libacx.so", "-Wl,-rpath," + rpath]
    if verbose:
        print("[pyahocorasick_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build_libacx(force="--force" in sys.argv)
    build_dropin(force="--force" in sys.argv)
    build_dropin(force="--force" in sys.argv, unicode=True)
    print(LIB)
    print(dropin_path())
    print(dropin_path(True))
