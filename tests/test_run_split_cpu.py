"""acx_ppm_slot_first_tile / acx_ppm_tile_slot (pyahocorasick_amd/csrc/acx_ppm_layout.h): the unequal runs of a block's 16
waves.  The scan kernels cut a block's 16 * tpw tiles with the first function, k_ppm_gather finds a tile's wave with the
second, k_ppm_gather_pos recomputes every wave's run with the first: they must describe ONE partition — every tile in exactly
one slot, slots in order, the four slots of an age group equally long, the whole block covered.  Compiled with g++ (the
header is plain C++ on the host) and run here."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include <cstdint>
#include "acx_ppm_layout.h"
int main() {
    long checked = 0;
    for (uint32_t tpw = 8; tpw <= 300; tpw += (tpw < 40 ? 1 : 7))
        for (uint32_t pa = 0; pa <= 400; pa += 40)
            for (uint32_t pb = 0; pb <= pa; pb += 35) {
                const uint32_t a = (tpw * pa + 500) / 1000, b = (tpw * pb + 500) / 1000;
                if (a >= tpw) continue;
                if (acx_ppm_slot_first_tile(0, tpw, a, b) != 0 || acx_ppm_slot_first_tile(16, tpw, a, b) != 16 * tpw) { printf("ends %u %u %u\n", tpw, a, b); return 1; }
                for (uint32_t s = 0; s < 16; s++) {
                    const uint32_t lo = acx_ppm_slot_first_tile(s, tpw, a, b), hi = acx_ppm_slot_first_tile(s + 1, tpw, a, b);
                    if (hi <= lo) { printf("empty slot %u: %u %u %u\n", s, tpw, a, b); return 1; }
                    if ((s & 3) && hi - lo != acx_ppm_slot_first_tile(s, tpw, a, b) - acx_ppm_slot_first_tile(s - 1, tpw, a, b)) { printf("uneven group %u\n", s); return 1; }
                    for (uint32_t t = lo; t < hi; t++, checked++)
                        if (acx_ppm_tile_slot(t, tpw, a, b) != s) { printf("tile %u of slot %u -> %u (tpw %u a %u b %u)\n", t, s, acx_ppm_tile_slot(t, tpw, a, b), tpw, a, b); return 1; }
                }
            }
    printf("ok %ld\n", checked);
    return 0;
}
"""


def test_run_split_is_one_partition():
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.cpp"), os.path.join(d, "t")
        with open(src, "w") as f:
            f.write(SRC)
        subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "pyahocorasick_amd", "csrc"),
                        src, "-o", exe], check=True)
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr
