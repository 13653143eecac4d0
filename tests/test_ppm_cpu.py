"""Position-parallel scan image (include/acx_blob.h "ppm", built by csrc/acx_ppm.cpp): the CPU
restatement oracle/ppm_walk.c walks it position by position and must reproduce the reference's
iter() output (the pinned oracle, the reference-generated fixtures) on every dictionary shape."""
import random
import struct

import pytest

from helpers import build_pair, expected_pairs, load_json, dna_workload
from oracle import orc
import pyahocorasick_amd as acx

RANDOM = load_json("ref_random.json")
VECTORS = load_json("ref_vectors.json")


def _ppm_header(blob):
    off_ppm = struct.unpack_from("<Q", blob, 248)[0]
    if not off_ppm:
        return None
    names = ("magic", "K", "sym_bits", "pow2", "C", "F")
    d = dict(zip(names, struct.unpack_from("<6I", blob, off_ppm)))
    d["off_hot4"], d["off_cid"] = struct.unpack_from("<2Q", blob, off_ppm + 232)      # (acx_ppm_header: 18 uint32, 8 uint64, top_base[22], off_chains, then these)
    return d


def _case_values(c):
    keys = [bytes.fromhex(k) for k in c["keys_hex"]]
    return keys, c.get("values")


@pytest.mark.parametrize("c", RANDOM["cases"], ids=[c["id"] for c in RANDOM["cases"]])
def test_ppm_walk_matches_reference_fixtures(c):
    keys, values = _case_values(c)
    A, O = build_pair(keys, values if c["store"] not in ("length", "ints_default") else None, c["store"])
    blob = A.flat_image_bytes()
    assert _ppm_header(blob) is not None
    for h in c["hays"]:
        hay = bytes.fromhex(h["hay_hex"])
        assert orc.ppm_iter(blob, hay) == expected_pairs(h["iter"])
        assert orc.ppm_iter(blob, hay, fill=5) == expected_pairs(h["iter"])


def test_ppm_walk_randomised_vs_oracle():
    rng = random.Random(23)
    shapes = set()
    for trial in range(60):
        alpha = rng.choice([b"a", b"ab", b"abc", b"ACGT", b"ACGTN", bytes(range(97, 97 + 16)), bytes(range(97, 97 + 26)) + b" ",
                            bytes([0x61, 0x80, 0xFF, 0x00]), bytes(range(256))])
        maxlen = rng.choice([3, 9, 14, 40])
        keys = list({bytes(rng.choice(alpha) for _ in range(rng.randint(1, maxlen))) for _ in range(rng.randint(1, 200))})
        vals = [rng.randint(-2**40, 2**40) for _ in keys]
        A, O = build_pair(keys, vals)
        blob = A.flat_image_bytes()
        hdr = _ppm_header(blob)
        assert hdr is not None
        shapes.add((hdr["sym_bits"], hdr["pow2"], hdr["F"] > hdr["C"]))
        assert orc.ppm_check_hot(blob) == 0
        if hdr["sym_bits"] == 2:                                         # ACX_FLATTEN_HOT12: 12-byte hot cells (value AND id), no cid section — pinned against the 32-byte cells too
            b12 = A.flat_image_bytes(acx.ACX_FLATTEN_HOT12)
            h12 = _ppm_header(b12)
            assert h12["off_hot4"] != 0 and h12["off_cid"] == 0 and orc.ppm_check_hot(b12) == 0
            assert orc.ppm_iter(b12, b"".join(keys)[:200]) == O.iter(b"".join(keys)[:200])
        text_alpha = alpha if rng.random() < 0.5 else alpha + b"#"       # a byte no key contains
        for _ in range(8):
            hay = bytes(rng.choice(text_alpha) for _ in range(rng.randint(0, 300)))
            want = O.iter(hay)
            assert orc.ppm_iter(blob, hay) == want, (alpha, keys[:5])
            # the kernels ask the filter and pick the cell with the symbols "as they stand": what lies before a
            # haystack or behind a byte of no key is arbitrary there, and must not matter
            for fill in (1, 2):
                assert orc.ppm_iter(blob, hay, fill=fill) == want, (alpha, keys[:5], fill)
    assert {s[0] for s in shapes} == {2, 4, 8} and {s[1] for s in shapes} == {0, 1}


def test_ppm_walk_many_outputs_per_position():
    keys = [b"a" * n for n in range(1, 41)]
    A, O = build_pair(keys, list(range(100, 140)))
    blob = A.flat_image_bytes()
    hay = b"a" * 50 + b"b" + b"a" * 45
    assert orc.ppm_iter(blob, hay) == O.iter(hay)


def test_ppm_walk_dna_dictionary_deep_rows():
    """config-2-like dictionary at reduced size: cells with child and grandchild summaries, deep rows"""
    keys, reads = dna_workload(20000, 300, 150, seed=3)
    A, O = build_pair(list(keys))
    blob = A.flat_image_bytes()
    hdr = _ppm_header(blob)
    assert hdr["K"] == 4 and hdr["sym_bits"] == 2 and hdr["pow2"] == 1 and hdr["F"] == hdr["C"] + 1
    for r in reads[:300]:
        hay = r.tobytes()
        assert orc.ppm_iter(blob, hay) == O.iter(hay)


def test_hashed_copy_of_a_global_filter():
    """all 256 byte values in keys of three bytes and more: the filter (three 8-bit symbols, 2 MiB) lives in global memory and
    the image carries a hashed copy of it for LDS (include/acx_blob.h "gh") — ppm_check_hot pins it as exactly the image of
    the bitmap under the hash (no false negatives); a dictionary that fills it (the copy would reject nothing) gets none"""
    rng = random.Random(77)
    for shortest, want_gh in ((3, True), (1, False)):            # (one-byte keys: every position ends a key, the bitmap is full)
        keys = list({bytes(rng.randrange(256) for _ in range(rng.randint(shortest, 9))) for _ in range(4000)})
        A, O = build_pair(keys, list(range(len(keys))))
        blob = A.flat_image_bytes()
        off_ppm = struct.unpack_from("<Q", blob, 248)[0]
        magic, K, sb, pow2, C, F, g_global = struct.unpack_from("<7I", blob, off_ppm)
        assert (K, sb, C, F, g_global) == (256, 8, 2, 3, 1)
        assert (struct.unpack_from("<Q", blob, off_ppm + 256 - 8)[0] != 0) == want_gh
        assert orc.ppm_check_hot(blob) == 0
        for _ in range(5):
            hay = bytes(rng.randrange(256) for _ in range(400))
            assert orc.ppm_iter(blob, hay) == O.iter(hay)


def test_ppm_absent_when_not_asked_for():
    import pyahocorasick_amd as acx
    A, O = build_pair([b"he", b"she"])
    assert _ppm_header(A.flat_image_bytes(acx.ACX_FLATTEN_NO_PPM)) is None
    assert orc.ppm_iter(A.flat_image_bytes(acx.ACX_FLATTEN_NO_PPM), b"ushers") is None
    assert _ppm_header(A.flat_image_bytes()) is not None


def test_arithmetic_symbol_map_for_four_letter_alphabets():
    """alphabets of exactly four bytes that a shift tells apart get symbol = (byte >> s) & 3 (the stream kernel then
    needs no table lookup); others keep the table; results are the same either way"""
    rng = random.Random(5)
    for alpha, want_arith in ((b"ACGT", True), (b"acgt", True), (b"\x00\x01\x02\x03", True), (b"ACGU", True), (b"AEIM", True), (b"ABCDE", False), (b"ACG", False)):
        keys = list({bytes(rng.choice(alpha) for _ in range(rng.randint(2, 14))) for _ in range(150)})
        A, O = build_pair(keys, list(range(len(keys))))
        blob = A.flat_image_bytes()
        off_ppm = struct.unpack_from("<Q", blob, 248)[0]
        sym_arith, _gw, _g2w, sym_lut = struct.unpack_from("<4I", blob, off_ppm + 32)
        assert (sym_arith != 0) == want_arith, (alpha, sym_arith)
        if sym_arith:
            assert sorted(sym_lut.to_bytes(4, "little")) == sorted(alpha)
        assert orc.ppm_check_hot(blob) == 0
        for _ in range(6):
            hay = bytes(rng.choice(alpha + b"N") for _ in range(rng.randint(0, 200)))
            assert orc.ppm_iter(blob, hay) == O.iter(hay)
