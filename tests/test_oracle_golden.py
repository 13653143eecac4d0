"""CPU suite, part 1: PIN THE ORACLE.

oracle/ac_oracle.c (the restatement of the reference algorithm) is checked against
 (a) the known-answer vectors of the reference's own tests (golden/ref_vectors.json),
 (b) fixtures produced by running the reference itself (golden/ref_random.json),
 (c) the reference module live, when oracle/_ref is present (build container).
Then the product's CPU-side pieces (libacx host trie + flattener) are checked against the
pinned oracle by walking the flat image with oracle/flat_walk.c — no GPU involved.
"""
import random

import pytest

import pyahocorasick_amd as acx

from helpers import build_pair, expected_pairs, i32, load_json
from oracle import orc


def _oracle_from(keys, values):
    O = orc.Oracle()
    for k, v in zip(keys, values):
        O.add_word(k, v)
    O.make_automaton()
    return O


VECTORS = load_json("ref_vectors.json")["vectors"]
RANDOM = load_json("ref_random.json")


@pytest.mark.parametrize("v", VECTORS, ids=[v["id"] for v in VECTORS])
def test_oracle_matches_reference_test_vectors(v):
    keys = [bytes.fromhex(k) for k in v["keys_hex"]]
    hay = bytes.fromhex(v["hay_hex"])
    O = _oracle_from(keys, range(len(keys)))
    s = 0 if v["start"] is None else v["start"]
    e = len(hay) if v["end"] is None else v["end"]
    got = O.iter_long(hay, s, e) if v["mode"] == "iter_long" else O.iter(hay, s, e)
    assert got == expected_pairs(v["expected"]), v["source"]


def _case_values(c):
    keys = [bytes.fromhex(k) for k in c["keys_hex"]]
    if c["store"] == "length":
        return keys, [len(k) for k in keys]
    if c["store"] == "ints_default":
        vals, seen = [], set()
        for k in keys:
            vals.append(len(seen) + 1)       # src/Automaton.c:238-242: count + 1 at insertion time
            seen.add(k)
        return keys, vals
    return keys, c["values"]


@pytest.mark.parametrize("c", RANDOM["cases"], ids=[c["id"] for c in RANDOM["cases"]])
def test_oracle_matches_reference_generated_fixtures(c):
    keys, values = _case_values(c)
    O = _oracle_from(keys, values)
    for h in c["hays"]:
        hay = bytes.fromhex(h["hay_hex"])
        assert O.iter(hay) == expected_pairs(h["iter"])
        assert O.iter(hay) == expected_pairs(h["find_all"])          # iter == find_all (tests/test_issue_56.py)
        assert O.iter_long(hay) == expected_pairs(h["iter_long"])
        if "slice" in h:
            s, e = h["slice"]["start"], h["slice"]["end"]
            assert O.iter(hay, s, e) == expected_pairs(h["slice"]["iter"])
            assert O.iter(hay, s, e) == expected_pairs(h["slice"]["find_all"])
            assert O.iter_long(hay, s, e) == expected_pairs(h["slice"]["iter_long"])
    # chunked streaming: state carried over, shift accumulated (src/AutomatonSearchIter.c:344-352)
    state, shift = 0, 0
    for part_hex, exp in zip(c["chunks"]["parts_hex"], c["chunks"]["iter_set"]):
        part = bytes.fromhex(part_hex)
        e, v, state = O.iter_arrays(part, state=state, shift=shift)
        assert list(zip(e.tolist(), v.tolist())) == expected_pairs(exp)
        shift += len(part)


@pytest.mark.parametrize("c", RANDOM["special"], ids=[c["id"] for c in RANDOM["special"]])
def test_oracle_matches_reference_special_cases(c):
    keys = [bytes.fromhex(k) for k in c["keys_hex"]]
    if c["store"] == "length":
        values = [len(k) for k in keys]
    else:
        values = [v if v is not None else i + 1 for i, v in enumerate(c["values"])]
    O = _oracle_from(keys, values)
    hay = bytes.fromhex(c["hay_hex"])
    assert O.iter(hay) == expected_pairs(c["iter"])
    assert O.iter_long(hay) == expected_pairs(c["iter_long"])


def test_oracle_vs_live_reference_randomised():
    """differential test against the reference module itself (build container only)"""
    ref = orc.load_reference()
    if ref is None:
        pytest.skip("oracle/_ref not present (reference sources absent on this machine)")
    rng = random.Random(7)
    for trial in range(60):
        alpha = rng.choice([b"ab", b"ACGT", bytes([0x61, 0x80, 0xFF]), bytes(range(256))])
        keys = list({bytes(rng.choice(alpha) for _ in range(rng.randint(1, 7))) for _ in range(rng.randint(1, 40))})
        vals = [rng.randint(-2**40, 2**40) for _ in keys]
        R = ref.Automaton(ref.STORE_INTS)
        for k, v in zip(keys, vals):
            R.add_word(k, v)
        R.make_automaton()
        O = _oracle_from(keys, vals)
        for _ in range(10):
            hay = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 120)))
            assert O.iter(hay) == list(R.iter(hay))
            assert O.iter_long(hay) == list(R.iter_long(hay))
            assert O.iter(hay, ignore_ws=True) == list(R.iter(hay, ignore_white_space=True))


# ----------------------------------------------------------------------------------------
# product CPU pieces (host trie, BFS fail links, flattener) vs the pinned oracle
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("c", RANDOM["cases"], ids=[c["id"] for c in RANDOM["cases"]])
def test_flat_image_walk_matches_reference_fixtures(c):
    keys, values = _case_values(c)
    A, O = build_pair(keys, values if c["store"] not in ("length", "ints_default") else None, c["store"])
    blob = A.flat_image_bytes()
    for h in c["hays"]:
        hay = bytes.fromhex(h["hay_hex"])
        exp = expected_pairs(h["iter"])
        if c["store"] == "any":
            pass  # values are ids == ordinals == the stored ints
        got, _ = orc.flat_iter(blob, hay)
        assert got == exp
        assert orc.flat_iter_long(blob, hay) == expected_pairs(h["iter_long"])
    state, shift = 0, 0
    for part_hex, exp in zip(c["chunks"]["parts_hex"], c["chunks"]["iter_set"]):
        part = bytes.fromhex(part_hex)
        got, state = orc.flat_iter(blob, part, state=state, index_base=shift)
        assert got == expected_pairs(exp)
        shift += len(part)


def test_flat_image_walk_randomised_vs_oracle():
    rng = random.Random(11)
    for trial in range(40):
        alpha = rng.choice([b"ab", b"abc", b"ACGT", bytes([0x61, 0x80, 0xFF, 0x00]), bytes(range(256))])
        keys = list({bytes(rng.choice(alpha) for _ in range(rng.randint(1, 9))) for _ in range(rng.randint(1, 60))})
        vals = [rng.randint(-2**40, 2**40) for _ in keys]
        A, O = build_pair(keys, vals)
        blob = A.flat_image_bytes()
        for _ in range(8):
            hay = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 200)))
            got, fin = orc.flat_iter(blob, hay)
            assert got == O.iter(hay)
            assert orc.flat_iter_long(blob, hay) == O.iter_long(hay)


def test_flat_image_escape_counts():
    """a state with more than 30 outputs uses the escape count (include/acx_blob.h)"""
    keys = [b"a" * n for n in range(1, 41)]
    A, O = build_pair(keys, list(range(100, 140)))
    blob = A.flat_image_bytes()
    hay = b"a" * 50
    got, _ = orc.flat_iter(blob, hay)
    assert got == O.iter(hay)
    assert len(got) == sum(min(i + 1, 40) for i in range(50))
    assert orc.flat_iter_long(blob, hay) == O.iter_long(hay)


def test_int_truncation_matches_reference_probe():
    A, O = build_pair([b"ab", b"b"], [2**40 + 5, -3])
    got, _ = orc.flat_iter(A.flat_image_bytes(), b"xabx")
    assert got == [(2, 5), (2, -3)] == O.iter(b"xabx")
    assert i32(2**40 + 5) == 5


def test_wide_layout_flat_walk():
    """the 27-bit-state / 2-bit-count entry layout used for automata beyond 2^24 states or a
    4 GiB table, forced here on small automata"""
    import struct
    rng = random.Random(13)
    for trial in range(25):
        alpha = rng.choice([b"ab", b"ACGT", bytes([0x61, 0x80, 0xFF, 0x00]), bytes(range(256))])
        keys = list({bytes(rng.choice(alpha) for _ in range(rng.randint(1, 9))) for _ in range(rng.randint(1, 60))})
        A, O = build_pair(keys, [rng.randint(-2**40, 2**40) for _ in keys])
        blob = A.flat_image_bytes(acx.ACX_FLATTEN_WIDE)
        assert struct.unpack_from("<I", blob, 136)[0] == 27
        for _ in range(6):
            hay = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 200)))
            got, _ = orc.flat_iter(blob, hay)
            assert got == O.iter(hay)
            assert orc.flat_iter_long(blob, hay) == O.iter_long(hay)
    # counts >= 3 take the escape path in this layout
    keys = [b"a" * n for n in range(1, 9)]
    A, O = build_pair(keys)
    blob = A.flat_image_bytes(acx.ACX_FLATTEN_WIDE)
    got, _ = orc.flat_iter(blob, b"a" * 20)
    assert got == O.iter(b"a" * 20)


def test_implicit_top_of_trie_structures():
    """itop (include/acx_blob.h): the ND4 table + entries + level-D row copies let shallow states
    be walked without table rows.  flat_walk.c:flat_iter_itop is the CPU restatement of
    k_walk_itop and cross-checks every step against the explicit table."""
    import struct
    HOST = acx.ACX_FLATTEN_TABLE_HOST                       # the CPU walkers read the table from the blob
    rng = random.Random(17)
    alphabets = [b"ab", b"ACGT", b"ACGTN", bytes([0x61, 0x80, 0xFF, 0x00]), b"0123456789", b"abcdefghijklmnop",
                 b"abcdefghijklmnopqrstuvwxyz ", bytes(range(256)), b"0123456789abcdef", b"xyz"]
    seen = set()
    for trial in range(60):
        alpha = alphabets[trial % len(alphabets)]
        hay_alpha = alpha + (b"#" if len(alpha) < 256 and trial % 3 == 0 else b"")   # bytes outside the key alphabet
        keys = list({bytes(rng.choice(alpha) for _ in range(rng.randint(1, 12))) for _ in range(rng.randint(1, 400))})
        A, O = build_pair(keys, [rng.randint(-2**40, 2**40) for _ in keys])
        blob = A.flat_image_bytes(HOST)
        D = struct.unpack_from("<I", blob, 140)[0]
        cell_bytes = struct.unpack_from("<I", blob, 220)[0]
        sigma = len(set(b"".join(keys)))
        if sigma > 16:                                    # a cell describes at most 16 children: no itop
            assert D == 0
            continue
        if max(map(len, keys)) * max(1, (sigma - 1).bit_length()) >= 5:
            assert D >= 1 and cell_bytes == (4 if sigma <= 4 else 8)
        if D == 0:
            continue
        seen.add((D, cell_bytes))
        for _ in range(6):
            hay = bytes(rng.choice(hay_alpha) for _ in range(rng.randint(0, 300)))
            if rng.random() < 0.5 and keys:
                hay += rng.choice(keys) * 2 + bytes(rng.choice(hay_alpha) for _ in range(5))
            got, fin = orc.flat_iter_itop(blob, hay)
            exp, fin2 = orc.flat_iter(blob, hay)
            assert got == exp == O.iter(hay)
            assert fin == fin2
    assert len({d for d, _ in seen}) >= 3 and {c for _, c in seen} == {4, 8}     # different depths, both cell widths


def test_implicit_top_dna_depth():
    """D = min(what fits LDS: 9 for 2-bit symbols, dense levels + 2): the 2-bit depth field of ND4
    escapes to a probe when the automaton falls below D - 2, which must stay rare"""
    import struct
    from pyahocorasick_amd.workloads import dna_workload
    for n_keys in (300, 3000, 30000):
        keys, reads = dna_workload(n_keys, 200, 150, seed=8)
        A, O = build_pair(keys)
        blob = A.flat_image_bytes()
        D, = struct.unpack_from("<I", blob, 140)
        bits, = struct.unpack_from("<I", blob, 164)
        dense = 0                                     # levels at least 95 % full
        while len({k[:dense + 1] for k in keys if len(k) > dense}) * 100 >= 95 * 4 ** (dense + 1):
            dense += 1
        assert bits == 2 and D == min(9, dense + 2)
        for r in reads[:60]:
            got, _ = orc.flat_iter_itop(blob, r.tobytes())
            assert got == O.iter(r.tobytes())


def test_itop_flags_of_the_flat_image():
    """header word 240 (`itop_flags`): bit 0 promises the itop walk that the depth field of ND4 never escapes
    once D symbols have been seen, i.e. that every level up to D - 2 holds EVERY k-gram over the key
    alphabet (the kernel instantiation without the probe path relies on it); bit 1 that the tflags section
    holds whole packed entries (the state id in the state field)"""
    import struct
    from pyahocorasick_amd.workloads import dna_workload
    rng = random.Random(23)
    cases = [dna_workload(n, 1, 10, seed=9)[0] for n in (40, 300, 3000, 30000)]
    cases.append([bytes(rng.choice(b"ab") for _ in range(rng.randint(1, 14))) for _ in range(3000)])
    cases.append([b"acgtacgtacgt", b"ac", b"g"])                       # sparse: nothing is complete beyond level 1
    both = set()
    for keys in cases:
        keys = list(set(bytes(k) for k in keys))
        A, _ = build_pair(keys)
        blob = A.flat_image_bytes()
        D, = struct.unpack_from("<I", blob, 140)
        if D == 0:
            continue
        flags, = struct.unpack_from("<I", blob, 240)
        n_states, = struct.unpack_from("<I", blob, 24)
        off_tflags, = struct.unpack_from("<Q", blob, 192)
        tf = struct.unpack_from("<%dI" % n_states, blob, off_tflags)
        assert flags & 2 and all((v & 0xFFFFFF) == i for i, v in enumerate(tf))
        sigma = len(set(b"".join(keys)))
        complete = 0                                                   # levels 1..complete hold all sigma^d k-grams
        while len({k[:complete + 1] for k in keys if len(k) > complete}) == sigma ** (complete + 1):
            complete += 1
        assert bool(flags & 1) == (complete + 2 >= D), (D, complete, flags)
        both.add(flags & 1)
    assert both == {0, 1}


# ----------------------------------------------------------------------------------------
# ignore_white_space: fixtures written by the reference (tests/golden/make_ws_golden.py)
# ----------------------------------------------------------------------------------------
WS_CASES = load_json("ref_ws.json")["cases"]


@pytest.mark.parametrize("c", WS_CASES, ids=[c["id"] for c in WS_CASES])
def test_oracle_ignore_white_space_matches_reference_fixtures(c):
    keys = [bytes.fromhex(k) for k in c["keys_hex"]]
    O = _oracle_from(keys, list(range(len(keys))))
    for h in c["hays"]:
        hay = bytes.fromhex(h["hay_hex"])
        assert O.iter(hay, ignore_ws=True) == expected_pairs(h["iter_ws"])
        assert O.iter(hay) == expected_pairs(h["iter"])
        if "slice" in h:
            s = h["slice"]
            assert O.iter(hay, s["start"], s["end"], ignore_ws=True) == expected_pairs(s["iter_ws"])
    # the stream: state and shift carried across set(chunk), src/AutomatonSearchIter.c:303-368
    state, shift = 0, 0
    for chunk_hex, exp in zip(c["stream"]["chunks_hex"], c["stream"]["iter_ws_set"]):
        chunk = bytes.fromhex(chunk_hex)
        e, v, state = O.iter_arrays(chunk, 0, len(chunk), ignore_ws=True, state=state, shift=shift)
        assert list(zip(e.tolist(), v.tolist())) == expected_pairs(exp)
        shift += len(chunk)
