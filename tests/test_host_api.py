"""CPU suite: host logic of the Automaton mirror — trie API, kinds, versions, argument and
error conventions of the reference (tests modelled on reference tests/test_unit.py)."""
import pytest

import pyahocorasick_amd as acx
from pyahocorasick_amd.automaton import _parse_start_end


def test_constructor_errors():                      # reference tests/test_unit.py:35-43
    with pytest.raises(ValueError):
        acx.Automaton(1234)
    with pytest.raises(ValueError):
        acx.Automaton(acx.STORE_ANY, 5)


def test_constants_match_reference_module():        # src/pyahocorasick.c:113-134, src/Automaton.h:16-41
    assert (acx.EMPTY, acx.TRIE, acx.AHOCORASICK) == (0, 1, 2)
    assert (acx.STORE_INTS, acx.STORE_LENGTH, acx.STORE_ANY) == (10, 20, 30)
    assert (acx.KEY_STRING, acx.KEY_SEQUENCE) == (100, 200)
    assert acx.unicode == 0


def test_kind_transitions_and_len():
    A = acx.Automaton()
    assert A.kind == acx.EMPTY and len(A) == 0
    assert A.add_word(b"he", "x") is True
    assert A.add_word(b"he", "y") is False           # existing key: value replaced, not new
    assert A.kind == acx.TRIE and len(A) == 1
    assert A.get(b"he") == "y"
    assert A.make_automaton() is None
    assert A.kind == acx.AHOCORASICK
    assert A.make_automaton() is False               # src/Automaton.c:574-575
    A.add_word(b"she", "z")
    assert A.kind == acx.TRIE                        # src/trie.c:60
    A.clear()
    assert A.kind == acx.EMPTY and len(A) == 0


def test_add_word_value_rules():
    A = acx.Automaton()                              # STORE_ANY needs a value (src/Automaton.c:218-222)
    with pytest.raises(ValueError):
        A.add_word(b"x")
    with pytest.raises(TypeError):
        A.add_word("text", 1)                        # bytes build wants bytes
    B = acx.Automaton(acx.STORE_INTS)
    B.add_word(b"a")
    B.add_word(b"b")
    B.add_word(b"c", 77)
    assert [B.get(k) for k in (b"a", b"b", b"c")] == [1, 2, 77]   # default = count + 1
    with pytest.raises(TypeError):
        B.add_word(b"d", "not an int")
    C = acx.Automaton(acx.STORE_LENGTH)
    C.add_word(b"hello")
    assert C.get(b"hello") == 5
    assert A.add_word(b"", 1) is False               # empty key ignored (src/Automaton.c:257)


def test_exists_get_longest_prefix_contains():
    A = acx.Automaton(acx.STORE_INTS)
    for i, k in enumerate([b"he", b"her", b"hers", b"she"]):
        A.add_word(k, i)
    assert A.exists(b"her") and b"she" in A and not A.exists(b"h") and b"x" not in A
    assert A.get(b"nope", 42) == 42
    with pytest.raises(KeyError):
        A.get(b"nope")
    assert A.longest_prefix(b"herxyz") == 3 and A.longest_prefix(b"zzz") == 0


def test_remove_word_and_pop():
    A = acx.Automaton()
    for k in (b"he", b"her", b"hers", b"she"):
        A.add_word(k, k)
    assert A.remove_word(b"her") is True and not A.exists(b"her") and A.exists(b"hers") and A.exists(b"he")
    assert A.remove_word(b"her") is False
    assert A.pop(b"hers") == b"hers" and not A.exists(b"hers") and A.exists(b"he")
    with pytest.raises(KeyError):
        A.pop(b"hers")
    assert len(A) == 2
    A.add_word(b"hers", b"again")
    assert A.get(b"hers") == b"again"


def test_version_bumps_like_reference():
    A = acx.Automaton(acx.STORE_INTS)
    v0 = A._version
    A.add_word(b"a", 1); v1 = A._version
    A.add_word(b"a", 2); v2 = A._version             # existing key: no bump (src/Automaton.c:283-284)
    A.make_automaton(); v3 = A._version
    A.remove_word(b"a"); v4 = A._version
    A.clear(); v5 = A._version
    assert v1 == v0 + 1 and v2 == v1 and v3 == v2 + 1 and v4 == v3 + 1 and v5 == v4 + 1


def test_search_before_make_automaton():
    A = acx.Automaton()
    with pytest.raises(AttributeError):
        A.iter(b"x")
    A.add_word(b"word", None)
    with pytest.raises(AttributeError):
        A.iter(b"x")
    with pytest.raises(AttributeError):
        A.iter_long(b"x")
    assert A.find_all(b"x", b"any arg") is None      # silently None (src/Automaton.c:666-667)
    with pytest.raises(AttributeError):
        A.iter_batch([b"x"])


def test_argument_types():
    A = acx.Automaton(acx.STORE_INTS)
    A.add_word(b"a", 1)
    A.make_automaton()
    with pytest.raises(TypeError, match="bytes required"):
        A.iter(None)
    with pytest.raises(TypeError, match="bytes required"):
        A.iter("text")
    with pytest.raises(TypeError, match="callable"):
        A.find_all(b"_sherhershe_", None)
    with pytest.raises(TypeError):
        A.iter(b"x", ignore_white_space2=True)


def test_parse_start_end_quirks():                   # src/utils.c:293-359
    assert _parse_start_end((), 0, 12) == (0, 12)
    assert _parse_start_end((0,), 0, 12) == (0, 12)
    assert _parse_start_end((-3, 4), 0, 12) == (9, 4)
    assert _parse_start_end((0, -1), 0, 12) == (0, 10)     # len - 1 + end (sic)
    with pytest.raises(IndexError, match="end index not in range 0..12"):
        _parse_start_end((0, 17), 0, 12)
    with pytest.raises(IndexError, match="start index not in range 0..12"):
        _parse_start_end((-13, 3), 0, 12)
    with pytest.raises(IndexError):
        _parse_start_end((12,), 0, 12)


def test_flat_image_header_and_validation():
    import ctypes as C
    from pyahocorasick_amd._lib import lib, check
    A = acx.Automaton(acx.STORE_INTS)
    for i, k in enumerate([b"ACGT", b"CG", b"T"]):
        A.add_word(k, i)
    A.make_automaton()
    blob = A.flat_image_bytes()
    assert blob[:8] == b"ACXBLOB1"
    buf = C.create_string_buffer(blob, len(blob))
    check(lib().acx_blob_validate(buf, len(blob)))
    bad = bytearray(blob)
    bad[1000] ^= 0xFF
    buf2 = C.create_string_buffer(bytes(bad), len(bad))
    assert lib().acx_blob_validate(buf2, len(bad)) != 0
    assert lib().acx_blob_validate(buf, len(blob) - 256) != 0
    # DNA keys -> 4 used bytes + the shared "other" class
    import struct
    n_states, n_classes = struct.unpack_from("<II", blob, 24)
    assert n_classes == 5 and n_states == 8


def test_save_image_round_trip(tmp_path):
    """the flat image is relocatable: written to a file and read back it validates (checksum,
    structure) and is byte-identical; STORE_ANY is refused (object values do not travel)"""
    import ctypes as C
    from pyahocorasick_amd._lib import lib, check
    A = acx.Automaton(acx.STORE_INTS)
    for i, k in enumerate([b"he", b"her", b"hers", b"she", b"\xff\x80"]):
        A.add_word(k, 100 + i)
    A.make_automaton()
    p = tmp_path / "she.acx"
    A.save_image(str(p))
    data = p.read_bytes()
    assert data == A.flat_image_bytes()
    buf = C.create_string_buffer(data, len(data))
    check(lib().acx_blob_validate(buf, len(data)))
    B = acx.Automaton()
    B.add_word(b"x", object())
    B.make_automaton()
    with pytest.raises(ValueError):
        B.save_image(str(tmp_path / "no.acx"))


def test_add_words_equals_add_word_in_a_loop():
    """acx_trie_add_words / Automaton.add_words (an extension: one call for many keys) == add_word per pair, in order"""
    import random
    import pyahocorasick_amd as acx
    rng = random.Random(4)
    keys = [bytes(rng.choice(b"abc") for _ in range(rng.randint(0, 7))) for _ in range(400)]     # duplicates and empty keys included
    for store, vals in ((acx.STORE_INTS, [rng.randint(-2**40, 2**40) for _ in keys]), (acx.STORE_INTS, None), (acx.STORE_LENGTH, None)):
        A, B = acx.Automaton(store), acx.Automaton(store)
        n_new = 0
        for i, k in enumerate(keys):
            n_new += bool(A.add_word(k, vals[i]) if vals is not None else A.add_word(k))
        assert B.add_words(keys, vals) == n_new
        assert len(A) == len(B)
        assert sorted(A.items()) == sorted(B.items())
        A.make_automaton(); B.make_automaton()
        assert A.flat_image_bytes() == B.flat_image_bytes()
    C_ = acx.Automaton()                                               # STORE_ANY: objects
    assert C_.add_words([b"x", b"xy", b"x"], ["a", "b", "c"]) == 2 and C_.get(b"x") == "c"


def test_key_sequence_automata_come_from_the_extension():
    """pyahocorasick_amd.Automaton(store, KEY_SEQUENCE) is the extension's Automaton (src/utils.c:238-289: keys are tuples
    of integers): one implementation of that flavour, and the ctypes mirror does not refuse it"""
    A = acx.Automaton(acx.STORE_INTS, acx.KEY_SEQUENCE)
    assert type(A).__module__ == "ahocorasick"
    assert A.add_word((1, 2, 3), 7) and A.add_word((2, 3), 9) and not A.add_word((1, 2, 3), 8)
    assert A.get((1, 2, 3)) == 8 and A.exists((2, 3)) and not A.exists((3,)) and len(A) == 2
    assert A.longest_prefix((1, 2, 9)) == 2
    with pytest.raises(ValueError):
        A.add_word((1, -2), 1)
    B = acx.Automaton(acx.STORE_INTS, acx.KEY_STRING)
    assert type(B) is acx.Automaton
