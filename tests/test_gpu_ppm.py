"""Position-parallel scan kernels (csrc/acx_ppm_kernels.hip) through the C-ABI on a real MI355X:
three-way parity — position-parallel (default) == serial walk kernels (variant bit 23) == oracle —
on every dictionary shape the image builder distinguishes (2/4/8-bit symbols, power-of-two and
mixed-radix codes, with and without a filter level below the cells), fixed-stride and offset
batches, bytes outside the key alphabet, empty and ragged haystacks, index_base, many outputs per
position, record-pool overflow and regrowth."""
import struct

import numpy as np
import pytest

import pyahocorasick_amd as acx
from pyahocorasick_amd.device import DeviceBuffer, Image, Scanner
from helpers import build_pair

pytestmark = pytest.mark.gpu

SERIAL = 1 << 23          # serial walk kernels instead of the position-parallel ones
GENERAL = (1 << 24) | (1 << 28)   # k_ppm_scan (any batch; bit 28: even where the serial walks would be chosen) instead of k_ppm_stream
STREAM_ANY = (1 << 19) | (1 << 28)  # k_ppm_stream (every alphabet) where k_ppm_stream4 (four letters, fixed stride) would take the scan


def _ppm_fields(blob):
    off = struct.unpack_from("<Q", blob, 248)[0]
    assert off, "image carries no ppm section"
    magic, K, sb, pow2, C, F, g_global, F2 = struct.unpack_from("<8I", blob, off)
    return dict(K=K, sym_bits=sb, pow2=pow2, C=C, F=F, g_global=g_global, F2=F2)


def _three_way(A, O, data, off, stride=None, index_base=None, want_final=True):
    """scan `data` cut by `off` with both kernel families (and, for equal lengths, through the
    fixed-stride entry too); all must equal the oracle"""
    n = len(off) - 1
    mo, oe, ov = O.batch(bytes(data), off, 0)
    if index_base is not None:
        oe = oe + np.repeat(index_base, np.diff(mo)).astype(np.int32)
    img = Image.from_automaton(A)
    d_hay = DeviceBuffer.from_numpy(np.frombuffer(bytes(data), dtype=np.uint8), pad=64)
    d_off = DeviceBuffer.from_numpy(np.asarray(off, dtype=np.int64))
    d_base = DeviceBuffer.from_numpy(np.asarray(index_base, dtype=np.int32)) if index_base is not None else None
    fins = []
    shortest = int(np.diff(np.asarray(off)).min()) if n else 0
    # offsets entry twice: with the caller's lower bound on the haystack lengths (>= 8: the stream kernel) and without
    entries = [dict(dev_off=d_off), dict(dev_off=d_off, min_hay_len=shortest)] + ([dict(stride=stride)] if stride else [])
    for variant in (1 << 28, STREAM_ANY, GENERAL, SERIAL):
        for kw in entries:
            if variant == STREAM_ANY and "stride" not in kw:
                continue                                  # (k_ppm_stream4 takes fixed-stride batches only: nothing to tell apart)
            sc = Scanner(img)
            sc.scan(d_hay, len(data), n, dev_index_base=d_base, want_final_state=want_final, variant=variant, **kw)
            moff, e, v, fin = sc.fetch()
            assert np.array_equal(moff, mo), (variant, list(kw.keys()), int(np.argmax(moff != mo)) if len(moff) == len(mo) else -1)
            assert np.array_equal(e, oe) and np.array_equal(v, ov), (variant, kw.keys())
            fins.append(fin)
    if want_final:
        for f in fins[1:]:
            assert np.array_equal(f, fins[0])
    return len(oe)


def test_he_her_hers_she():
    A, O = build_pair([b"he", b"her", b"hers", b"she"])
    hay = b"_sherhershe_ ushers he"
    assert _three_way(A, O, hay, [0, len(hay)]) > 0
    assert [(i, v) for i, v in A.iter(b"_sherhershe_")] == O.iter(b"_sherhershe_")


ALPHABETS = [
    ("dna", b"ACGT", b"ACGT"), ("dna+N", b"ACGT", b"ACGTN"), ("binary", b"ab", b"abz"), ("unary", b"a", b"ab"),
    ("ternary", b"abc", b"abc"), ("acgtn5", b"ACGTN", b"ACGTNX"), ("digits", b"0123456789", b"0123456789 -"),
    ("hex16", b"0123456789abcdef", b"0123456789abcdefg"), ("text27", bytes(range(97, 123)) + b" ", bytes(range(97, 123)) + b" ."),
    ("bytes256", bytes(range(256)), bytes(range(256))), ("high4", bytes([0x61, 0x80, 0xFF, 0x00]), bytes([0x61, 0x80, 0xFF, 0x00, 0x7F])),
]


@pytest.mark.parametrize("name,alpha,hay_alpha", ALPHABETS, ids=[a[0] for a in ALPHABETS])
def test_alphabets_three_way(name, alpha, hay_alpha):
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    a = np.frombuffer(alpha, dtype=np.uint8)
    maxlen = 40 if len(alpha) <= 2 else 14
    keys = list({bytes(rng.choice(a, size=int(k)).tobytes()) for k in rng.integers(1, maxlen, size=3000)})
    A, O = build_pair(keys)
    f = _ppm_fields(A.flat_image_bytes())
    assert f["K"] == len(alpha)
    ha = np.frombuffer(hay_alpha, dtype=np.uint8)
    n, L = 500, 173
    reads = np.ascontiguousarray(ha[rng.integers(0, len(ha), size=(n, L))])
    for i in range(0, n, 3):
        k = np.frombuffer(keys[int(rng.integers(0, len(keys)))], dtype=np.uint8)
        o = int(rng.integers(0, L - len(k)))
        reads[i, o:o + len(k)] = k
    off = np.arange(n + 1, dtype=np.int64) * L
    assert _three_way(A, O, reads.tobytes(), off, stride=L) > 0


@pytest.mark.parametrize("stride", [1, 2, 3, 5, 8, 9, 64, 150, 255, 256, 257, 1000, 1023, 1024, 2047, 2048, 2049, 4099])
def test_fixed_stride_shapes(stride):
    rng = np.random.default_rng(stride)
    a = np.frombuffer(b"ACGT", dtype=np.uint8)
    keys = list({bytes(rng.choice(a, size=int(k)).tobytes()) for k in rng.integers(1, 12, size=400)})
    A, O = build_pair(keys)
    n = max(3, 20000 // stride)
    reads = np.ascontiguousarray(a[rng.integers(0, 4, size=(n, stride))])
    off = np.arange(n + 1, dtype=np.int64) * stride
    base = rng.integers(0, 1000, size=n).astype(np.int32)
    _three_way(A, O, reads.tobytes(), off, stride=stride, index_base=base)


@pytest.mark.parametrize("stride", [9, 1999, 2047, 2048])
def test_runs_of_tiles_per_wave_around_the_division_thresholds(stride):
    """> 8.4 M positions: every wave of k_ppm_stream takes a run of tiles, so the offset of a tile's first byte in its
    haystack is carried from tile to tile; strides at both ends of the 24-bit-multiply division and just past it"""
    rng = np.random.default_rng(stride)
    a = np.frombuffer(b"ACGT", dtype=np.uint8)
    keys = list({bytes(rng.choice(a, size=int(k)).tobytes()) for k in rng.integers(6, 14, size=3000)})
    A, O = build_pair(keys)
    n = 9_500_000 // stride
    reads = np.ascontiguousarray(a[rng.integers(0, 4, size=(n, stride))])
    off = np.arange(n + 1, dtype=np.int64) * stride
    assert _three_way(A, O, reads.tobytes(), off, stride=stride) > 0


def test_second_level_filter_text_one_long_haystack():
    """text-like alphabet (27 symbols of 8 bits: the first filter is capped at the 4 symbols of a 32-bit window) gets
    the second-level filter in L2; one haystack of 10 MB (every wave takes a run of tiles), as config 3 is scanned.
    With and without it (variant bit 20), against the oracle, record for record."""
    rng = np.random.default_rng(41)
    alpha = np.frombuffer(bytes(range(97, 123)) + b" ", dtype=np.uint8)
    words = [bytes(rng.choice(alpha[:26], size=int(k)).tobytes()) for k in rng.integers(2, 9, size=4000)]
    keys = list({b" ".join(words[int(j)] for j in rng.integers(0, len(words), size=int(m)))[:int(t)]
                 for m, t in zip(rng.integers(1, 5, size=6000), rng.integers(5, 30, size=6000))})
    keys = [k for k in keys if k]
    A, O = build_pair(keys)
    f = _ppm_fields(A.flat_image_bytes())
    assert f["sym_bits"] == 8 and not f["pow2"] and f["F"] == 4 and f["F2"] > f["F"], f
    text = b" ".join(words[int(j)] for j in rng.integers(0, len(words), size=1_700_000))
    text = text[:10_000_000] + b" " + keys[0] + b"." + keys[1]          # a byte of no key near the end
    off = np.array([0, len(text)], dtype=np.int64)
    mo, oe, ov = O.batch_records(np.frombuffer(text, dtype=np.uint8), off, 0)
    assert mo[-1] > 1000
    img = Image.from_automaton(A)
    d_hay = DeviceBuffer.from_numpy(np.frombuffer(text, dtype=np.uint8), pad=64)
    d_off = DeviceBuffer.from_numpy(off)
    for variant in (0, 1 << 20):
        sc = Scanner(img)
        sc.scan(d_hay, len(text), 1, dev_off=d_off, min_hay_len=len(text), variant=variant)
        moff, e, v, _ = sc.fetch()
        assert np.array_equal(moff, mo) and np.array_equal(e, oe) and np.array_equal(v, ov), variant


def test_ragged_offsets_with_empty_haystacks_and_long_ones():
    rng = np.random.default_rng(5)
    a = np.frombuffer(b"ab", dtype=np.uint8)
    keys = [bytes(rng.choice(a, size=int(k)).tobytes()) for k in rng.integers(1, 41, size=300)]
    keys = list(dict.fromkeys(keys + [b"a" * 40, b"ab" * 20, b"b" * 33]))
    A, O = build_pair(keys)
    lens = [0, 1, 39, 40, 41, 255, 256, 257, 319, 512, 640, 5000, 0, 0, 100_001, 7, 33_333, 0]
    hays = [bytes(rng.choice(a, size=k).tobytes()) for k in lens]
    hays[11] = (b"ab" * 20 + b"a" * 40) * 80
    data = b"".join(hays)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    base = (np.arange(len(lens)) * 1000).astype(np.int32)
    _three_way(A, O, data, off, index_base=base)


def test_many_outputs_per_position_and_wide_counts():
    """a*1 .. a*40: up to 40 records per position, tile totals beyond 8 bits; keys of 300 bytes:
    32-bit per-position counters and a halo longer than a tile"""
    keys = [b"a" * k for k in range(1, 41)]
    A, O = build_pair(keys, list(range(100, 140)))
    hay = (b"a" * 700 + b"b") * 5
    _three_way(A, O, hay, [0, len(hay)])
    _three_way(A, O, hay[:3500], np.arange(8) * 500, stride=500)
    keys = [b"ab" * 150, b"b" * 270, b"abab", b"a"]
    A, O = build_pair(keys)
    hay = b"ab" * 400 + b"b" * 600 + b"ab" * 200
    _three_way(A, O, hay, [0, len(hay)])


def test_unaligned_device_pointer_and_tail():
    """the staging loads whole dwords: a haystack buffer that starts on an odd address and ends
    flush with the allocation must not read outside [dev_hay, dev_hay + capacity)"""
    rng = np.random.default_rng(9)
    a = np.frombuffer(b"ACGT", dtype=np.uint8)
    keys = list({bytes(rng.choice(a, size=int(k)).tobytes()) for k in rng.integers(2, 10, size=200)})
    A, O = build_pair(keys)
    img = Image.from_automaton(A)
    n, L = 37, 151
    reads = np.ascontiguousarray(a[rng.integers(0, 4, size=(n, L))])
    off = np.arange(n + 1, dtype=np.int64) * L
    mo, oe, ov = O.batch(reads.tobytes(), off, 0)
    for lead in (1, 2, 3):
        buf = np.concatenate([np.zeros(lead, dtype=np.uint8), reads.reshape(-1)])
        d = DeviceBuffer.from_numpy(buf)
        sc = Scanner(img)
        sc.scan(d.ptr.value + lead, n * L, n, stride=L)
        moff, e, v, _ = sc.fetch()
        assert np.array_equal(moff, mo) and np.array_equal(e, oe) and np.array_equal(v, ov)


def test_record_pool_overflow_regrows():
    """first scan of a result object sizes the pool from the haystack bytes / 8; a match-dense batch
    overflows it and must be scanned again with a larger pool, transparently"""
    keys = [b"a" * k for k in range(1, 33)]
    A, O = build_pair(keys)
    hay = b"a" * 20000
    off = np.arange(0, 20001, 200, dtype=np.int64)
    _three_way(A, O, hay, off, stride=200)


def test_dna_config2_shape_reduced():
    from helpers import dna_workload
    keys, reads = dna_workload(20000, 20000, 150, seed=3)
    A, O = build_pair(list(keys))
    f = _ppm_fields(A.flat_image_bytes())
    assert f["sym_bits"] == 2 and f["pow2"] == 1 and f["F"] == f["C"] + 1
    off = np.arange(len(reads) + 1, dtype=np.int64) * reads.shape[1]
    assert _three_way(A, O, reads.tobytes(), off, stride=reads.shape[1]) > 10000
