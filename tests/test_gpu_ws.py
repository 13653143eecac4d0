"""GPU suite (-m gpu): ACX_SCAN_SKIP_WS — iter(..., ignore_white_space=True) on the device.

The reference steps over every white-space letter without touching its state and reports indices of the original
string (src/AutomatonSearchIter.c:269-274).  Here the batch is compacted on the device, scanned by the kernels that scan
any other batch, and the end indices are mapped back; the oracle (orc_iter with ignore_ws) is the checker.
Nothing here reads /root/reference.
"""
import numpy as np
import pytest

from pyahocorasick_amd._lib import ACX_SCAN_LONG
from pyahocorasick_amd.device import DeviceBuffer, Image, Scanner
from helpers import build_pair
from test_dropin_module import ahocorasick  # noqa: F401  (the fixture: the drop-in extension, freshly built)

pytestmark = pytest.mark.gpu

WS = np.frombuffer(b" \t\n\v\f\r", dtype=np.uint8)


def _sprinkle(rng, arr, frac):
    """white space over a fraction of the bytes (runs included)"""
    out = arr.copy()
    hit = rng.random(arr.size) < frac
    out[hit] = WS[rng.integers(0, len(WS), size=int(hit.sum()))]
    return out


def _pairs(moff, e, v, i):
    return list(zip(e[moff[i]:moff[i + 1]].tolist(), v[moff[i]:moff[i + 1]].tolist()))


@pytest.mark.parametrize("alpha_b,kmax", [(b"ACGT", 12), (b"abcdefghijklmnopqrstuvwxyz", 8), (bytes(range(256)), 5)],
                         ids=["dna", "text", "bytes"])
def test_batch_with_white_space_equals_the_oracle(alpha_b, kmax):
    """offsets and fixed-stride batches, every kernel family, ragged and empty haystacks, haystacks of white space only,
    an unaligned buffer; keys that themselves hold white space can never match (the reference never feeds it)."""
    rng = np.random.default_rng(5)
    alpha = np.frombuffer(alpha_b, dtype=np.uint8)
    keys = list({bytes(alpha[rng.integers(0, len(alpha), size=int(k))]) for k in rng.integers(2, kmax + 1, size=400)})
    A, O = build_pair(keys)
    img = Image.from_automaton(A)
    # fixed stride
    n, L = 900, 200
    hay = alpha[rng.integers(0, len(alpha), size=(n, L))]
    for i in range(0, n, 2):                                             # keys broken up by white space
        k = np.frombuffer(keys[int(rng.integers(0, len(keys)))], dtype=np.uint8)
        o = int(rng.integers(0, L - 3 * len(k)))
        spaced = np.empty(2 * len(k), dtype=np.uint8)
        spaced[0::2] = k
        spaced[1::2] = WS[rng.integers(0, len(WS), size=len(k))]
        hay[i, o:o + spaced.size] = spaced
    hay = _sprinkle(rng, hay.reshape(-1), 0.15).reshape(n, L)
    hay[7] = 0x20                                                        # white space only
    hay[8, :L - 1] = 0x0a
    want = [O.iter(hay[i].tobytes(), ignore_ws=True) for i in range(n)]
    assert sum(map(len, want)) > n // 4                                  # the test is not vacuous
    assert want != [O.iter(hay[i].tobytes()) for i in range(n)]
    d_hay = DeviceBuffer.from_numpy(hay.reshape(-1), pad=64)
    for variant in (0, (1 << 24) | (1 << 28), 1 << 23):
        sc = Scanner(img)
        sc.scan(d_hay, hay.size, n, stride=L, variant=variant, skip_white_space=True)
        moff, e, v, _ = sc.fetch()
        assert [_pairs(moff, e, v, i) for i in range(n)] == want, variant
    # offsets: ragged, empty ones, a base per haystack, an unaligned buffer
    lens = rng.integers(0, 400, size=700)
    lens[::50] = 0
    off = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    flat = _sprinkle(rng, alpha[rng.integers(0, len(alpha), size=int(off[-1]))], 0.2)
    base = rng.integers(0, 1000, size=len(lens)).astype(np.int32)
    want = [[(e + int(base[i]), v) for e, v in O.iter(flat[off[i]:off[i + 1]].tobytes(), ignore_ws=True)] for i in range(len(lens))]
    for shift in (0, 3):
        padded = np.concatenate([np.zeros(shift, dtype=np.uint8), flat])
        d_flat = DeviceBuffer.from_numpy(padded, pad=64)
        d_off = DeviceBuffer.from_numpy(off)
        d_base = DeviceBuffer.from_numpy(base)
        for variant in (0, (1 << 24) | (1 << 28), 1 << 23):
            sc = Scanner(img)
            sc.scan(d_flat.ptr.value + shift, flat.size, len(lens), dev_off=d_off, dev_index_base=d_base, variant=variant,
                    skip_white_space=True)
            moff, e, v, _ = sc.fetch()
            assert [_pairs(moff, e, v, i) for i in range(len(lens))] == want, (shift, variant)


def test_streams_with_white_space_and_contexts():
    """dev_skip counts bytes of the batch as given, white space included: a stream cut into chunks, every chunk scanned
    behind the bytes that hold its last longest_word - 1 letters, gives what one scan of the stream gives."""
    rng = np.random.default_rng(23)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    keys = list({bytes(alpha[rng.integers(0, 4, size=int(k))]) for k in rng.integers(3, 14, size=300)})
    A, O = build_pair(keys)
    img = Image.from_automaton(A)
    halo = max(len(k) for k in keys) - 1
    n, chunk = 500, 128
    streams = _sprinkle(rng, alpha[rng.integers(0, 4, size=n * 3 * chunk)], 0.25).reshape(n, 3 * chunk)
    want = [O.iter(streams[i].tobytes(), ignore_ws=True) for i in range(n)]
    for asynchronous in (False, True):
        got = [[] for _ in range(n)]
        for c in range(3):
            # the context: back from the cut until `halo` letters are inside (or the stream's start)
            parts, skips = [], np.zeros(n, dtype=np.int32)
            for i in range(n):
                lo = c * chunk
                letters = 0
                while lo > 0 and letters < halo:
                    lo -= 1
                    letters += int(streams[i, lo] not in WS)
                skips[i] = c * chunk - lo
                parts.append(streams[i, lo:(c + 1) * chunk])
            off = np.zeros(n + 1, dtype=np.int64)
            np.cumsum([len(p) for p in parts], out=off[1:])
            flat = np.concatenate(parts)
            d = [DeviceBuffer.from_numpy(flat, pad=64), DeviceBuffer.from_numpy(off), DeviceBuffer.from_numpy(skips),
                 DeviceBuffer.from_numpy(np.full(n, c * chunk, dtype=np.int32))]
            sc = Scanner(img)
            sc.scan(d[0], flat.size, n, dev_off=d[1], dev_skip=d[2], dev_index_base=d[3], min_hay_len=int(np.diff(off).min()),
                    skip_white_space=True, asynchronous=asynchronous)
            moff, e, v, _ = sc.fetch()
            for i in range(n):
                got[i] += _pairs(moff, e, v, i)
        assert got == want, asynchronous


def test_iter_long_with_white_space_equals_iter_long_of_the_letters():
    """the flag is a property of the batch, not of the mode: ACX_SCAN_LONG over a batch with white space reports what
    it reports over the letters alone, at the positions the letters came from"""
    rng = np.random.default_rng(31)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    keys = list({bytes(alpha[rng.integers(0, 4, size=int(k))]) for k in rng.integers(3, 10, size=200)})
    A, O = build_pair(keys)
    img = Image.from_automaton(A)
    n, L = 300, 150
    hay = _sprinkle(rng, alpha[rng.integers(0, 4, size=n * L)], 0.2).reshape(n, L)
    want = []
    for i in range(n):
        keep = np.flatnonzero(~np.isin(hay[i], WS))
        want.append([(int(keep[e]), v) for e, v in O.iter_long(hay[i][keep].tobytes())])
    sc = Scanner(img)
    sc.scan(DeviceBuffer.from_numpy(hay.reshape(-1), pad=64), hay.size, n, stride=L, mode=ACX_SCAN_LONG, skip_white_space=True)
    moff, e, v, _ = sc.fetch()
    assert [_pairs(moff, e, v, i) for i in range(n)] == want


def test_iterators_of_both_host_sides_ignore_white_space(ahocorasick):
    """iter(string, start, end, ignore_white_space=True) and set() across chunks, drop-in and ctypes mirror"""
    import pyahocorasick_amd as acx
    words = [b"he", b"her", b"hers", b"she", b"us"]
    text = b"s h\te\nr s  he\rrs u s\x0b\x0cshe"
    for mod in (ahocorasick, acx):
        A = mod.Automaton()
        for i, w in enumerate(words):
            A.add_word(w, i)
        A.make_automaton()
        _, O = build_pair(words, list(range(len(words))), "ints")
        assert list(A.iter(text, ignore_white_space=True)) == O.iter(text, ignore_ws=True)
        assert list(A.iter(text, 3, 17, ignore_white_space=True)) == O.iter(text, 3, 17, ignore_ws=True)
        assert list(A.iter(text)) == O.iter(text)
        it = A.iter(b"", ignore_white_space=True)
        got = []
        for cut in (0, 5, 6, 13, 20):
            nxt = {0: 5, 5: 6, 6: 13, 13: 20, 20: len(text)}[cut]
            it.set(text[cut:nxt])
            got += list(it)
        assert got == O.iter(text, ignore_ws=True)


def test_reference_white_space_fixtures_on_both_host_sides(ahocorasick):
    """tests/golden/ref_ws.json (written by the reference, tests/golden/make_ws_golden.py): iter(..., ignore_white_space=True)
    on whole strings and slices, and a stream continued with iter().set(chunk) — drop-in extension and ctypes mirror"""
    import pyahocorasick_amd as acx
    from helpers import expected_pairs, load_json
    for c in load_json("ref_ws.json")["cases"]:
        keys = [bytes.fromhex(k) for k in c["keys_hex"]]
        for mod in (ahocorasick, acx):
            A = mod.Automaton()
            for i, k in enumerate(keys):
                A.add_word(k, i)
            A.make_automaton()
            for h in c["hays"]:
                hay = bytes.fromhex(h["hay_hex"])
                assert list(A.iter(hay, ignore_white_space=True)) == expected_pairs(h["iter_ws"]), c["id"]
                assert list(A.iter(hay)) == expected_pairs(h["iter"]), c["id"]
                if "slice" in h:
                    s = h["slice"]
                    assert list(A.iter(hay, s["start"], s["end"], ignore_white_space=True)) == expected_pairs(s["iter_ws"]), c["id"]
            it = A.iter(b"", ignore_white_space=True)
            for chunk_hex, exp in zip(c["stream"]["chunks_hex"], c["stream"]["iter_ws_set"]):
                it.set(bytes.fromhex(chunk_hex))
                assert list(it) == expected_pairs(exp), c["id"]
