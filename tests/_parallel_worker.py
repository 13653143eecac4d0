"""worker for tests/test_parallel_cpu.py — run under torch.distributed.run, backend gloo.
Exercises the N>1 plumbing on CPU: blob broadcast, sharding, rank-order concatenation.
The scan itself has no CPU path; each rank stands in for its GPU with the test-only flat
image walker (oracle/flat_walk.c), which is fine here: this is test infrastructure."""
import ctypes as C
import hashlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyahocorasick_amd as acx  # noqa: E402
from pyahocorasick_amd import _lib  # noqa: E402
from pyahocorasick_amd.parallel import broadcast_blob, gather_csr, halo_shard, shard_range, shard_range_by_bytes  # noqa: E402
from pyahocorasick_amd.workloads import dna_workload  # noqa: E402
from oracle import orc  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    keys, reads = dna_workload(500, 301, 60, seed=4, klo=3, khi=9)
    blob = None
    if rank == 0:
        A = acx.Automaton(acx.STORE_INTS)
        for i, k in enumerate(keys):
            A.add_word(k, i)
        A.make_automaton()
        blob = A.flat_image_bytes()
    t = broadcast_blob(blob, src=0)
    got = t.numpy().tobytes()
    # every rank holds a valid, identical image
    buf = C.create_string_buffer(got, len(got))
    _lib.check(_lib.lib().acx_blob_validate(buf, len(got)))
    digest = hashlib.sha256(got).hexdigest()
    digests = [None] * world
    dist.all_gather_object(digests, digest)
    assert len(set(digests)) == 1

    # ragged batch: drop a varying tail from each read
    lens = np.array([60 - (i % 7) for i in range(len(reads))], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(lens)])
    data = b"".join(reads[i, :lens[i]].tobytes() for i in range(len(reads)))
    for splitter in (lambda: shard_range(len(reads), rank, world), lambda: shard_range_by_bytes(off, rank, world)):
        lo, hi = splitter()
        spans = [None] * world
        dist.all_gather_object(spans, (lo, hi))
        assert spans[0][0] == 0 and spans[-1][1] == len(reads)
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))      # contiguous, disjoint, complete
        loc_off, loc_e, loc_v = [0], [], []
        for h in range(lo, hi):
            pairs, _ = orc.flat_iter(got, data[off[h]:off[h + 1]])
            loc_e += [p[0] for p in pairs]
            loc_v += [p[1] for p in pairs]
            loc_off.append(len(loc_e))
        res = gather_csr(np.array(loc_off, dtype=np.int64), np.array(loc_e, dtype=np.int32), np.array(loc_v, dtype=np.int32))
        if rank == 0:
            O = orc.Oracle()
            for i, k in enumerate(keys):
                O.add_word(k, i)
            O.make_automaton()
            mo, e, v = O.batch(data, off, 0)
            assert np.array_equal(res[0], mo) and np.array_equal(res[1], e) and np.array_equal(res[2], v)
        else:
            assert res is None
    # iter_long shards (bench.py --mode iter_long): WHOLE haystacks per rank, by count and by bytes, no halo — a restart
    # depends on everything in front of a position inside its haystack and on nothing outside it.  Rank order = the
    # sequential result (oracle: automaton_search_iter_long_next, src/AutomatonSearchIterLong.c:89-153)
    for splitter in (lambda: shard_range(len(reads), rank, world), lambda: shard_range_by_bytes(off, rank, world)):
        lo, hi = splitter()
        loc_off, loc_e, loc_v = [0], [], []
        for h in range(lo, hi):
            pairs = orc.flat_iter_long(got, data[off[h]:off[h + 1]])
            loc_e += [p[0] for p in pairs]
            loc_v += [p[1] for p in pairs]
            loc_off.append(len(loc_e))
        res = gather_csr(np.array(loc_off, dtype=np.int64), np.array(loc_e, dtype=np.int32), np.array(loc_v, dtype=np.int32))
        if rank == 0:
            O = orc.Oracle()
            for i, k in enumerate(keys):
                O.add_word(k, i)
            O.make_automaton()
            mo, e, v = O.batch(data, off, 1)
            assert len(e) > 100
            assert np.array_equal(res[0], mo) and np.array_equal(res[1], e) and np.array_equal(res[2], v)
    # one long haystack (config-3 style), cut with a longest_word - 1 halo: every rank scans its own shard plus the
    # halo in front of it, keeps the matches that end in its own range; rank order = the sequential result
    from pyahocorasick_amd.workloads import text_corpus, text_keys, text_vocab
    vocab = text_vocab(3000, seed=2)
    tkeys = text_keys(vocab, 400, seed=3)
    corpus = text_corpus(vocab, 60_000, seed=4)
    for k in tkeys[:40]:
        pos = int.from_bytes(k[:4], "little") % (len(corpus) - 64)
        corpus[pos:pos + len(k)] = np.frombuffer(k, dtype=np.uint8)
    tblob = None
    if rank == 0:
        B = acx.Automaton(acx.STORE_INTS)
        for i, k in enumerate(tkeys):
            B.add_word(k, i)
        B.make_automaton()
        tblob = B.flat_image_bytes()
    tgot = broadcast_blob(tblob, src=0).numpy().tobytes()
    longest = max(len(k) for k in tkeys)
    hay = corpus.tobytes()
    s0, lo, hi = halo_shard(len(hay), rank, world, longest)
    pairs, _ = orc.flat_iter(tgot, hay[s0:hi])
    mine = [(e + s0, v) for e, v in pairs if e + s0 >= lo]
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    if rank == 0:
        O = orc.Oracle()
        for i, k in enumerate(tkeys):
            O.add_word(k, i)
        O.make_automaton()
        whole = O.iter(hay)
        assert len(whole) >= 40
        assert [x for part in parts for x in part] == whole
        # without the halo the cut loses the matches that straddle it (the test would not notice a missing halo otherwise)
        lo1, hi1 = shard_range(len(hay), 1, world)
        no_halo, _ = orc.flat_iter(tgot, hay[lo1:hi1])
        straddle = [m for m in whole if lo1 <= m[0] < lo1 + longest - 1]
        assert len([1 for e, v in no_halo if e + lo1 < lo1 + longest - 1]) <= len(straddle)
    # ---- iter_long: the dictionary of its position-parallel form rides behind the blob in ONE broadcast (parallel.pack_with_long):
    #      every rank finds, at the offset rank 0 announces, a pack whose dictionary image validates and whose values are rank 0's
    from pyahocorasick_amd.parallel import pack_with_long
    import struct
    import torch
    payload, pack_off = (pack_with_long(blob) if rank == 0 else (b"", 0))
    meta = torch.tensor([len(blob) if rank == 0 else 0, pack_off], dtype=torch.int64)
    dist.broadcast(meta, src=0)
    blob_bytes, pack_off = int(meta[0]), int(meta[1])
    pgot = broadcast_blob(payload, src=0).numpy().tobytes()
    assert pgot[:blob_bytes] == got and pack_off % 256 == 0 and pack_off >= blob_bytes
    pack = pgot[pack_off:]
    magic, total, d_off, d_bytes, real_off, n_real, longest_d = struct.unpack_from("<QQQQQQI", pack, 0)
    assert magic == int.from_bytes(b"ACXLONG1", "little") and total == len(pack) and d_bytes > 0 and n_real > len(keys)
    dbuf = C.create_string_buffer(pack[d_off:d_off + d_bytes], d_bytes)
    _lib.check(_lib.lib().acx_blob_validate(dbuf, d_bytes))
    digests = [None] * world
    dist.all_gather_object(digests, hashlib.sha256(pack).hexdigest())
    assert len(set(digests)) == 1
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("PARALLEL_CPU_OK")


if __name__ == "__main__":
    main()
