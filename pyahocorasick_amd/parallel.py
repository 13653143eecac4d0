"""Multi-GPU plumbing: one process per GPU, `torch.distributed` with the
"nccl" backend (= RCCL over xGMI on ROCm).

The path shards embarrassingly: haystacks are independent, so each rank scans a
contiguous slice of the batch and there is NO collective on the data path.  The only
collective is at setup: the flat automaton image (one contiguous blob, include/acx_blob.h)
is replicated from the rank that built it with ONE broadcast.  Results are concatenated in
rank order, which is the order a sequential scan of the whole batch produces.

torch is used for device memory and the process group only; the scan itself is libacx.
The same functions run on CPU tensors with the "gloo" backend (tests/test_parallel_cpu.py)
— there they move bytes and partition work, they never scan.
"""
import numpy as np

from ._lib import ACX_BLOB_HEADER_BYTES


def shard_range(n_items, rank, world):
    """contiguous, balanced-by-count slice [lo, hi) of rank `rank`"""
    lo = (n_items * rank) // world
    hi = (n_items * (rank + 1)) // world
    return lo, hi


def shard_range_by_bytes(offsets, rank, world):
    """contiguous slice balanced by BYTES (variable-length haystacks): rank r takes
    the haystacks whose start offset falls in [r*total/world, (r+1)*total/world)."""
    off = np.asarray(offsets, dtype=np.int64)
    total = int(off[-1])
    n = len(off) - 1
    cuts = [int(np.searchsorted(off[:-1], (total * r) // world, side="left")) for r in range(world + 1)]
    cuts[0], cuts[-1] = 0, n
    return cuts[rank], cuts[rank + 1]


def halo_shard(total_len, rank, world, longest_word):
    """One long haystack cut for `world` ranks: rank r owns the end positions [lo, hi) and scans the
    bytes [s0, hi), s0 = lo - (longest_word - 1) clipped at 0 — every match ending in [lo, hi) depends only on the
    longest_word bytes up to its end, so the shards are independent (exact for iter, not for iter_long).  The rank
    keeps the matches whose end index is >= lo; concatenated in rank order they are the sequential result.
    Returns (s0, lo, hi)."""
    lo, hi = shard_range(total_len, rank, world)
    s0 = max(0, lo - max(0, int(longest_word) - 1))
    return s0, lo, hi


def broadcast_blob(blob, src=0, device=None):
    """Replicate the flat image bytes.  Returns a uint8 tensor on `device` holding the blob on
    every rank.  ONE broadcast for the payload (preceded by an 8-byte size broadcast so the
    receivers can allocate)."""
    import torch
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    device = device if device is not None else torch.device("cpu")
    if not distributed:
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
        return t.to(device)
    rank = dist.get_rank()
    size = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(size, src=src)
    nbytes = int(size.item())
    if rank == src:
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    else:
        t = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dist.broadcast(t, src=src)          # the single RCCL broadcast of the automaton
    return t


def pack_with_long(blob):
    """blob + the iter_long dictionary built from it (device.long_pack, host only), laid end to end for ONE broadcast: the pack starts
    at the first 256-byte boundary behind the blob.  Returns (payload bytes, offset of the pack)."""
    from .device import long_pack
    pack = long_pack(blob)
    off = (len(blob) + 255) & ~255
    return bytes(blob) + b"\0" * (off - len(blob)) + pack, off


def broadcast_image(blob, src=0, device=None, long_pack=False):
    """broadcast_blob + adopt the received device buffer as a libacx image (no extra copy).
    Returns (Image, tensor); the Image keeps the tensor alive.

    long_pack=True (ACX_SCAN_LONG workloads): the source rank builds the dictionary of the position-parallel iter_long ONCE from its
    host blob and sends it behind the blob in the same broadcast; every rank adopts it in place (acx_image_set_long) — otherwise each
    rank's first iter_long scan copies its whole image back to the host and builds the dictionary there."""
    import torch
    import torch.distributed as dist
    from .device import Image
    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if distributed else src
    payload, pack_off = blob, 0
    if long_pack:
        if rank == src:
            payload, pack_off = pack_with_long(blob)
        meta = torch.tensor([len(blob) if rank == src else 0, pack_off], dtype=torch.int64, device=device if device is not None else torch.device("cpu"))
        if distributed:
            dist.broadcast(meta, src=src)
        blob_bytes, pack_off = int(meta[0].item()), int(meta[1].item())
    t = broadcast_blob(payload if rank == src else b"", src, device)
    if t.device.type != "cuda":
        raise RuntimeError("broadcast_image needs a GPU tensor; the scan has no CPU path")
    if not long_pack:
        blob_bytes = t.numel()
    header = bytes(t[:ACX_BLOB_HEADER_BYTES].cpu().numpy().tobytes())
    img = Image.adopt(t.data_ptr(), blob_bytes, header, keepalive=t)
    if long_pack:
        img.set_long(t.data_ptr() + pack_off, t.numel() - pack_off, on_device=True)
    return img, t


def gather_csr(local_off, local_end, local_val, dst=0):
    """Concatenate per-rank CSR results in rank order on `dst` (host side; results of
    different haystacks are independent, so this IS the sequential order).  Returns
    (match_off, end_index, value) on dst, None elsewhere."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_off, local_end, local_val
    world, rank = dist.get_world_size(), dist.get_rank()
    parts = [None] * world if rank == dst else None
    dist.gather_object((np.asarray(local_off), np.asarray(local_end), np.asarray(local_val)), parts, dst=dst)
    if rank != dst:
        return None
    offs, ends, vals, base = [np.zeros(1, dtype=np.int64)], [], [], 0
    for o, e, v in parts:
        offs.append(o[1:] + base)
        base += int(o[-1])
        ends.append(e)
        vals.append(v)
    return np.concatenate(offs), np.concatenate(ends), np.concatenate(vals)
