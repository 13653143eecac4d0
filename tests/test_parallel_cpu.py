"""CPU suite: the N>1 path with world_size 2 over gloo."""
import os
import socket
import subprocess
import sys

import numpy as np

from pyahocorasick_amd.parallel import shard_range, shard_range_by_bytes

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_ranges_cover_and_balance():
    for n in (0, 1, 7, 64, 1000):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
    off = np.concatenate([[0], np.cumsum([10] * 50 + [1000] * 5 + [10] * 45)])
    spans = [shard_range_by_bytes(off, r, 4) for r in range(4)]
    assert spans[0][0] == 0 and spans[-1][1] == 100 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    byts = [int(off[hi] - off[lo]) for lo, hi in spans]
    assert max(byts) <= 2 * (off[-1] // 4) + 1000


def test_world_size_2_gloo_broadcast_shard_gather():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "_parallel_worker.py")]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "PARALLEL_CPU_OK" in out, out[-3000:]
