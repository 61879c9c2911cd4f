"""Multi-rank host logic on CPU: the landmark partition and the torch.distributed
all-reduce hook (gloo, world_size 2) that bench.py installs with backend nccl (RCCL) on
the GPUs.  The engine itself needs a GPU; its sharded path is covered on one device by
tests/test_gpu_parity.py::test_two_shards_equal_one."""
import ctypes
import os
import socket

import numpy as np
import pytest

from ba_amd import sharding


def test_partition_is_contiguous_complete_and_balanced():
    rng = np.random.default_rng(0)
    k = rng.integers(2, 40, 10000)
    for n in (1, 2, 3, 8):
        sh = sharding.landmark_shards(k, n)
        assert sh[0][0] == 0 and sh[-1][1] == len(k)
        assert all(sh[i][1] == sh[i + 1][0] for i in range(n - 1))
        work = [float((k[a:b] * (k[a:b] + 1) / 2).sum()) for a, b in sh]
        assert max(work) < 1.05 * (sum(work) / n) + 40 * 41 / 2
    # degenerate: more ranks than landmarks
    sh = sharding.landmark_shards([3, 3], 4)
    assert sh[0][0] == 0 and sh[-1][1] == 2 and all(a <= b for a, b in sh)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hook = sharding.torch_allreduce_hook(dist, "cpu")
    # what the engine hands to the hook: raw pointers to doubles / uint64 histograms
    s_part = np.arange(12, dtype=np.float64) * (rank + 1)
    hist = np.array([rank, 5, 7 * rank], dtype=np.uint64)
    rc1 = hook(s_part.ctypes.data_as(ctypes.c_void_p).value, s_part.size, 0)
    rc2 = hook(hist.ctypes.data_as(ctypes.c_void_p).value, hist.size, 1)
    # the shard partition every rank computes must agree
    lo, hi = sharding.landmark_shards(np.full(1000, 10), world)[rank]
    cnt = np.array([float(hi - lo)])
    hook(cnt.ctypes.data_as(ctypes.c_void_p).value, 1, 0)
    # the collectives of the distributed solve: broadcast from rank 1, in-place reduce-scatter
    coll = sharding.torch_collectives_hook(dist, "cpu")
    msg = np.full(5, 10.0 + rank)
    rc3 = coll(1, msg.ctypes.data_as(ctypes.c_void_p).value, msg.size, 1)
    chunks = np.arange(8, dtype=np.float64) + 100.0 * rank      # 2 chunks of 4
    rc4 = coll(2, chunks.ctypes.data_as(ctypes.c_void_p).value, 4, 0)
    mine = chunks[4 * rank:4 * rank + 4]
    # the point-to-point block rows (ops 3 / 4): both ranks send first (a send never blocks on the receiver),
    # then receive — the order the engine uses inside one exchange group
    p2p_out = np.full(6, 1000.0 + rank)
    p2p_in = np.zeros(6)
    rc5 = coll(3, p2p_out.ctypes.data_as(ctypes.c_void_p).value, p2p_out.size, 1 - rank)
    p2p_out[:] = -1.0     # the hook copied the message: the caller may reuse its staging buffer
    rc6 = coll(4, p2p_in.ctypes.data_as(ctypes.c_void_p).value, p2p_in.size, 1 - rank)
    if rank == 0:
        np.save(out, np.concatenate([[rc1, rc2], s_part, hist.astype(np.float64), cnt, [rc3, rc4], msg, mine,
                                     [rc5, rc6], p2p_in]))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_allreduce_hook_world_size_2(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "r0.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = np.load(out)
    assert r[0] == 0 and r[1] == 0
    assert np.array_equal(r[2:14], np.arange(12) * 3.0)   # (rank0: x1) + (rank1: x2)
    assert np.array_equal(r[14:17], [1.0, 10.0, 7.0])
    assert r[17] == 1000.0                                  # shards cover all landmarks
    assert r[18] == 0 and r[19] == 0
    assert np.array_equal(r[20:25], np.full(5, 11.0))       # broadcast from rank 1
    assert np.array_equal(r[25:29], 2 * np.arange(4) + 100.0)  # rank 0's chunk, summed over ranks
    assert r[29] == 0 and r[30] == 0
    assert np.array_equal(r[31:37], np.full(6, 1001.0))     # rank 1's message, as it was when it was sent
