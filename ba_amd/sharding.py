"""Landmark sharding for multi-GPU runs.

The reduced pose system is a sum of independent per-landmark terms
(/root/reference/src/BundleAdjuster.cpp:409-485), so landmarks — with all their
observations — are split into contiguous ranges, one per rank; poses, cameras and masks
are replicated; pose-pose residuals (unary/binary/IMU) are added on rank 0 only.  Per
iteration the engine asks for a cross-shard SUM of: the lower storage of S with its rhs
row, the un-reduced rhs_p, a few scalars, and the selection histograms of the Huber median
(doubles or uint64 counts).  This module provides the partition and two all-reduce hooks
for ba_hip_set_allreduce: torch.distributed (RCCL on GPUs, gloo on CPU buffers) and an
in-process hook for several engines driven by threads on one device (tests).
"""
import ctypes
import threading

import numpy as np


def landmark_shards(obs_per_landmark, nranks):
    """Contiguous landmark ranges [lo, hi) balanced by the Schur work  sum k(k+1)/2  of the
    landmarks' track lengths k (not by count)."""
    k = np.asarray(obs_per_landmark, dtype=np.float64)
    work = np.concatenate([[0.0], np.cumsum(k * (k + 1) / 2.0)])
    total = work[-1]
    bounds = [0]
    for r in range(1, nranks):
        bounds.append(int(np.searchsorted(work, total * r / nranks, side="left")))
    bounds.append(len(k))
    bounds = np.maximum.accumulate(np.array(bounds))
    return [(int(bounds[r]), int(bounds[r + 1])) for r in range(nranks)]


def landmark_shards_along_trajectory(lm_ref_pose, obs_per_landmark, nranks):
    """Landmark id arrays, one per rank: the landmarks in the order of their reference pose, cut into ranges balanced
    by the Schur work  sum k(k+1)/2  like landmark_shards.  A shard then touches the poses of one stretch of the
    trajectory (plus the visibility window either side), i.e. a band of the reduced pose system instead of all of it:
    the sparse exchange of S (ba_amd/csrc/k_chol.hip: dist_scatter_S_sparse) moves a fraction of what shards that are
    contiguous in a spatially random id do.  Ids inside a shard stay ascending."""
    ref = np.asarray(lm_ref_pose)
    order = np.argsort(ref, kind="stable")
    k = np.broadcast_to(np.asarray(obs_per_landmark, dtype=np.float64), ref.shape)[order]
    return [np.sort(order[lo:hi]).astype(np.int64) for lo, hi in landmark_shards(k, nranks)]


class _DevArray:
    """__cuda_array_interface__ view of a raw device pointer."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr,
                                         "data": (ptr, False), "version": 2}


def _host_view(ptr, count, dtype):
    ctype = ctypes.c_double if dtype == 0 else ctypes.c_uint64
    return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctype)), shape=(count,))


def torch_allreduce_hook(dist, device="cuda"):
    """Hook (ptr, count, dtype) -> 0 summing over the torch.distributed default group.
    device="cuda": ptr is device memory (backend nccl = RCCL over xGMI);
    device="cpu": ptr is host memory (backend gloo; used by the CPU tests)."""
    import torch

    def fn(ptr, count, dtype):
        try:
            if device == "cuda":
                t = torch.as_tensor(_DevArray(ptr, count, "<f8" if dtype == 0 else "<i8"), device="cuda")
                # S of the 10k-pose scene is a 14 GB message: hand it to RCCL in 4 GB pieces
                step = 1 << 29
                for lo in range(0, count, step):
                    dist.all_reduce(t[lo:lo + step], op=dist.ReduceOp.SUM)
                # the collective is ordered into torch's current stream: wait for that stream only
                # (a device-wide synchronize would also wait for the engine's look-ahead stream)
                torch.cuda.current_stream().synchronize()
            else:
                a = _host_view(ptr, count, dtype)
                t = torch.from_numpy(a.view(np.int64) if dtype == 1 else a)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return 0
        except Exception as exc:  # the engine reports "allreduce hook failed"
            import sys
            print("allreduce hook:", exc, file=sys.stderr)
            return 1
    return fn


def torch_collectives_hook(dist, device="cuda"):
    """Hook (op, ptr, count, root) -> 0 for ba_hip_set_collectives over the torch.distributed
    default group (RCCL): op 1 = broadcast `count` doubles from `root`; op 2 = reduce-scatter
    (sum) of world_size chunks of `count` doubles, the result lands in this rank's chunk; op 3 / 4 =
    send to / receive from rank `root` (point-to-point messages of the distributed solve)."""
    import torch
    pending = []   # (work, buffer) of the sends in flight
    deferred = []  # nccl backend: (buffer, destination) of the sends waiting for their batch

    def fn(op, ptr, count, root):
        try:
            world, rank = dist.get_world_size(), dist.get_rank()
            nccl = device == "cuda" and dist.get_backend() == "nccl"
            if op == 5:      # end of an exchange
                if nccl and deferred:   # sends nobody batched with a receive (this rank receives nothing)
                    for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, b, d) for b, d in deferred]):
                        w.wait()
                    deferred.clear()
                return 0
            n = count * world if op == 2 else count
            if device == "cuda":
                t = torch.as_tensor(_DevArray(ptr, n, "<f8"), device="cuda")
            else:  # host buffers (gloo; CPU tests)
                t = torch.from_numpy(_host_view(ptr, n, 0))
            if op == 1:
                dist.broadcast(t, src=root)
            elif op == 3:   # send to `root`: must not block on the receiver
                if nccl:
                    # RCCL point to point outside a group deadlocks when two ranks send to each other first: the
                    # sends are deferred and launched in ONE batch with the first receive (or at op 5)
                    deferred.append((t.clone(), root))
                    return 0
                # (gloo moves host memory: device buffers are staged through the host for it)
                buf = t.cpu() if (device == "cuda" and dist.get_backend() == "gloo") else t.clone()
                pending.append((dist.isend(buf, dst=root), buf))
                return 0
            elif op == 4:   # receive from `root`
                if nccl:
                    ops = [dist.P2POp(dist.isend, b, d) for b, d in deferred] + [dist.P2POp(dist.irecv, t, root)]
                    for w in dist.batch_isend_irecv(ops):
                        w.wait()
                    deferred.clear()
                elif device == "cuda" and dist.get_backend() == "gloo":
                    tmp = torch.empty(count, dtype=torch.float64)
                    dist.recv(tmp, src=root)
                    t.copy_(tmp)
                else:
                    dist.recv(t, src=root)
                pending[:] = [(w, b) for (w, b) in pending if not w.is_completed()]
            elif op == 2:
                if device == "cuda" and dist.get_backend() != "gloo":
                    out = torch.empty(count, dtype=torch.float64, device="cuda")
                    dist.reduce_scatter_tensor(out, t, op=dist.ReduceOp.SUM)
                    t[rank * count:(rank + 1) * count].copy_(out)
                    del out
                else:  # gloo has no reduce-scatter: sum everything, keep the own chunk
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
            else:
                raise ValueError("unknown collective op %d" % op)
            if device == "cuda":
                torch.cuda.current_stream().synchronize()  # not device-wide: see torch_allreduce_hook
            return 0
        except Exception as exc:  # the engine reports the failure
            import sys
            print("collectives hook:", exc, file=sys.stderr)
            return 1
    return fn


class ThreadAllReduce:
    """Sums buffers of `nranks` engines that live in one process and are driven by one
    thread each (shards emulated on a single GPU)."""

    def __init__(self, nranks):
        import torch
        self.torch = torch
        self.n = nranks
        self.barrier = threading.Barrier(nranks)
        self.slots = [None] * nranks
        self.failed = False
        import queue
        self.mail = {(a, b): queue.Queue() for a in range(nranks) for b in range(nranks)}

    def hook(self, rank):
        torch = self.torch

        def fn(ptr, count, dtype):
            try:
                t = torch.as_tensor(_DevArray(ptr, count, "<f8" if dtype == 0 else "<i8"), device="cuda")
                self.slots[rank] = t
                self.barrier.wait()
                if rank == 0:
                    total = self.slots[0].clone()
                    for r in range(1, self.n):  # fixed rank order: reproducible sums
                        total += self.slots[r]
                    self.total = total
                    torch.cuda.synchronize()
                self.barrier.wait()
                t.copy_(self.total)
                torch.cuda.synchronize()
                self.barrier.wait()
                return 0
            except Exception as exc:
                import sys
                print("ThreadAllReduce:", exc, file=sys.stderr)
                self.failed = True
                try:
                    self.barrier.abort()
                except Exception:
                    pass
                return 1
        return fn

    def collectives(self, rank):
        """Broadcast / reduce-scatter among the engines of this process (see hook())."""
        torch = self.torch

        def fn(op, ptr, count, root):
            try:
                if op == 5:      # end of an exchange: sends are eager here
                    return 0
                n = count * self.n if op == 2 else count
                t = torch.as_tensor(_DevArray(ptr, n, "<f8"), device="cuda")
                if op == 3:      # send: a private copy goes into the receiver's mailbox, no waiting
                    buf = t.clone()
                    torch.cuda.synchronize()
                    self.mail[(rank, root)].put(buf)
                    return 0
                if op == 4:      # receive: blocks until the sender's copy is there
                    buf = self.mail[(root, rank)].get(timeout=120)
                    assert buf.numel() == count
                    t.copy_(buf)
                    torch.cuda.synchronize()
                    return 0
                self.slots[rank] = t
                self.barrier.wait()
                if op == 1:
                    if rank != root:
                        t.copy_(self.slots[root])
                elif op == 2:
                    mine = self.slots[0][rank * count:(rank + 1) * count].clone()
                    for r in range(1, self.n):  # fixed rank order
                        mine += self.slots[r][rank * count:(rank + 1) * count]
                    torch.cuda.synchronize()
                    self.barrier.wait()  # everybody has read the inputs before anybody overwrites
                    t[rank * count:(rank + 1) * count].copy_(mine)
                else:
                    raise ValueError("unknown collective op %d" % op)
                torch.cuda.synchronize()
                self.barrier.wait()
                return 0
            except Exception as exc:
                import sys
                print("ThreadAllReduce.collectives:", exc, file=sys.stderr)
                self.failed = True
                try:
                    self.barrier.abort()
                except Exception:
                    pass
                return 1
        return fn
