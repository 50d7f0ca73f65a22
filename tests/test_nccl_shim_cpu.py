"""CPU test of tests/nccl_shim (the host-staged stand-in for RCCL that lets several rank processes of the team transport
share one GPU in tests/test_gpu_team_ranks.py): 2, 3 and 5 processes, host buffers (NCCL_SHIM_HOST=1).  The pattern is the
one csrc/team.h issues: communicator bootstrap from a broadcast id, ncclCommCount, an in-place ncclAllGather of 32 64-bit
words per rank, one ncclSend/ncclRecv group with several messages of different sizes per neighbour pair, ncclBroadcast."""
import ctypes as C
import multiprocessing as mp
import os

import numpy as np
import pytest


def _load():
    from tests import nccl_shim
    lib = C.CDLL(nccl_shim.build())
    lib.ncclGetErrorString.restype = C.c_char_p
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclBroadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    return lib


U64, F64, U8 = 5, 8, 1


class UniqueId(C.Structure):                                             # ncclUniqueId travels BY VALUE into ncclCommInitRank
    _fields_ = [("internal", C.c_char * 128)]


def _msg(src, dst, k, n):
    return np.arange(n, dtype=np.float64) + 1000.0 * src + 100.0 * dst + 10.0 * k


def _worker(rank, world, ident, q):
    os.environ["NCCL_SHIM_HOST"] = "1"
    os.environ["NCCL_SHIM_TIMEOUT_MS"] = "20000"
    try:
        lib = _load()

        def ck(rc):
            assert rc == 0, lib.ncclGetErrorString(rc)
        comm = C.c_void_p()
        ck(lib.ncclCommInitRank(C.byref(comm), world, UniqueId.from_buffer_copy(ident), rank))
        n = C.c_int()
        ck(lib.ncclCommCount(comm, C.byref(n)))
        assert n.value == world
        for rep in range(3):                                            # the same communicator, several rounds: counters and slots are reused
            g = np.zeros(world * 32, dtype=np.uint64)
            g[rank * 32:(rank + 1) * 32] = np.arange(32, dtype=np.uint64) + 100 * rank + 10000 * rep
            ck(lib.ncclAllGather(g[rank * 32:].ctypes.data, g.ctypes.data, 32, U64, comm, None))      # in place, as team.h does
            assert np.array_equal(g, np.concatenate([np.arange(32, dtype=np.uint64) + 100 * j + 10000 * rep for j in range(world)]))
            # halo group: three messages each way with both ring neighbours, sizes differ per message
            sizes = (17, 4128, 300)
            peers = sorted({(rank - 1) % world, (rank + 1) % world} - {rank})
            recv = {(p, k): np.full(sizes[k], -1.0) for p in peers for k in range(3)}
            send = {(p, k): _msg(rank, p, k + 3 * rep, sizes[k]) for p in peers for k in range(3)}
            ck(lib.ncclGroupStart())
            for p in peers:
                for k in range(3):
                    ck(lib.ncclSend(send[p, k].ctypes.data, sizes[k], F64, p, comm, None))
            for p in peers:
                for k in range(3):
                    ck(lib.ncclRecv(recv[p, k].ctypes.data, sizes[k], F64, p, comm, None))
            ck(lib.ncclGroupEnd())
            for p in peers:
                for k in range(3):
                    assert np.array_equal(recv[p, k], _msg(p, rank, k + 3 * rep, sizes[k])), (rank, p, k)
            b = np.frombuffer(bytes([rank + 1 + rep] * 128), dtype=np.uint8).copy()
            ck(lib.ncclBroadcast(b.ctypes.data, b.ctypes.data, 128, U8, 0, comm, None))
            assert (b == 1 + rep).all()
        lib.ncclCommDestroy(comm)
        q.put((rank, "ok"))
    except BaseException as e:                                           # noqa: BLE001 -- report to the parent, whatever it was
        q.put((rank, repr(e)))


@pytest.mark.parametrize("world", [2, 3, 5])
def test_shim_collectives_between_processes(world):
    lib = _load()
    ident = UniqueId()
    assert lib.ncclGetUniqueId(C.byref(ident)) == 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, bytes(ident), q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
    assert got == {r: "ok" for r in range(world)}, got


def test_shim_reports_a_missing_peer_instead_of_hanging(monkeypatch):
    monkeypatch.setenv("NCCL_SHIM_HOST", "1")
    monkeypatch.setenv("NCCL_SHIM_TIMEOUT_MS", "300")
    lib = _load()
    ident = UniqueId()
    assert lib.ncclGetUniqueId(C.byref(ident)) == 0
    comm = C.c_void_p()
    rc = lib.ncclCommInitRank(C.byref(comm), 2, ident, 0)               # rank 1 never shows up
    assert rc != 0 and b"timed out" in lib.ncclGetErrorString(rc)
