"""Row-slab domain decomposition of the CG path across the GPUs of one node: one process per
GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI) for the two things that cross ranks.

The reference is single-process (no collectives anywhere); this module is the
scaling surface around the same kernels.  Two ways to move the halo, same kernels:

  halo="gather" (default): ONE collective per phase.  The update phase's record = the rank's reduced partial
    sums followed by the first and last owned row of the residual; `all_gather_into_tensor` makes every record
    visible everywhere and a small kernel copies the two neighbours' rows into the ghost rows.  The stencil
    phase's record is the sums only (16 doubles).  Per iteration:
        stencil (whole slab) ; flip ; record{(Ap,p),(r,p)} ; all_gather
        update x, r          ;        record{r.r, max-norms | r rows} ; all_gather ; scatter ghost rows
  halo="p2p": the residual's boundary rows travel as isend/irecv pairs with the two neighbours and overlap
    compute; both phases all-gather 16 doubles per rank.  Per iteration:
        stencil (interior rows)      -- overlaps the r rows still in flight
        wait halo ; stencil (first + last owned row) ; flip ; all_gather sums
        update x, r ; all_gather sums ; isend/irecv r rows    -- overlap the next interior stencil
  The DIRECTION never crosses ranks: the stencil launch recomputes p_new on its halo anyway (from the ghost copies
  of r and p_old) and keeps the result in the ghost rows, bit-identical to the neighbour's rows, so the update
  launch -- which rebuilds A p from p and therefore reads p's ghost rows -- finds them locally.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from . import _capi
from .solver import default_params



def slab_rows(n: int, world: int, rank: int):
    """Owned grid rows [y_lo, y_hi] of `rank`: contiguous slabs balanced by unknown count."""
    lo, hi = C.c_int(), C.c_int()
    _capi.check(_capi.load().mi355cg_slab_rows(n, world, rank, C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


class _DevView:
    """Zero-copy torch view of device memory owned by libmi355cg (via __cuda_array_interface__)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8",
                                         "data": (int(ptr), False), "version": 2}


def _view(ptr: int, count: int, device) -> torch.Tensor:
    return torch.as_tensor(_DevView(ptr, count), device=device)


class SlabEngine:
    """One rank's slab on one GPU: thin wrapper over the mi355cg_dist_* C ABI."""

    def __init__(self, n: int, y_lo: int, y_hi: int, device: int = 0,
                 domain=(1.0, 2.0, 1.0, 2.0)):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        a, b, c, d = domain
        rc = self._lib.mi355cg_create_slab(n, n, a, b, c, d, _capi.F64, device, y_lo, y_hi, C.byref(self._h))
        if rc == _capi.ERR_INVALID:
            raise ValueError(self._lib.mi355cg_last_error().decode())
        _capi.check(rc)
        self.n, self.y_lo, self.y_hi = n, y_lo, y_hi
        self.device = torch.device("cuda", device)
        pb, pl = C.c_longlong(), C.c_longlong()
        _capi.check(self._lib.mi355cg_owned_range(self._h, C.byref(pb), C.byref(pl), None, None))
        self.packed_begin, self.packed_len = pb.value, pl.value
        hdr = C.c_int()
        _capi.check(self._lib.mi355cg_dist_record_layout(self._h, C.byref(hdr), None, None))
        self.rec_header = hdr.value                         # doubles of sums/maxes at the head of a record
        self._sums = {}
        self._halo_cache = {}
        for which in (0, 1):
            p, cnt = C.c_void_p(), C.c_int()
            _capi.check(self._lib.mi355cg_dist_sums_ptr(self._h, which, C.byref(p), C.byref(cnt)))
            self._sums[which] = _view(p.value, cnt.value, self.device)

    def close(self):
        if self._h:
            self._lib.mi355cg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    # -- engine protocol ----------------------------------------------------------------------
    def begin(self, params: _capi.Params):
        _capi.check(self._lib.mi355cg_dist_begin(self._h, C.byref(params), self._stream()))

    def reduce(self, which: int, with_rows: bool = False):
        _capi.check(self._lib.mi355cg_dist_reduce(self._h, which, 1 if with_rows else 0, self._stream()))

    def record(self, which: int) -> torch.Tensor:
        """This rank's record: [sums and maxes (rec_header doubles) | first owned row | last owned row]."""
        return self._sums[which]

    def scatter_ghosts(self, vector: int, gathered: torch.Tensor, rank: int):
        nranks = gathered.numel() // self._sums[0].numel()
        _capi.check(self._lib.mi355cg_dist_scatter_ghosts(self._h, vector, gathered.data_ptr(), nranks, rank, self._stream()))

    def stencil(self, gathered_b: torch.Tensor, estride: int, rows: int = 0):
        _capi.check(self._lib.mi355cg_dist_stencil(self._h, gathered_b.data_ptr(), gathered_b.numel() // estride,
                                                   estride, rows, self._stream()))

    def flip(self):
        _capi.check(self._lib.mi355cg_dist_flip(self._h))

    def update(self, gathered_a: torch.Tensor, estride: int, rows: int = 0):
        _capi.check(self._lib.mi355cg_dist_update(self._h, gathered_a.data_ptr(), gathered_a.numel() // estride,
                                                  estride, rows, self._stream()))

    @property
    def update_reads_ghosts(self) -> bool:
        """True when the update phase recomputes A p (8-word iteration): its edge rows read the direction's ghost rows."""
        return bool(self._lib.mi355cg_dist_update_reads_ghosts(self._h))

    def check(self, gathered_b: torch.Tensor, estride: int):
        _capi.check(self._lib.mi355cg_dist_check(self._h, gathered_b.data_ptr(), gathered_b.numel() // estride,
                                                 estride, self._stream()))

    def summary(self):
        torch.cuda.current_stream().synchronize()
        res, done = _capi.Results(), C.c_int()
        _capi.check(self._lib.mi355cg_dist_summary(self._h, C.byref(res), C.byref(done)))
        return res, bool(done.value)

    def finish(self):
        _capi.check(self._lib.mi355cg_dist_finish(self._h, self._stream()))

    def history(self, it: int):
        p, r, e = C.c_double(), C.c_double(), C.c_double()
        _capi.check(self._lib.mi355cg_dist_history(self._h, it, C.byref(p), C.byref(r), C.byref(e)))
        return p.value, r.value, e.value

    def halo(self, vector: int):
        """Boundary rows of vector (0 = r, 1 = current direction) as tensor views."""
        sl, rl, sh, rh = (C.c_void_p() for _ in range(4))
        nl, nh, fl, fh = (C.c_longlong() for _ in range(4))
        _capi.check(self._lib.mi355cg_dist_halo(self._h, vector, C.byref(sl), C.byref(rl), C.byref(nl),
                                                C.byref(sh), C.byref(rh), C.byref(nh)))
        key = (sl.value, rl.value, sh.value, rh.value)
        if key in self._halo_cache:
            return self._halo_cache[key]
        _capi.check(self._lib.mi355cg_dist_halo_recv_counts(self._h, C.byref(fl), C.byref(fh)))
        self._halo_cache[key] = {"send_lo": _view(sl.value, nl.value, self.device), "recv_lo": _view(rl.value, fl.value, self.device),
                "send_hi": _view(sh.value, nh.value, self.device), "recv_hi": _view(rh.value, fh.value, self.device)}
        return self._halo_cache[key]

    def _owned(self, fn) -> np.ndarray:
        out = np.empty(self.packed_len)
        _capi.check(fn(self._h, out))
        return out

    def solution(self): return self._owned(self._lib.mi355cg_get_solution)
    def rhs(self): return self._owned(self._lib.mi355cg_get_rhs)
    def true_solution(self): return self._owned(self._lib.mi355cg_get_true_solution)
    def recursive_residual(self): return self._owned(self._lib.mi355cg_get_recursive_residual)


# ---------------------------------------------------------------------------------------------
class _Comm:
    """The two communication patterns, over whatever backend the process group has.  With
    "nccl" (RCCL) device tensors go straight to the collectives; with "gloo" (CPU tests, or
    several ranks sharing one GPU) device tensors are staged through host memory."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self.stage = self.backend == "gloo"
        # test hook: run the collectives even at world size 1 (exercises RCCL on a one-GPU box)
        self.force = os.environ.get("MI355CG_FORCE_COLLECTIVES", "0") == "1" and dist.is_initialized()

    def all_gather(self, out: torch.Tensor, local: torch.Tensor):
        if self.world == 1 and not self.force:
            out.copy_(local)
            return
        if self.stage and local.is_cuda:
            loc = local.cpu()
            parts = [torch.empty_like(loc) for _ in range(self.world)]
            dist.all_gather(parts, loc, group=self.group)
            out.copy_(torch.cat(parts).to(out.device))
        elif self.stage:
            parts = list(out.view(self.world, -1).unbind(0))
            dist.all_gather(parts, local, group=self.group)
        else:
            dist.all_gather_into_tensor(out, local, group=self.group)

    def halo_start(self, h: dict):
        """Exchange boundary rows with rank-1 / rank+1.  Returns a token for halo_wait."""
        if self.world == 1:
            return None
        lo, hi = self.rank - 1, self.rank + 1
        if self.stage:
            reqs, post = [], []
            for peer, send, recv in ((lo, "send_lo", "recv_lo"), (hi, "send_hi", "recv_hi")):
                if 0 <= peer < self.world:
                    s = h[send].cpu() if h[send].is_cuda else h[send]
                    r = torch.empty(h[recv].numel(), dtype=h[recv].dtype)
                    reqs.append(dist.isend(s.contiguous(), peer, group=self.group))
                    reqs.append(dist.irecv(r, peer, group=self.group))
                    post.append((h[recv], r))
            return ("staged", reqs, post)
        ops = []
        for peer, send, recv in ((lo, "send_lo", "recv_lo"), (hi, "send_hi", "recv_hi")):
            if 0 <= peer < self.world:
                ops.append(dist.P2POp(dist.isend, h[send], peer, self.group))
                ops.append(dist.P2POp(dist.irecv, h[recv], peer, self.group))
        return ("nccl", dist.batch_isend_irecv(ops)) if ops else None

    @staticmethod
    def halo_wait(token):
        if token is None:
            return
        if token[0] == "staged":
            for r in token[1]:
                r.wait()
            for dst, src in token[2]:
                dst.copy_(src.to(dst.device))
        else:
            for r in token[1]:
                r.wait()                      # stream-level wait for NCCL: the host does not block


@dataclass
class DistResults:
    iterations: int
    converged: bool
    stop_reason: int
    final_residual_norm: float
    final_precision: float
    final_error_norm: float
    r_norm2: float
    initial_r_norm2: float
    seconds: float


class DistributedCG:
    """CG over row slabs.  `engine` implements the SlabEngine protocol."""

    def __init__(self, engine, group=None, halo: str = "p2p", overlap: bool = True):
        assert halo in ("gather", "p2p")
        self.eng = engine
        self.comm = _Comm(group)
        self.halo = halo
        self.overlap = overlap and halo == "p2p"
        rec = engine.record(0)
        self.WA = engine.rec_header                                        # stencil phase: sums only
        self.W = rec.numel() if halo == "gather" else engine.rec_header   # update phase: sums [+ the residual's boundary rows]
        self.gA = torch.zeros(self.comm.world * self.WA, dtype=torch.float64, device=rec.device)
        self.gB = torch.zeros(self.comm.world * self.W, dtype=torch.float64, device=rec.device)

    def _gather(self, which: int):
        """All-gather the per-rank sums of a phase.  Only the update phase (which = 1) also moves rows: the residual's
        boundary rows.  The direction never crosses ranks -- the stencil launch keeps it in the ghost rows itself."""
        eng, comm = self.eng, self.comm
        if which == 1 and self.halo == "gather":
            eng.reduce(1, with_rows=True)
            comm.all_gather(self.gB, eng.record(1))
            if comm.world > 1:
                eng.scatter_ghosts(0, self.gB, comm.rank)
        else:
            eng.reduce(which, with_rows=False)
            comm.all_gather(self.gA if which == 0 else self.gB, eng.record(which)[:eng.rec_header])

    def _stop_anywhere(self, stop) -> bool:
        """A stop request on any rank stops every rank at the same poll (one tiny all-gather per poll).  Every rank takes part in
        that all-gather whether or not IT was given a `stop` callable: a rank without one contributes 0."""
        mine = torch.tensor([1.0 if (stop is not None and stop()) else 0.0], dtype=torch.float64)
        if self.comm.world == 1:
            return bool(mine.item())
        if not self.comm.stage:
            mine = mine.to(self.gA.device)
        out = torch.zeros(self.comm.world, dtype=torch.float64, device=mine.device)
        self.comm.all_gather(out, mine)
        return bool(out.max().item() > 0)

    def solve(self, params: _capi.Params, callback=None, stop=None) -> DistResults:
        """stop: optional callable; when it returns True on ANY rank the loop ends at the next poll on ALL ranks with
        stop reason INTERRUPTED (msg_solver.cpp:82-87; the reference polls every iteration, this harness every chunk)."""
        eng, comm, W = self.eng, self.comm, self.W
        p2p = self.halo == "p2p"
        msg = params.rule == _capi.RULE_MSG_MAXNORM
        t0 = time.perf_counter()
        eng.begin(params)                                   # x = 0, r = b, p = 0 ; partial norms of r0
        self._gather(1)                                     # (gather mode: also the ghost rows of r0 = b)
        tok_r = comm.halo_start(eng.halo(0)) if p2p else None
        eng.check(self.gB, W)
        res, done = eng.summary()
        if msg and callback:
            callback(0, res.final_precision, res.final_residual_norm, res.final_error_norm)
        sync_every = params.sync_every if params.sync_every > 0 else (100 if msg else 200)
        sync_every = min(sync_every, 512)
        every = params.callback_every
        it_done = 0
        WA = self.WA
        assert not eng.update_reads_ghosts                  # the stencil phase keeps the direction's ghost rows itself
        interrupted = False
        first = msg and every > 0                           # a function of the parameters only: the same on every rank
        while not done:
            if self._stop_anywhere(stop):
                interrupted = True
                break
            m = min(sync_every, max(1, params.max_iterations - it_done))
            if first:
                m, first = 1, False
            if msg and every > 0:
                m = min(m, every - it_done % every)
            for _ in range(m):
                if self.overlap and comm.world > 1:
                    eng.stencil(self.gB, W, rows=1)         # interior rows: no ghost needed
                    comm.halo_wait(tok_r)
                    eng.stencil(self.gB, W, rows=2)         # first + last owned row (+ the new direction's ghost rows)
                else:
                    comm.halo_wait(tok_r)
                    eng.stencil(self.gB, W, rows=0)
                eng.flip()
                self._gather(0)                             # (Ap, p), (r, p): 16 doubles per rank
                eng.update(self.gA, WA)                     # A p recomputed from the direction incl. its local ghost rows
                self._gather(1)
                if p2p:
                    tok_r = comm.halo_start(eng.halo(0))    # r's boundary rows (overlaps the next interior stencil)
            eng.check(self.gB, W)
            res, done = eng.summary()
            if msg and callback:
                for it in range(it_done + 1, res.iterations + 1):
                    stopped_here = done and res.stop_reason != _capi.STOP_ITERATIONS and it == res.iterations
                    if (it == 1 or (every > 0 and it % every == 0)) and not stopped_here:
                        callback(it, *eng.history(it))
            it_done = res.iterations
        comm.halo_wait(tok_r)
        eng.finish()                                        # flush the x update still pending after an odd iteration count
        if msg and callback:
            callback(res.iterations, res.final_precision, res.final_residual_norm, res.final_error_norm)
        if interrupted:
            return DistResults(res.iterations, False, _capi.STOP_INTERRUPTED, res.final_residual_norm,
                               res.final_precision, res.final_error_norm, res.r_norm2, res.initial_r_norm2,
                               time.perf_counter() - t0)
        return DistResults(res.iterations, bool(res.converged), res.stop_reason, res.final_residual_norm,
                           res.final_precision, res.final_error_norm, res.r_norm2, res.initial_r_norm2,
                           time.perf_counter() - t0)


# ---------------------------------------------------------------------------------------------
def decompose(n: int, world: int, decomp: int = _capi.DECOMP_ROWS):
    """[(y_lo, y_hi, x_lo, x_hi)] per part: rows inclusive, columns [x_lo, x_hi).  Pure host arithmetic."""
    lib = _capi.load()
    out = []
    for r in range(world):
        v = [C.c_int() for _ in range(4)]
        rc = lib.mi355cg_decompose(n, world, decomp, r, *[C.byref(i) for i in v])
        if rc == _capi.ERR_INVALID:
            raise ValueError(lib.mi355cg_last_error().decode())
        _capi.check(rc)
        out.append(tuple(i.value for i in v))
    return out


def halo_plan(n: int, world: int, decomp: int, rank: int):
    """The halo messages of `rank` per iteration (dicts with id, peer, send, kind, y0, y1, x0, x1, count)."""
    lib = _capi.load()
    cnt = C.c_int()
    _capi.check(lib.mi355cg_halo_plan(n, world, decomp, rank, 0, C.byref(cnt), None))
    msgs = (_capi.HaloMsg * max(cnt.value, 1))()
    _capi.check(lib.mi355cg_halo_plan(n, world, decomp, rank, cnt.value, C.byref(cnt), msgs))
    return [{f: getattr(m, f) for f, _ in _capi.HaloMsg._fields_} for m in msgs[:cnt.value]]


class Team:
    """The native multi-GPU loop (csrc/team.h) over the C ABI.

    Team.local(n, world, ...): this process drives every part (one or several GPUs).
    Team.rccl(n, ...): one process per GPU; the library creates its own RCCL communicator, the 128-byte id travels
    through torch.distributed (any backend) from rank 0."""

    def __init__(self, handle, n):
        self._lib = _capi.load()
        self._h = handle
        self.n = n
        w, nl, dc, sz = C.c_int(), C.c_int(), C.c_int(), C.c_longlong()
        _capi.check(self._lib.mi355cg_team_info(self._h, C.byref(w), C.byref(nl), C.byref(dc), C.byref(sz)))
        self.world, self.nlocal, self.decomp, self.size = w.value, nl.value, dc.value, sz.value

    @classmethod
    def local(cls, n, world, decomp=_capi.DECOMP_ROWS, devices=None, domain=(1.0, 2.0, 1.0, 2.0)):
        lib = _capi.load()
        h = C.c_void_p()
        a, b, c, d = domain
        devs = (C.c_int * len(devices))(*devices) if devices else None
        rc = lib.mi355cg_team_create_local(n, n, a, b, c, d, world, devs, len(devices) if devices else 0, decomp, C.byref(h))
        if rc == _capi.ERR_INVALID:
            raise ValueError(lib.mi355cg_last_error().decode())
        _capi.check(rc)
        return cls(h, n)

    @classmethod
    def rccl(cls, n, decomp=_capi.DECOMP_ROWS, device=None, group=None, domain=(1.0, 2.0, 1.0, 2.0)):
        lib = _capi.load()
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        ident = [None]
        if rank == 0:
            buf = (C.c_ubyte * 128)()
            _capi.check(lib.mi355cg_team_unique_id(buf))
            ident = [bytes(buf)]
        if world > 1:
            dist.broadcast_object_list(ident, src=0, group=group)
        idbuf = (C.c_ubyte * 128).from_buffer_copy(ident[0])
        h = C.c_void_p()
        a, b, c, d = domain
        dev = torch.cuda.current_device() if device is None else device
        rc = lib.mi355cg_team_create_rccl(n, n, a, b, c, d, world, rank, dev, idbuf, decomp, C.byref(h))
        if rc == _capi.ERR_INVALID:
            raise ValueError(lib.mi355cg_last_error().decode())
        _capi.check(rc)
        return cls(h, n)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mi355cg_team_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve(self, params: _capi.Params, callback=None, stop_flag: Optional[C.c_int] = None) -> _capi.Results:
        res = _capi.Results()
        cb = _capi.ITER_CB(lambda user, it, p, r, e: callback(it, p, r, e)) if callback else _capi.ITER_CB()
        sp = C.cast(C.pointer(stop_flag), C.c_void_p) if stop_flag is not None else None
        rc = self._lib.mi355cg_team_solve(self._h, C.byref(params), cb, None, sp, C.byref(res))
        if rc == _capi.ERR_INVALID:
            raise ValueError(self._lib.mi355cg_last_error().decode())
        _capi.check(rc)
        return res

    def vector(self, which: int, out: Optional[np.ndarray] = None) -> np.ndarray:
        """which: 0 x, 1 recursive residual, 2 right-hand side, 3 exact solution.  Global packed order; only the entries
        owned by this process's parts are written (all of them for a local team)."""
        if out is None:
            out = np.full(self.size, np.nan)
        _capi.check(self._lib.mi355cg_team_get_vector(self._h, which, out))
        return out

    def set_vector(self, which: int, v: np.ndarray):
        """which: 2 right-hand side, 3 exact solution; v in global packed order (every rank of an RCCL team passes the same vector)."""
        v = np.ascontiguousarray(v, dtype=np.float64)
        if v.size != self.size:
            raise ValueError(f"expected {self.size} entries, got {v.size}")
        _capi.check(self._lib.mi355cg_team_set_vector(self._h, which, v))

    def set_dtype(self, dtype: int):
        """_capi.F32_MIXED: the team's solves become fp64 iterative refinement around an fp32 inner CG (REL_2NORM only, row slabs
        only) -- BASELINE config 3 across GPUs; _capi.F64 switches back.  Collective: every rank makes the same call."""
        rc = self._lib.mi355cg_team_set_dtype(self._h, dtype)
        if rc == _capi.ERR_INVALID:
            raise ValueError(self._lib.mi355cg_last_error().decode())
        _capi.check(rc)

    def checksum(self, which: int):
        o = (C.c_double * 2)()
        _capi.check(self._lib.mi355cg_team_checksum(self._h, which, o))
        return o[0], o[1]

    def setup_on_device(self):
        _capi.check(self._lib.mi355cg_team_setup_on_device(self._h))

    def describe(self) -> dict:
        """What the next solve uses: {"transport", "records", "wait", "halo", "split", "ipc", "shared_device", "rccl_nranks", "rccl_lib"}."""
        buf = C.create_string_buffer(512)
        _capi.check(self._lib.mi355cg_team_describe(self._h, buf, len(buf)))
        out = dict(kv.split("=", 1) for kv in buf.value.decode().split())
        for k in ("split", "ipc", "shared_device", "rccl_nranks"):
            out[k] = int(out[k])
        return out

    def set_profiling(self, on: bool):
        _capi.check(self._lib.mi355cg_team_set_profiling(self._h, 1 if on else 0))

    def phase_times(self):
        k, c, w = C.c_double(), C.c_double(), C.c_double()
        _capi.check(self._lib.mi355cg_team_phase_times(self._h, C.byref(k), C.byref(c), C.byref(w)))
        return {"kernels_ms": k.value, "comm_ms": c.value, "wall_ms": w.value}


# ---------------------------------------------------------------------------------------------
def weak_scaling_n(n1: int, world: int) -> int:
    """Grid size whose unknown count is ~world x that of n1 (even)."""
    n = int(round(n1 * (world ** 0.5) / 2.0)) * 2
    return max(n, 6)
