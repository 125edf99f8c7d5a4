"""Device operations of the recombination engine on the HIP library (the only product backend).

`HipOps` owns no algorithmic decision: it allocates workspaces with torch and enqueues the C-ABI
kernels on the current stream.  The level logic lives in `_engine.py`."""
from __future__ import annotations

import os
import warnings

import torch

from . import _native as nat
from ._ops_levels import _LevelOps
from ._ops_nystrom import _NystromOps
from ._ops_plan import Plan, _PlanOps      # noqa: F401  (Plan is part of this module's interface)


class HipOps(_PlanOps, _NystromOps, _LevelOps):
    """One backend object per device: the stages live in `_ops_plan` (the plan and its caches), `_ops_nystrom` (the Nystrom
    job) and `_ops_levels` (level loops, final level, Caratheodory step); here: construction, workspaces, event pairs for the
    bench, the live list and the host copies."""
    name = "hip"

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise nat.SoberHipError("sober_amd needs a HIP device (torch device 'cuda'); got "
                                    f"{self.device}.  There is no CPU fallback.")
        nat.load()
        self._pin = {}
        self.prof = None        # list -> (start_event, stop_event, kernel entries, launches), one entry per launch of the
        #                         level kernel (main and leftover): the events ride in the dispatch itself
        #                         (sober_set_launch_events), so they hold the kernel's own duration -- rocprofv3's number --
        #                         and cost the stream nothing
        self.use_mfma = True    # False: VALU-only level kernel (direct differences), kept for A/B runs
        # False (or SOBER_NO_QUEUE in the environment): every level sized by the host after a synchronisation
        self.queue_levels = os.environ.get("SOBER_NO_QUEUE") is None
        # False: every level's set sums evaluated; True: levels 1 .. D of a pool without leftovers gathered and scaled from
        # level 0's element-class sums (csrc/level_class.hip; the library's SOBER_LEVEL_NO_CLASSES switch gives depth 0 too)
        self.level_classes = True
        # acquisition-guided branch: the second elimination's direction from csrc/null_vector.hip (False: from a second
        # Caratheodory step on the b + 1 survivors, rounds 2-5 -- which stays for the sizes that kernel does not cover)
        self.obj_null_kernel = True
        # ... and that branch's levels as one queued chain of the level executor (False: a visit to Python and a read-back per
        # level, rounds 2-6a -- what still takes the levels the chain does not complete)
        self.queue_obj_levels = True
        # matrix-core path, unweighted: no scaled copy of the pool (the final level scales its <= 2 b rows itself); False: the
        # copy of rounds 1-5 (A/B, and what the VALU level kernel and the weighted mode still use)
        self.lazy_scaled_pool = True
        self.last_levels = None  # {"R": live positions per level, "derived": D} of the last level_loop call
        # the jitter ladder's probes on eight workgroups per rung (False: one workgroup each -- where the process stays
        # after a rung's workgroups lost each other)
        self._probe_mc = torch.cuda.get_device_properties(self.device).multi_processor_count >= 256   # (8 XCDs x 32 CUs: unpartitioned)
        # the rung of the Caratheodory step on this device: CAR_DEFAULT (launches whose workgroups wait for partner
        # workgroups: fused / multi-CU), CAR_SAFE after one of them gave up (launches without such waits: the single-workgroup
        # kernels up to batch 100, csrc/car_big.hip's launch per dependency beyond), CAR_HOST beyond the device kernels
        # (host LAPACK + C++ pivots) -- SOBER/_rchq.py:224-270 never fails, so neither may this
        self.car_mode = nat.CAR_DEFAULT

    def size_cliff(self, which: str, message: str):
        """A size beyond the compiled device kernels sends a phase to the host: correct, much slower -- said ONCE per
        backend and phase, naming the limit (the reference takes any N_nys and batch: SOBER/_rchq.py:34-39, :224-270)."""
        seen = self.__dict__.setdefault("_cliffs_seen", set())
        if which not in seen:
            seen.add(which)
            import warnings
            warnings.warn("sober_amd: " + message, RuntimeWarning, stacklevel=3)

    # ------------------------------------------------------------------ levels
    def prof_reserve(self, n_pairs: int):
        """Pre-create event pairs for `self.prof` (so that a timed region does not pay for their creation)."""
        pool = getattr(self, "_ev_pool", None)
        if pool is None:
            pool = self._ev_pool = []
        st = torch.cuda.current_stream(self.device)
        for _ in range(n_pairs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); e1.record(st)                    # creates the hipEvent_t; the executor re-records it
            pool.append((e0, e1))

    def _prof_pair(self):
        pool = getattr(self, "_ev_pool", None)
        if not pool:
            self.prof_reserve(1)
            pool = self._ev_pool
        return pool.pop()

    def _buf(self, p, name, numel):
        t = p.ws.get(name)
        if t is None or t.numel() < numel:
            t = torch.empty(numel, dtype=torch.float64, device=self.device)
            p.ws[name] = t
        return t

    def _buf_u8(self, p, name, nbytes):
        t = p.ws.get(name)
        if t is None or t.numel() < nbytes:
            t = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=self.device)
            p.ws[name] = t
        return t

    # ------------------------------------------------------------------ plumbing
    def nonzero_start(self, mu):
        """Enqueue the live list idx_story = arange(N)[mu != 0] (SOBER/_rchq.py:63-65) NOW, as a compaction whose
        count stays on the device and travels to pinned memory behind it; `nonzero_finish` waits for that copy only.
        (torch.nonzero synchronises to size its output: called where the list is first needed, that wait sat behind the
        whole Nystrom chain and the first level could not be enqueued until it was over.)"""
        N = mu.numel()
        key = ("nz", N)
        bufs = self._pin.get(key)
        if bufs is None:
            ws = torch.zeros(nat.nonzero_ws_bytes(N), dtype=torch.uint8, device=self.device)   # (zero once: every call leaves it zero)
            bufs = self._pin[key] = (ws, torch.zeros(1, dtype=torch.int64, device=self.device),
                                     torch.zeros(1, dtype=torch.int64).pin_memory())
        ws, cnt, pin = bufs
        out = torch.empty(N, dtype=torch.int32, device=self.device)
        nat.nonzero_i32(mu, out, cnt, ws)
        pin.copy_(cnt, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return out, pin, ev

    def nonzero_finish(self, pending):
        out, pin, ev = pending
        ev.synchronize()
        return out, int(pin[0])

    def nonzero_i32(self, mu):
        nz = torch.nonzero(mu != 0).flatten()
        out = torch.empty(max(nz.numel(), 1), dtype=torch.int32, device=self.device)
        if nz.numel():
            nat.i64_to_i32(nz, out)
        return out, int(nz.numel())

    def empty_i32(self, n):
        return torch.empty(max(n, 1), dtype=torch.int32, device=self.device)

    def to_host(self, *tensors, before_sync=None):
        """Device tensors -> host copies through pinned staging buffers, one synchronisation.  `before_sync`
        (optional callable) runs after the copies are enqueued: whatever it enqueues overlaps with the host
        work that follows this call (the wait is on the copies only)."""
        outs = []
        for i, t in enumerate(tensors):
            key = (i, t.dtype, tuple(t.shape))
            buf = self._pin.get(key)
            if buf is None:
                buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                self._pin[key] = buf
            buf.copy_(t, non_blocking=True)
            outs.append(buf)
        if before_sync is None:
            torch.cuda.current_stream(self.device).synchronize()
        else:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            before_sync()
            ev.synchronize()
        return [o.clone() for o in outs]

    def from_host(self, t, dtype=None):
        return t.to(self.device, dtype=dtype, non_blocking=False)


class MatrixKernelOps(HipOps):
    """Backend for kernels whose matrix against the Nystrom points is RESIDENT in HBM: Kmat = K(X_cand, X_nys),
    (N, M) candidate-major -- 400 MB at N = 100k, M = 500, a corner of the MI355X's 288 GB.  It is built once
    per step and every level is then the HBM-bound gather-sum `sober_level_gather`; nothing else differs
    from the fused path (Nystrom basis, projection on the matrix cores, Caratheodory steps, weight update).

    Two kinds of kernel arrive here:
      * a native provider with `materialise(X_cand, X_nys) -> (N, M)` (BASQ's g-space kernel, whose
        exp(C_h) - 1 is non-linear in the posterior covariance, so the sum-first shortcut of the fused
        path does not apply -- SURVEY.md 8 row f4);
      * ANY callable `kernel(x, y)` following the protocol of SOBER/_rchq.py:9,20 (2-D `x`, 2-D or 3-D
        `y`): its matrix is evaluated chunk-wise by the caller's own torch code on the device, once."""

    CHUNK_ROWS = 1 << 15       # candidates per call of a foreign callable

    def build_plan(self, kernel_fn, mode, X_nys, X_cand, pool_owner=None) -> Plan:
        p = Plan()
        p.kernel_fn, p.mode = kernel_fn, "matrix"
        p.X_nys, p.X_cand = X_nys, X_cand
        p.M = p.Mtot = X_nys.shape[0]
        p.weighted, p.T, p.P, p.wmul, p.da = False, None, None, None, -1
        p.ws = {}
        N = X_cand.shape[0]
        if hasattr(kernel_fn, "materialise"):
            p.Kmat = kernel_fn.materialise(X_cand, X_nys)
        else:
            p.Kmat = torch.empty(N, p.M, dtype=torch.float64, device=self.device)
            for lo in range(0, N, self.CHUNK_ROWS):
                hi = min(N, lo + self.CHUNK_ROWS)
                p.Kmat[lo:hi] = kernel_fn(X_nys, X_cand[lo:hi]).to(torch.float64).T
        if tuple(p.Kmat.shape) != (N, p.M) or p.Kmat.dtype != torch.float64 or p.Kmat.stride(1) != 1:
            raise nat.SoberHipError(f"kernel matrix must be ({N}, {p.M}) float64 row-major, got "
                                    f"{tuple(p.Kmat.shape)} {p.Kmat.dtype}")
        return p

    def gram(self, p):
        if hasattr(p.kernel_fn, "materialise"):
            return p.kernel_fn.materialise(p.X_nys, p.X_nys, gram=True)
        return p.kernel_fn(p.X_nys, p.X_nys).to(torch.float64).contiguous()

    def set_projection(self, p, U):
        p.P = U.to(self.device, torch.float64).contiguous()
        p.n = p.P.shape[0]

    def direct_columns(self, p, idx, count):
        Kc = p.Kmat[idx[:count].long()].contiguous()                      # (count, M)
        Xtr = torch.empty(p.n, count, dtype=torch.float64, device=self.device)
        nat.dgemm(p.P, Kc, Xtr, transb=True)
        out = torch.empty(count, p.n, dtype=torch.float64, device=self.device)
        nat.barycentres(Xtr, p.n, count, None, out)
        return out


CallableKernelOps = MatrixKernelOps
