"""Host logic of one kernel-recombination step (SOBER/_rchq.py:34-270), device agnostic.

The engine owns every *decision* of the reference's algorithm -- the halving loop, the grouping
of live positions into 2b sets, quirks Q1-Q6, the Caratheodory pivots -- and
delegates every *bulk computation* to an `ops` backend.  The product has exactly one backend,
`HipOps` (hand-written HIP kernels behind the C ABI); there is no CPU path.

The parity-critical pieces -- the PSD repair and randomised range finder of the Nystrom Gram (its `randn` draw is the
CPU generator's, so `torch.manual_seed` before the call reproduces the reference's subspace) and the null space of
each Caratheodory step -- run on the DEVICE (HipOps.nystrom_basis_device, the k_car_* / k_mc_* kernels) with the
reference's decisions and random draw.  The literal host routes the reference has them on
(`ker_svd_sparsify_host`, `car_host`: LAPACK) remain as the last rung: a symmetric Gram (mode "kernel"), a borderline
jitter ladder, sizes beyond the kernels, or a device whose multi-workgroup launches gave up (HipOps.car_mode).

Multi-GPU: the live position list is sharded in contiguous ranges, one per rank;
each level needs ONE all-reduce of the projected partial set sums (n x S doubles) and set masses
(S doubles); the survivor compaction is closed-form, so no candidate row ever moves.
"""
from __future__ import annotations

import os
import time
import warnings
from typing import Optional

import numpy as np
import torch

from . import _native as nat
from ._utils import SafeTensorOperator


# --------------------------------------------------------------------------- #
# communicators
# --------------------------------------------------------------------------- #
class SoloComm:
    rank, world = 0, 1

    def allreduce_sum(self, *tensors):
        return

    def allgather_counts(self, n: int):
        return [n]

    def allgather_rows(self, t: torch.Tensor, counts):
        return t

    def broadcast0(self, t: torch.Tensor):
        return t


class DistComm:
    """torch.distributed group (backend 'nccl' = RCCL over xGMI on the MI355X node; 'gloo' in the
    CPU tests).  Only small tensors travel: (n*S + S) doubles per level."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def allreduce_sum(self, *tensors):
        if len(tensors) == 1:
            self.dist.all_reduce(tensors[0], group=self.group)
            return
        flat = torch.cat([t.reshape(-1) for t in tensors])        # one message per level
        self.dist.all_reduce(flat, group=self.group)
        o = 0
        for t in tensors:
            t.copy_(flat[o:o + t.numel()].view_as(t))
            o += t.numel()

    def _dev(self):
        return torch.device("cuda", torch.cuda.current_device()) \
            if self.dist.get_backend(self.group) == "nccl" else torch.device("cpu")

    def allgather_counts(self, n: int):
        """One int per rank (a plain tensor all-gather: all_gather_object would pickle through the device)."""
        dev = self._dev()
        mine = torch.tensor([int(n)], dtype=torch.int64, device=dev)
        outs = [torch.empty(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
        self.dist.all_gather(outs, mine, group=self.group)
        return torch.cat(outs).tolist()

    def broadcast0(self, t: torch.Tensor):
        """Rank 0's tensor to everyone (the Nystrom basis: its randn draw must be shared)."""
        self.dist.broadcast(t, src=self.dist.get_global_rank(self.group, 0) if self.group else 0,
                            group=self.group)
        return t

    _RCCL = {}          # one RCCL communicator of our own per torch.distributed group
    _PEER = {}          # one direct-peer exchange (csrc/peer_reduce.hip) per group, sized for the largest message so far

    def native_allreduce(self, flat: torch.Tensor, device):
        """-> (address of a sober_allreduce_fn, communicator pointer, keep-alive) for the level executor's sharded
        loop.  Backend nccl: RCCL called from C on the stream (csrc/rccl_link.cpp).  Any other backend (gloo in the
        one-GPU tests): a callback into this group's all_reduce on the flat buffer -- correct, not fast."""
        import ctypes as C
        dist, group = self.dist, self.group
        key = id(group) if group is not None else 0
        peer_mode = os.environ.get("SOBER_PEER_ALLREDUCE", "1")     # "0": never; "force": also over a non-nccl group (tests)
        if (dist.get_backend(group) == "nccl" and peer_mode != "0") or peer_mode == "force":
            # the one-shot direct-peer all-reduce (every rank reads its peers' messages over xGMI and sums
            # in rank order: deterministic, one kernel) first; it checks itself on every rank when it is set up, and the
            # group falls back to RCCL together when the node does not support it (SOBER_PEER_ALLREDUCE=0: never tried)
            ent = DistComm._PEER.get(key)
            pc = ent[1] if (ent is not None and ent[0] is group) else None   # (id() of a destroyed group can come back: the entry holds the group itself)
            if pc is None or (pc is not False and pc.n_max < flat.numel()):
                if pc:
                    # a larger message: the region is rebuilt.  A peer's last kernel may still be reading this rank's slot
                    # (my stream's synchronisation does not cover ITS kernels) -- everybody leaves the old region first
                    dist.barrier(group=group)
                    torch.cuda.synchronize(device)
                    dist.barrier(group=group)
                    pc.close()
                why = "its set-up or self-check did not pass on every rank"
                try:
                    pc = nat.PeerComm(dist, group, device, max(int(flat.numel()), 1 << 15))
                    if pc.ok:
                        pc.self_check(dist, group)
                    if not pc.ok:
                        pc = False
                except Exception as e:                      # (a collective inside may have failed on this rank only)
                    pc, why = False, f"{type(e).__name__}: {e}"
                if pc is False:
                    warnings.warn("sober_amd: the direct-peer all-reduce is not used for this group (" + why + "); the level "
                                  "loop's all-reduce goes through RCCL / torch.distributed")
                DistComm._PEER[key] = (group, pc)
            if pc is not False:
                return pc.fn_ptr, pc.handle, pc
        if dist.get_backend(group) == "nccl":
            ent = DistComm._RCCL.get(key)
            rc = ent[1] if (ent is not None and ent[0] is group) else None
            if rc is None:
                try:
                    rc = nat.RcclComm(dist, group, device)
                except Exception as e:                     # (the same image on every rank: all of them land here)
                    import warnings
                    warnings.warn(f"sober_amd: RCCL could not be bound from C ({e}); the level loop's all-reduce goes "
                                  "through torch.distributed instead")
                    rc = False
                DistComm._RCCL[key] = (group, rc)
            if rc is not False:
                return rc.fn_ptr, rc.handle, rc

        def cb(_comm, _buf, _n, _stream):
            try:
                torch.cuda.current_stream(device).synchronize()
                dist.all_reduce(flat, group=group)
                return 0
            except Exception:                              # an exception must not unwind through the C caller
                return -1
        fn = nat.ALLREDUCE_FN(cb)
        return C.cast(fn, C.c_void_p), None, fn

    def allgather_rows(self, t: torch.Tensor, counts):
        """Concatenate per-rank row blocks (rank r contributes counts[r] rows) in rank order."""
        mx = max(counts)
        shape = (mx,) + tuple(t.shape[1:])
        pad = torch.zeros(shape, dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        bufs = [torch.empty_like(pad) for _ in range(self.world)]
        self.dist.all_gather(bufs, pad, group=self.group)
        return torch.cat([b[:c] for b, c in zip(bufs, counts)], 0)


# --------------------------------------------------------------------------- #
# host pieces of the algorithm
# --------------------------------------------------------------------------- #
HOST_LAPACK_THREADS = 8  # threads for the large host LAPACK sections (eigvalsh / svd_lowrank of the Nystrom Gram)


class host_lapack_threads:
    """Run ONE host LAPACK section with the thread count that is fastest for its size, then put the caller's
    setting back.  torch's default of one CPU thread per core is pathological for the small LAPACK problems of the
    fallback routes on a many-core GPU host (MI355X box, 128 threads: eigvalsh(500x500) 361 ms vs 8.7 ms with
    one thread, svd(100x200) 20 ms vs 1.3 ms).  Scoped on purpose: the surrounding BO loop's own CPU work (the GP
    fit) keeps whatever thread count its owner chose."""

    def __init__(self, size: int):
        self.n = 1 if size <= 1024 else HOST_LAPACK_THREADS

    def __enter__(self):
        self.old = torch.get_num_threads()
        if self.old != self.n:
            torch.set_num_threads(self.n)

    def __exit__(self, *exc):
        if self.old != self.n:
            torch.set_num_threads(self.old)


def ker_svd_sparsify_host(gram_host: torch.Tensor, s: int, tm: Optional[SafeTensorOperator] = None):
    """SOBER/_rchq.py:34-39 from `mat = kernel(pt, pt)` on: PSD repair, torch.svd_lowrank (its
    randn(M, s) comes from the global CPU generator), U = -_U.T."""
    tm = tm or SafeTensorOperator()
    with host_lapack_threads(gram_host.shape[0]):
        mat = tm.make_cov_psd(gram_host)
        _U, S, _ = torch.svd_lowrank(mat, q=s)
    return S, -1 * _U.T


def car_host(X: torch.Tensor, mu: torch.Tensor):
    """Tchernychova_Lyons_CAR, SOBER/_rchq.py:224-270, on host tensors: X (N', n'), mu (N',)
    (modified in place).  Null space from LAPACK's full SVD exactly like :231-234; the pivot loop
    is the C++ routine of csrc/car_host.cpp.  Returns (w_star, idx_star)."""
    dt = X.dtype
    X1 = torch.cat([torch.ones(X.size(0), 1, dtype=dt), X], dim=1)
    N, n = X1.shape
    with host_lapack_threads(N):
        _, _, V = torch.linalg.svd(X1.T)
    Phi = V[-(N - n):, :].T.contiguous()
    nat.car_pivot_host(Phi, mu)
    keep = mu > 0
    return mu[keep], torch.arange(N)[keep]


def second_elimination_host(Xp, obj_p, w_star, idx_star):
    """The extra null-vector elimination of the acquisition-guided branch, SOBER/_rchq.py:87-106 and
    :177-196 (host tensors): among the feasible recombinations move towards larger sum w * calc_obj
    and drop one more point."""
    dt = Xp.dtype
    Xp = torch.cat((Xp, torch.ones(1, len(idx_star), dtype=dt)), 0)
    with host_lapack_threads(Xp.shape[1]):
        _, _, w_null = torch.linalg.svd(Xp)
    w_null = w_null[-1]
    if torch.dot(obj_p, w_null) < 0:
        w_null = -w_null
    lm = len(w_star)
    plis = w_null > 0
    alpha = torch.zeros(lm, dtype=dt)
    alpha[plis] = w_star[plis] / w_null[plis]
    idx_sp = torch.arange(lm)[plis]
    idx_sp = idx_sp[torch.argmin(alpha[plis])]
    w_star = w_star - alpha[idx_sp] * w_null
    w_star[idx_sp] = 0.0
    keep = w_star > 0
    return w_star[keep], idx_star[keep]


def survivors_before(p: int, S: int, E: int, kept_prefix, n_keep: int, last_kept: bool) -> int:
    """Number of surviving list positions strictly below global position p after a level that kept
    the sets with kept_prefix[s] = #kept sets < s (closed-form compaction)."""
    ES = E * S
    if p <= ES:
        return (p // S) * n_keep + kept_prefix[p % S]
    return E * n_keep + ((p - ES) if last_kept else 0)


# --------------------------------------------------------------------------- #
# the engine
# --------------------------------------------------------------------------- #
class RecombinationEngine:
    def __init__(self, ops, comm=None, row_offset: int = 0):
        self.ops = ops
        self.comm = comm or SoloComm()
        self.row_offset = int(row_offset)      # global index of local candidate 0 (sharded pools)
        self.tm = SafeTensorOperator()
        self.trace = None                      # set to a dict to record per-level data (tests)
        self.timers = {}                       # phase -> seconds (host wall clock, accumulated)
        self.force_host_car = False            # True: LAPACK null space + C++ pivots on the host
        self.force_host_nystrom = False        # True: make_cov_psd + svd_lowrank entirely on the host
        self.basis_override = None             # a ready Nystrom basis (the replicated finish of a sharded run)
        # sharded runs: once the global list is this short, the live rows are gathered and every rank finishes the
        # remaining levels replicated -- a level's all-reduce (tens of us over xGMI) then costs more than the sharding
        # of a level kernel that runs 30 us on one GPU saves.  An estimate from one-GPU kernel times; 0 disables.
        self.cutover_R = int(os.environ.get("SOBER_CUTOVER_R", "32768"))
        # a one-rank group normally runs as an unsharded pool; this sends it through the sharded loop and its
        # collectives all the same (the RCCL path on a one-GPU box)
        self.force_sharded = bool(os.environ.get("SOBER_FORCE_SHARDED"))

    def _tick(self, name, t0):
        t1 = time.perf_counter()
        self.timers[name] = self.timers.get(name, 0.0) + (t1 - t0)
        return t1

    # -- Nystrom basis --------------------------------------------------------
    def nystrom_basis(self, plan, s: int, overlap=None, literal=False, early=None):
        """-> U (s, M).  Device route when the backend has one (HipOps.nystrom_basis_device); the
        literal host route (LAPACK) otherwise or when the device route declines.  `overlap`: callable
        that enqueues device work independent of U; the device route calls it (once) while the host
        still works on the basis.  `early`: the same for short device work whose result the host wants soon -- the
        device route calls it right behind the Cholesky probes.  `literal`: take the host route (it returns
        svd_lowrank's U itself, final rotation U_B included)."""
        if self.basis_override is not None:
            return self.basis_override
        t0 = time.perf_counter()
        dev_route = getattr(self.ops, "nystrom_basis_device", None)
        if dev_route is not None and not self.force_host_nystrom and not literal:
            res = dev_route(plan, s, self.tm.max_iter, overlap, early) if (overlap is not None or early is not None) \
                else dev_route(plan, s, self.tm.max_iter)
            if res is not None:
                U, gram = res
                self._tick("nystrom_device", t0)
                if self.trace is not None:
                    (gram_h, U_h) = self.ops.to_host(gram, U)
                    self.trace.update(gram=gram_h, U=U_h)
                return U
        gram = self.ops.gram(plan)
        (gram_h,) = self.ops.to_host(gram)
        t0 = self._tick("gram_device", t0)
        _, U = ker_svd_sparsify_host(gram_h, s, self.tm)
        self._tick("nystrom_host", t0)
        if self.trace is not None:
            self.trace.update(gram=gram_h.clone(), U=U.clone())
        return U

    # -- the halving loop -------------------------------------------------------
    def run(self, plan, mu: torch.Tensor, num_pts: int, obj: Optional[torch.Tensor] = None, live=None):
        """Mod_Tchernychova_Lyons, SOBER/_rchq.py:51-221.  `mu` (local weights, device, float64)
        is modified in place (Q3).  `obj` = -calc_obj(samp) for the local candidates (or None).
        `live`: the live list itself (int32, ascending) instead of arange(N)[mu != 0] -- the replicated finish of a
        sharded run, whose list carries on from the levels before it (a survivor whose weight is exactly 0 keeps its
        place there, as it does in an unsharded run after the first level).
        Returns (idx_star int64 global indices, w_star) on the device, identical on every rank."""
        ops, comm = self.ops, self.comm
        self.obj = obj
        self.obj_head = None
        if obj is not None:
            # quirk of the final level (:89): the reference indexes the GLOBAL obj with list positions,
            # so the first 2b entries of the pool's obj are needed by every rank
            head = obj[:2 * num_pts].clone()
            if comm.world > 1:
                cnt = comm.allgather_counts(int(head.numel()))
                head = comm.allgather_rows(head, cnt)[:2 * num_pts]
            self.obj_head = head
        n = num_pts - 1
        S = 2 * (n + 1)
        # the live list and the first level's set sums do not depend on the Nystrom basis: on the device
        # route they are enqueued while the host runs the small SVD of svd_lowrank
        state = {}
        # (the list itself is requested at once, behind whatever the plan has enqueued: its count is on the host long
        #  before the first level wants it, with no synchronisation behind the Nystrom chain)
        # (requested as soon as the Cholesky probes of the device Nystrom route are enqueued -- `early` below --: its
        #  count then reaches the host long before the first level wants it, and the host's share of it is hidden
        #  behind the probes instead of standing in front of the whole chain)
        can_start = getattr(ops, "nonzero_start", None) is not None and live is None
        pend = {}

        def start_list():
            if can_start and "p" not in pend:
                pend["p"] = ops.nonzero_start(mu)
        sharded = comm.world > 1 or (self.force_sharded and getattr(comm, "native_allreduce", None) is not None)

        def first_sums():
            if live is not None:
                idx_cur, count = live, int(live.numel())
            else:
                start_list()
                idx_cur, count = ops.nonzero_finish(pend["p"]) if "p" in pend else ops.nonzero_i32(mu)
            #                                                idx_story = arange(N)[mu != 0]  (:63-65)
            counts = comm.allgather_counts(count)
            bounds = [0]
            for c in counts:
                bounds.append(bounds[-1] + c)              # every rank's range of list positions, kept in closed form
            state.update(idx_cur=idx_cur, count=count, pos0=bounds[comm.rank], R=bounds[-1], bounds=bounds)
            # (a sharded pool at or below the cut-over goes straight to the replicated finish: no level runs here)
            cut = sharded and obj is None and self.cutover_R > 0 and state["R"] <= self.cutover_R
            if state["R"] > S and getattr(ops, "level_car", None) is not None and count > 0 and not cut:
                ops.level_moments(plan, idx_cur, state["pos0"], count, S, state["R"] // S, mu, phase=1, n=n)
                state["sums_ready"] = True

        def head_once():
            if not state:
                first_sums()

        # The device Nystrom route returns an orthonormal basis of svd_lowrank's subspace WITHOUT the final rotation
        # U_B.  That is invisible to a Caratheodory step whose null space comes from the bidiagonalisation's right
        # reflectors (the device kernels; also what MKL's gesdd returns), but not to every LAPACK: with another
        # vendor's gesdd/gesvd the null-space basis of [1 | X O]^T differs from that of [1 | X]^T.  So whenever the
        # Caratheodory steps of this run go to the host's LAPACK the literal route (U_B included) is taken and the
        # run reproduces the reference on the same machine whatever its LAPACK.
        n_fun = n + 1 + (1 if obj is not None else 0)
        car_on_host = self.force_host_car or getattr(ops, "car_supported", None) is None \
            or not ops.car_supported(S, n_fun) or (obj is not None and getattr(ops, "car_obj_device", None) is None)
        # (sharded runs keep the collectives in one fixed order on every rank: no overlap there, the ranks'
        # Nystrom routes may differ -- each draws its own randn -- and only rank 0's result is used)
        if car_on_host and not self.force_host_car and getattr(ops, "size_cliff", None) is not None \
                and getattr(ops, "car_supported", None) is not None and not ops.car_supported(S, n_fun):
            ops.size_cliff("car", f"batch = {n + 1}: a Caratheodory step on 2 x batch = {S} points with {n_fun} test functions is beyond "
                                  "the device kernels (registers of one compute unit: batch <= 100, of nine: batch <= 224, memory-resident: "
                                  "2 x batch <= 2048); every level's step runs on host LAPACK + the C++ pivots instead -- seconds per level")
        if comm.world > 1:                                  # (no hook on that route)
            start_list()
        U = self.nystrom_basis(plan, n, overlap=head_once if comm.world == 1 else None, literal=car_on_host,
                               early=start_list if comm.world == 1 else None)
        if comm.world > 1:                                  # rank 0's randn draw is the one that counts
            U = comm.broadcast0(ops.from_host(U.contiguous()) if U.device != ops.device else U.contiguous())
            if hasattr(plan, "_proj_src"):
                plan._proj_src = None                       # (the broadcast may have filled the very tensor a projection was enqueued for)
        ops.set_projection(plan, U)
        head_once()
        idx_cur, count, pos0, R = state["idx_cur"], state["count"], state["pos0"], state["R"]
        bounds = state["bounds"]
        sums_ready = state.get("sums_ready", False)
        idx_new = ops.empty_i32(count)
        levels = [] if self.trace is not None else None

        if self.trace is not None:
            self.trace["levels"] = levels
        if not sharded and obj is None and levels is None and not self.force_host_car and R > S \
                and getattr(ops, "level_loop", None) is not None and ops.car_supported(S, n + 1):
            # unsharded pool, on-chip Caratheodory step: the whole loop below runs inside the level executor
            t0 = time.perf_counter()
            idx_cur, idx_new, R = ops.level_loop(plan, idx_cur, idx_new, R, S, mu, sums_ready, self.row_offset)
            pos0, count, bounds, sums_ready = 0, R, [0, R], False
            self._tick("levels_device", t0)
        elif not sharded and obj is not None and levels is None and not self.force_host_car and R > S \
                and getattr(ops, "level_loop_obj", None) is not None:
            # acquisition-guided branch: the levels that certainly exist as one queued chain of the level executor (every
            # verdict read on the device); an irregular level, the last levels and the final one stay with the loop below
            t0 = time.perf_counter()
            res = ops.level_loop_obj(plan, idx_cur, idx_new, R, S, mu, obj, sums_ready)
            if res is not None and res[3] > 0:
                idx_cur, idx_new, R = res[:3]
                pos0, count, bounds, sums_ready = 0, R, [0, R], False
            self._tick("levels_device", t0)
        elif sharded and obj is None and levels is None and not self.force_host_car and R > S \
                and getattr(ops, "level_loop_sharded", None) is not None and ops.car_supported(S, n + 1) \
                and getattr(comm, "native_allreduce", None) is not None:
            # sharded pool: the levels run inside the level executor with the all-reduce issued from C; below
            # cutover_R live positions the rows are gathered once and every rank finishes replicated
            t0 = time.perf_counter()
            idx_cur, idx_new, bounds = ops.level_loop_sharded(plan, idx_cur, idx_new, bounds, S, mu, sums_ready, comm,
                                                              self.cutover_R)
            R, pos0, count, sums_ready = bounds[-1], bounds[comm.rank], bounds[comm.rank + 1] - bounds[comm.rank], False
            self._tick("levels_device", t0)
            if R > S:
                return self._finish_replicated(plan, idx_cur, bounds, mu, num_pts, U)
        while True:
            if R <= n + 1:                                  # :72-75
                return self._finish_small(mu)
            if R <= S:                                      # :77-114
                return self._finish_direct(plan, idx_cur, bounds, R, mu, levels)

            E = R // S
            r = R - E * S
            t0 = time.perf_counter()
            if sums_ready:                                  # first level: the set sums are already there
                Xtr, tot = ops.level_moments(plan, idx_cur, pos0, count, S, E, mu, phase=2)
                sums_ready = False
            else:
                Xtr, tot = ops.level_moments(plan, idx_cur, pos0, count, S, E, mu)
            if obj is None:
                flat = getattr(ops, "level_flat", None)    # Xtr and tot in one buffer: one message, no packing
                if flat is not None:
                    comm.allreduce_sum(flat(plan))
                else:
                    comm.allreduce_sum(Xtr, tot)
            else:                                           # one more "test function" row (:138-150,157-159)
                orow = self._obj_set_sums(obj, mu, idx_cur, pos0, count, S, E)
                comm.allreduce_sum(Xtr, tot, orow)
                Xtr = torch.cat([Xtr, orow.unsqueeze(0)], 0)
            if obj is None and not self.force_host_car and getattr(ops, "level_car", None) is not None \
                    and ops.car_supported(S, n + 1):
                # barycentres + on-chip Caratheodory step + flags to the host: one executor call (:151,166,173-175)
                res = ops.level_car(plan, S)
                if res is None:                             # the launches gave up and no device rung is left for this size
                    X_tmp, tot = ops.level_barycentres(plan)
                    keep_rank_d, w_star_d, keep_rank, n_keep = self._car(X_tmp, tot, R, E, r, levels, t0)
                    keep_np = keep_rank.numpy()
                else:
                    keep_rank_d, w_star_d, keep_np, n_keep = res
                    self._tick("levels_device", t0)
                    if levels is not None:
                        X_h, mu_h, w_h = ops.level_trace(plan)
                        levels.append(dict(kind="level", R=R, E=E, r=r, X_tmp=X_h, tot_weights=mu_h,
                                           idx_star=torch.from_numpy(np.flatnonzero(keep_np >= 0)),
                                           w_star=w_h[:n_keep].clone()))
            else:
                X_tmp = ops.barycentres(Xtr, tot)           # :151,166
                keep_rank_d, w_star_d, keep_rank, n_keep = self._car(X_tmp, tot, R, E, r, levels, t0)  # :173-175
                keep_np = keep_rank.numpy()
            kept = keep_np >= 0
            last_kept = bool(kept[S - 1])
            kept_prefix = [0] + np.cumsum(kept).tolist()
            R_new = E * n_keep + (r if last_kept else 0)
            if R_new >= R:
                raise RuntimeError(
                    "recombination made no progress (the Caratheodory step cancelled nothing, "
                    "SOBER/_rchq.py:241-242); the reference would loop forever here")
            bounds = [survivors_before(b_, S, E, kept_prefix, n_keep, last_kept) for b_ in bounds]
            new_pos0, new_end = bounds[comm.rank], bounds[comm.rank + 1]
            if count > 0:
                ops.level_update(idx_cur, pos0, count, S, E, keep_rank_d, w_star_d, tot, n_keep, mu,
                                 idx_new, new_pos0)                                           # :198-221
            idx_cur, idx_new = idx_new, idx_cur
            pos0, count, R = new_pos0, new_end - new_pos0, R_new

    def _obj_set_sums(self, obj, mu, idx_cur, pos0, count, S, E):
        """X_for_obj of SOBER/_rchq.py:138-146 plus the leftover addition to the last set (:157-163)
        for the local positions: sum of obj*mu per set, leftovers counted in set p mod S AND in set S-1
        (same quirk Q1 as the kernel rows).  Reshape-sum: fixed order, no atomics."""
        dev = mu.device
        if count == 0:
            return torch.zeros(S, dtype=torch.float64, device=dev)
        native = getattr(self.ops, "obj_set_sums", None)
        if native is not None and obj.dtype == torch.float64 and obj.is_contiguous():
            return native(obj, mu, idx_cur, pos0, count, S, E)       # one launch (csrc/misc.hip: k_obj_set_sums)
        c = idx_cur[:count].long()
        v = obj[c] * mu[c]
        e_first = pos0 // S
        e_total = (pos0 + count + S - 1) // S - e_first
        dense = torch.zeros(e_total * S, dtype=torch.float64, device=dev)
        off = pos0 - e_first * S
        dense[off:off + count] = v
        out = dense.view(e_total, S).sum(0)
        n_left = pos0 + count - max(pos0, E * S)
        if n_left > 0:
            out[S - 1] += v[count - n_left:].sum()
        return out

    def _finish_replicated(self, plan, idx_cur, bounds, mu, num_pts, U):
        """The cut-over of a sharded run: every rank contributes the rows, weights and global indices of its live
        positions (one all-gather each, in global list order), builds the plan of that short pool with the SAME
        Nystrom basis and finishes the remaining levels as an unsharded run; the caller's shard of the weights is
        then brought to the final state (Q3)."""
        ops, comm = self.ops, self.comm
        counts = [bounds[r + 1] - bounds[r] for r in range(comm.world)]
        loc = idx_cur[:counts[comm.rank]].long()
        X_all = comm.allgather_rows(plan.X_cand_raw[loc].to(torch.float64).contiguous(), counts)
        w_all = comm.allgather_rows(mu[loc].contiguous(), counts)
        gid = comm.allgather_rows((loc + self.row_offset).contiguous(), counts)
        plan2 = ops.build_plan(plan.spec, plan.mode, plan.X_nys_raw, X_all)
        sub = RecombinationEngine(ops, SoloComm(), row_offset=0)
        sub.basis_override = U
        sub.force_host_nystrom, sub.force_host_car = self.force_host_nystrom, self.force_host_car
        mu2 = w_all.clone().contiguous()
        live2 = torch.arange(mu2.numel(), dtype=torch.int32, device=mu2.device)   # the gathered list, every entry of it
        idx2, w2 = sub.run(plan2, mu2, num_pts, live=live2)
        for k, v in sub.timers.items():
            self.timers[k] = self.timers.get(k, 0.0) + v
        idx_glob = gid[idx2]
        mu[loc] = 0.0                                                      # my cancelled rows
        mine = (idx_glob >= self.row_offset) & (idx_glob < self.row_offset + mu.numel())
        mu[idx_glob[mine] - self.row_offset] = w2[mine]
        return idx_glob, w2

    # -- one Caratheodory step ---------------------------------------------------
    def _car(self, X_dev, mu_dev, R, E, r, levels, t0, kind="level"):
        """Tchernychova_Lyons_CAR on (X_dev (N', n'), mu_dev (N')).  On-chip HIP kernel when the
        size fits (batch <= 100), host LAPACK + C++ pivots otherwise.  Returns device keep_rank /
        w_star, the host copy of keep_rank and n_keep."""
        ops = self.ops
        Np, n1 = X_dev.shape[0], X_dev.shape[1] + 1
        use_obj = getattr(self, "obj", None) is not None
        on_device = getattr(ops, "car_supported", None) is not None and ops.car_supported(Np, n1) \
            and not self.force_host_car
        if on_device and use_obj and getattr(ops, "car_obj_device", None) is not None:
            # acquisition-guided branch: Caratheodory step with the objective as one more function + the extra
            # elimination, both on the device (:84-106, :173-196)
            res = ops.car_obj_device(X_dev, mu_dev, None if kind == "level" else self.obj_head)
            if res is not None:
                keep_rank_d, w_star_d, keep_rank, n_keep, (kr1_h, w1_d, n1k) = res
                self._tick("levels_device", t0)
                if levels is not None:
                    X_h, mu_h, w1_h = ops.to_host(X_dev, mu_dev, w1_d)
                    levels.append(dict(kind=kind, R=R, E=E, r=r, X_tmp=X_h, tot_weights=mu_h,
                                       idx_star=torch.nonzero(kr1_h >= 0).flatten(), w_star=w1_h[:n1k].clone()))
                return keep_rank_d, w_star_d, keep_rank, n_keep
            on_device = False
        elif use_obj:
            on_device = False
        if on_device:
            # (a step whose launches gave up comes back redone on the single-workgroup kernels, or as None: host route)
            res = ops.car_device_checked(X_dev, mu_dev, also=(X_dev, mu_dev) if levels is not None else ())
            if res is not None:
                keep_rank_d, w_star_d, keep_rank, n_keep, extra = res
                self._tick("levels_device", t0)
                if levels is not None:
                    (w_h,) = ops.to_host(w_star_d)
                    levels.append(dict(kind=kind, R=R, E=E, r=r, X_tmp=extra[0], tot_weights=extra[1],
                                       idx_star=torch.nonzero(keep_rank >= 0).flatten(), w_star=w_h[:n_keep].clone()))
                return keep_rank_d, w_star_d, keep_rank, n_keep
        X_h, mu_h = ops.to_host(X_dev, mu_dev)
        t0 = self._tick("levels_device", t0)
        w_star, idx_star = car_host(X_h, mu_h.clone())
        if levels is not None:                              # (trace = the Caratheodory step's own output)
            levels.append(dict(kind=kind, R=R, E=E, r=r, X_tmp=X_h.clone(), tot_weights=mu_h.clone(),
                               idx_star=idx_star.clone(), w_star=w_star.clone()))
        if use_obj:                                         # :177-196 / :87-106
            nfun = X_h.shape[1] - 1
            obj_p = X_h[idx_star, nfun] if kind == "level" else self.obj_head.cpu()[idx_star]
            w_star, idx_star = second_elimination_host(X_h[idx_star, :nfun].T.contiguous(), obj_p, w_star, idx_star)
        self._tick("car_host", t0)
        n_keep = int(idx_star.numel())
        keep_rank = torch.full((Np,), -1, dtype=torch.int32)
        keep_rank[idx_star] = torch.arange(n_keep, dtype=torch.int32)
        return ops.from_host(keep_rank), ops.from_host(w_star), keep_rank, n_keep

    # -- terminal branches ------------------------------------------------------
    def _gather_result(self, idx_local, w_local, counts=None):
        comm = self.comm
        idx_glob = idx_local.to(torch.int64) + self.row_offset
        if comm.world == 1:
            return idx_glob, w_local
        if counts is None:
            counts = comm.allgather_counts(int(idx_glob.numel()))
        return comm.allgather_rows(idx_glob, counts), comm.allgather_rows(w_local, counts)

    def _finish_small(self, mu):
        idx_star = torch.nonzero(mu > 0).flatten()          # arange(len(mu))[mu > 0]  (:73)
        return self._gather_result(idx_star, mu[idx_star])

    def _finish_direct(self, plan, idx_cur, bounds, R, mu, levels):
        ops, comm = self.ops, self.comm
        n = plan.n
        counts = [bounds[r + 1] - bounds[r] for r in range(comm.world)]    # known everywhere: no exchange
        count = counts[comm.rank]
        obj = getattr(self, "obj", None)
        if comm.world == 1 and obj is None and levels is None and not self.force_host_car \
                and getattr(ops, "level_final", None) is not None:
            t0 = time.perf_counter()
            res = ops.level_final(plan, idx_cur, R, 2 * (n + 1), mu, self.row_offset)
            if res is not None:
                self._tick("levels_device", t0)
                return res
        nf = n + (1 if obj is not None else 0)
        if count > 0:
            X_loc = ops.direct_columns(plan, idx_cur, count)           # rows of (U @ K).T  (:78)
            mu_loc = mu[idx_cur[:count].long()]
            if obj is not None:                                        # :80-81
                X_loc = torch.cat([X_loc, obj[idx_cur[:count].long()].unsqueeze(1)], 1).contiguous()
        else:
            X_loc = torch.zeros(0, nf, dtype=torch.float64, device=ops.device)
            mu_loc = torch.zeros(0, dtype=torch.float64, device=ops.device)
        X_all = comm.allgather_rows(X_loc, counts)
        mu_all = comm.allgather_rows(mu_loc, counts)
        t0 = time.perf_counter()
        keep_rank_d, w_star_d, keep_rank, n_keep = self._car(X_all, mu_all, R, 0, 0, levels, t0, kind="final")  # :84-85
        idx_star = torch.nonzero(keep_rank >= 0).flatten()
        lo = bounds[comm.rank]
        mine = (idx_star >= lo) & (idx_star < lo + count)
        # how many selected points each rank owns (the Caratheodory step is replicated: every rank knows)
        sel_counts = [int(((idx_star >= bounds[r]) & (idx_star < bounds[r + 1])).sum()) for r in range(comm.world)]
        sel = (idx_star[mine] - lo).to(torch.int32)
        ranks = keep_rank[idx_star[mine]].long()
        mu.zero_()                                                     # mu[:] = 0  (:109)
        w_mine = w_star_d[ops.from_host(ranks)] if ranks.numel() else w_star_d[:0]
        out_idx = ops.scatter_weights(idx_cur, ops.from_host(sel), w_mine, mu)   # mu[idx_story] = w_star
        return self._gather_result(out_idx, w_mine, sel_counts)
