"""Device operations of the recombination engine on the HIP library (the only product backend).

`HipOps` owns no algorithmic decision: it allocates workspaces with torch and enqueues the C-ABI
kernels on the current stream.  The level logic lives in `_engine.py`."""
from __future__ import annotations

import os
import warnings

import torch

from . import _native as nat
from . import _rng
from ._kernel import KernelSpec, PointSet, posterior_mean, prepare_points, woodbury


class Plan:
    """Per-step device state shared by all levels.  The POOL's side of it (scaled / augmented candidates, the pool's
    posterior mean) is prepared at first use: the Nystrom chain needs the row table only, so the host enqueues that
    chain first and prepares the pool while the GPU is busy with it (build_plan: `_pool_prep`)."""
    _POOL_FIELDS = ("cand", "cand_aug", "rows_aug", "wmul")

    def __getattr__(self, name):                                  # (only reached when the attribute is not set yet)
        if name in Plan._POOL_FIELDS:
            prep = self.__dict__.get("_pool_prep")
            if prep is not None:
                self.__dict__["_pool_prep"] = None
                prep()
                if name in self.__dict__:
                    return self.__dict__[name]
        raise AttributeError(name)


class HipOps:
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
        # the jitter ladder's probes on eight workgroups per rung (False: one workgroup each -- where the process stays
        # after a rung's workgroups lost each other)
        self._probe_mc = torch.cuda.get_device_properties(self.device).multi_processor_count >= 256   # (8 XCDs x 32 CUs: unpartitioned)
        # the rung of the Caratheodory step on this device: CAR_DEFAULT (launches whose workgroups wait for partner
        # workgroups: fused / multi-CU), CAR_SAFE after one of them gave up (single-workgroup kernels, batch <= 100),
        # CAR_HOST beyond those (host LAPACK + C++ pivots) -- SOBER/_rchq.py:224-270 never fails, so neither may this
        self.car_mode = nat.CAR_DEFAULT

    # ------------------------------------------------------------------ plan
    def build_plan(self, spec: KernelSpec, mode: str, X_nys, X_cand, pool_owner=None) -> Plan:
        """`pool_owner`: the caller's own tensor object behind X_cand (recombination() hands the plan a detached view,
        a fresh Python object per call): what the caches of `_packed_pool` / `_pool_mean` hold their weak reference to."""
        p = Plan()
        pool_owner = X_cand if pool_owner is None else pool_owner
        p.spec, p.mode = spec, mode
        p.X_nys_raw, p.X_cand_raw = X_nys, X_cand                 # (the replicated finish of a sharded run rebuilds a plan)
        p.kind = nat.KIND_BY_NAME[spec.kind]
        p.M = X_nys.shape[0]
        dev = self.device
        corrected = mode != "kernel"
        p.n_obs = spec.X_obs.shape[0] if corrected else 0
        p.weighted = mode == "weighted_predictive_covariance"
        p.mean_nys = None
        p.T = None
        Xn64 = X_nys.to(torch.float64)
        native_rows = (p.kind != nat.KIND_TANIMOTO and not p.weighted and Xn64.stride(-1) == 1
                       and (not corrected or (spec.X_obs.dtype == torch.float64 and spec.X_obs.stride(-1) == 1
                                              and spec.S_cache.dtype == torch.float64 and spec.S_cache.stride(-1) == 1
                                              # sober_plan_rows forms W = S S^T for a SQUARE root (n_obs x n_obs); gpytorch's
                                              # Lanczos root beyond max_cholesky_size is n_obs x k: the Python route (woodbury)
                                              and spec.S_cache.dim() == 2
                                              and spec.S_cache.shape[0] == spec.S_cache.shape[1] == spec.X_obs.shape[0])))
        if native_rows:
            # the row table, Kall, W, T and the Gram matrix behind ONE native call (csrc/nystrom_exec.cpp: sober_plan_rows)
            f64 = torch.float64
            p.Mtot = p.M + p.n_obs
            dt = nat.padded_dim(X_nys.shape[1], generic=True)
            rows = torch.empty(p.Mtot, dt, dtype=f64, device=dev)
            G = torch.empty(p.M, p.M, dtype=f64, device=dev)
            Kall = W = T = None
            if corrected:
                Kall = torch.empty(p.Mtot, p.M, dtype=f64, device=dev)
                W = torch.empty(p.n_obs, p.n_obs, dtype=f64, device=dev)
                T = torch.empty(p.M, p.n_obs, dtype=f64, device=dev)
            nat.plan_rows(p.kind, Xn64, spec.X_obs if corrected else None, spec.lengthscale, spec.outputscale,
                          spec.S_cache if corrected else None, rows, Kall, W, T, G)
            p.rows = PointSet(rows, None, dt)
            p.T, p._gram = T, G
            if corrected:
                p.Kall = Kall
            stacked = None                                        # (built by pool_prep, where the augmented rows want it)
        else:
            stacked = torch.cat([Xn64, spec.X_obs], 0) if corrected else X_nys
            p.rows = prepare_points(spec, stacked)                # [X_nys; X_obs]
            p.Mtot = len(p.rows)
            if p.weighted:
                p.mean_nys = posterior_mean(spec, p.rows.rows(0, p.M))
            if corrected:
                # T = KxX @ W with KxX = k(X_nys, X_obs) (SOBER/_gp.py:293,295)
                Kall = torch.empty(p.Mtot, p.M, dtype=torch.float64, device=dev)
                nys = p.rows.rows(0, p.M)
                nat.pairwise(p.kind, p.rows.data, p.rows.norm, nys.data, nys.norm, None, p.M, p.rows.dt,
                             spec.outputscale, Kall)
                p.Kall = Kall
                W = woodbury(spec)
                p.T = torch.empty(p.M, p.n_obs, dtype=torch.float64, device=dev)
                nat.dgemm(Kall[p.M:], W, p.T, transa=True)        # K_Xn^T W == KxX W (k symmetric)
        p.da = nat.aug_dim(X_nys.shape[1]) if (p.kind != nat.KIND_TANIMOTO and self.use_mfma) else -1

        def pool_prep():
            # everything that reads the candidate pool -- first touched by the first level's set sums, which the engine
            # enqueues behind the Nystrom chain: this host work then runs beside that chain, not in front of it
            p.cand = self._packed_pool(spec, X_cand, pool_owner)
            p.wmul = self._pool_mean(spec, p.cand, X_cand, pool_owner) if p.weighted else None   # mu_y of SOBER/_kernel.py:41
            # matrix-core level kernel: augmented copies of the row table and the pool
            if p.da > 0:
                st = stacked if stacked is not None else (torch.cat([Xn64, spec.X_obs], 0) if corrected else Xn64)
                center = Xn64.mean(0).contiguous()                # any shift works; this one keeps |x~| small
                p.rows_aug = torch.empty(p.Mtot, p.da, dtype=torch.float64, device=dev)
                p.cand_aug = torch.empty(X_cand.shape[0], p.da, dtype=torch.float64, device=dev)
                nat.augment_points(st.to(torch.float64).contiguous(), spec.lengthscale, center, 0, p.rows_aug)
                Xc = X_cand if (X_cand.dtype == torch.float64 and X_cand.stride(-1) == 1) else \
                    X_cand.to(torch.float64).contiguous()
                nat.augment_points(Xc, spec.lengthscale, center, 1, p.cand_aug)
            else:
                p.rows_aug = p.cand_aug = None
        p._pool_prep = pool_prep
        p.P = None
        p.ws = {}
        return p

    def _packed_pool(self, spec, X_cand, owner):
        """prepare_points for the candidate pool.  A fingerprint pool (Tanimoto) arrives as an FP64 0/1 matrix --
        4 GB at 250k x 2048 -- and a dataset prior without pruning hands over the SAME tensor object at every BO
        iteration (SOBER/_sampler.py:351-382): its bit-packed form (64x smaller) is kept across calls.  Only the
        packed words are held: the pool itself is referenced WEAKLY, through the caller's own tensor object `owner`
        (a pruned prior builds a fresh tensor per iteration -- that entry then simply misses and the old words are
        dropped; nothing pins the caller's 4 GB).  A hit needs the same owner object (alive), the same memory, layout
        and in-place version counter (a detached view shares its base's counter); writes that bypass the counter
        (`.data`, DLPack, foreign kernels) are the caller's to announce with `clear_cache()`."""
        if spec.kind != "tanimoto":
            return prepare_points(spec, X_cand)
        if not self._shares_storage(X_cand, owner):
            # a converted copy of the caller's pool (CPU / float32 / bool pool: recombination() made X_cand itself): its
            # memory is freed after the call and the allocator may hand the same block to the next copy -- pointer, layout
            # and a fresh version counter would then match a pool the caller has modified since.  Never kept.
            self._pool_cache = None
            return prepare_points(spec, X_cand)
        import weakref
        key = (X_cand.data_ptr(), tuple(X_cand.shape), tuple(X_cand.stride()), X_cand.dtype, X_cand._version,
               owner.data_ptr(), owner._version)
        hit = getattr(self, "_pool_cache", None)
        if hit is not None and hit[0] == key and hit[1]() is owner:
            return hit[2]
        self._pool_cache = None                                   # (a miss frees the previous pool's words first)
        pts = prepare_points(spec, X_cand)
        self._pool_cache = (key, weakref.ref(owner), pts)
        return pts

    def _pool_mean(self, spec, cand, X_cand, owner):
        """The posterior mean over the pool (the per-candidate factor of the weighted kernel, SOBER/_kernel.py:41): a
        kernel-matvec over all N candidates, 0.37 ms at 250k x 2048 bits.  Kept while BOTH the pool (same tensor object,
        layout, version: the packed-pool cache's rule) and the model's snapshot (the same KernelSpec tensors, unmodified)
        come back -- several batches drawn from one fitted model; a live gpytorch model is re-read per call and always
        misses."""
        if not self._shares_storage(X_cand, owner):                # (a converted copy: see _packed_pool)
            self._mean_cache = None
            return posterior_mean(spec, cand)
        key = (X_cand.data_ptr(), tuple(X_cand.shape), tuple(X_cand.stride()), X_cand.dtype, X_cand._version,
               owner.data_ptr(), owner._version,
               id(spec.alpha), spec.alpha._version, id(spec.X_obs), spec.X_obs._version,
               id(spec.lengthscale), spec.lengthscale._version, spec.kind, float(spec.outputscale), float(spec.mean_const))
        hit = getattr(self, "_mean_cache", None)
        if hit is not None and hit[0] == key and hit[1]() is owner and hit[2]() is spec.alpha and hit[3]() is spec.X_obs:
            return hit[4]
        import weakref
        self._mean_cache = None
        out = posterior_mean(spec, cand)
        self._mean_cache = (key, weakref.ref(owner), weakref.ref(spec.alpha), weakref.ref(spec.X_obs), out)
        return out

    @staticmethod
    def _shares_storage(X_cand, owner):
        """True iff X_cand IS the caller's pool memory (the owner tensor itself or a view of it from its first element):
        only then does the owner's liveness pin the block and its version counter see every in-place write."""
        return (isinstance(owner, torch.Tensor) and owner.device == X_cand.device and owner.dtype == X_cand.dtype
                and owner.data_ptr() == X_cand.data_ptr())

    def size_cliff(self, which: str, message: str):
        """A size beyond the compiled device kernels sends a phase to the host: correct, much slower -- said ONCE per
        backend and phase, naming the limit (the reference takes any N_nys and batch: SOBER/_rchq.py:34-39, :224-270)."""
        seen = self.__dict__.setdefault("_cliffs_seen", set())
        if which not in seen:
            seen.add(which)
            import warnings
            warnings.warn("sober_amd: " + message, RuntimeWarning, stacklevel=3)

    def clear_cache(self):
        """Drop what is kept across calls (the packed fingerprint pool, the pool's posterior mean): after writing into a
        pool or a KernelSpec tensor through a path that bypasses torch's version counter."""
        self._pool_cache = None
        self._mean_cache = None

    def gram(self, p: Plan):
        """kernel(pt, pt) of SOBER/_rchq.py:35 for the plan's mode."""
        dev = self.device
        G = getattr(p, "_gram", None)
        if G is not None:                                         # (sober_plan_rows computed it with the row table)
            p._gram = None
            return G
        if p.T is None:
            nys = p.rows
            G = torch.empty(p.M, p.M, dtype=torch.float64, device=dev)
            nat.pairwise(p.kind, nys.data, nys.norm, nys.data, nys.norm, None, p.M, nys.dt,
                         p.spec.outputscale, G)
            return G
        G = p.Kall[:p.M].clone()
        nat.dgemm(p.T, p.Kall[p.M:], G, alpha=-1.0, beta=1.0)     # Kxy - (KxX W) KXy
        if p.weighted:
            G = p.mean_nys.unsqueeze(1) * G * p.mean_nys.unsqueeze(0)
        return G

    def set_projection(self, p: Plan, U):
        """P = [U diag(mean), -(U diag(mean)) T]: phi(x) = P k([X_nys; X_obs], x) is the vector of
        Nystrom test functions U @ C(X_nys, x) (SOBER/_rchq.py:78,148,156) with the posterior
        correction of SOBER/_gp.py:295 folded in (it is linear)."""
        if getattr(p, "_proj_src", None) is U:              # already enqueued for this very basis (nystrom_basis_device)
            return
        p._proj_src = U
        U = U.to(self.device, torch.float64).contiguous()
        p.n = U.shape[0]
        P = torch.empty(p.n, p.Mtot if p.T is not None else p.M, dtype=torch.float64, device=self.device)
        nat.projection(U, p.mean_nys if p.weighted else None, p.T, P)
        p.P = P

    # ------------------------------------------------------------------ Nystrom basis on the device
    NITER = 2                    # torch.svd_lowrank's default number of power iterations

    def nystrom_basis_device(self, p: Plan, s: int, max_iter: int = 10, overlap=None, early=None):
        """ker_svd_sparsify (SOBER/_rchq.py:34-39) with the N_nys x N_nys work on the GPU, no host decision inside, the
        whole chain behind ONE native call (csrc/nystrom_exec.cpp: sober_nystrom_basis):
          make_cov_psd: |cov| and the symmetry test in one kernel; every rung of the jitter ladder probed by one
                        launch; the first positive definite rung (or the diagonal fallback) applied on the device with
                        the reference's own sequence of additions;
          svd_lowrank : randn from the CPU generator (same draw as the reference: the host steps the Mersenne twister,
                        Box-Muller runs on the device); range finder with MFMA GEMMs + CholeskyQR on the matrix cores;
                        the result is an orthonormal basis of the reference's subspace;
          projection  : P = [U, -U T] enqueued before the flags are waited for.
        Returns (U (s, M) on the device, the Gram matrix) or None when the literal host path must decide (exactly
        symmetric Gram, sizes beyond the kernels, a borderline ladder, an ill-conditioned range finder); the CPU
        generator is then back where it was.

        WHY NO SMALL SVD.  torch/_lowrank.py goes on with B = Q^H A, its SVD and U = Q U_B.  U_B is a q x q
        ORTHOGONAL matrix, and nothing downstream can see it: the Caratheodory step (SOBER/_rchq.py:224-270) takes the
        null space of A = [1 | X]^T from the right Householder reflectors of A's bidiagonalisation (csrc/car.hip), and
        those depend on A only through its first row (the ones) and A^T A -- both unchanged when the remaining rows,
        i.e. the Nystrom test functions U k(X_nys, .), are mixed by an orthogonal matrix.  Same kept sets, same weights
        (tests/test_car_algorithm.py::test_car_invariant_under_orthogonal_mixing).  So any orthonormal basis of
        range(Q) serves, Q^T itself does.  (The literal host route still computes U_B.)"""
        dev, M = self.device, p.M
        if M > nat.nystrom_max_n() or s > 256 or s >= M:
            if s < M:                                             # (s >= M is the reference's own degenerate case, not a size limit)
                self.size_cliff("nystrom", f"N_nys = {M}, batch = {s + 1}: beyond the device Nystrom route (N_nys <= "
                                           f"{nat.nystrom_max_n()}, batch <= 257); make_cov_psd and svd_lowrank run on host LAPACK "
                                           "instead -- about 5-10x the device route's time for this phase")
            self.gram(p)
            return None
        G = self.gram(p)
        n_r, niter = max_iter + 1, self.NITER
        n_orth2 = 2 * (1 + 2 * niter)
        if getattr(p, "ws", None) is None:
            p.ws = {}
        # (the chain's buffers and its job live with the backend, not with the plan: a plan is built per step, and the
        #  pinned flag block alone costs more to allocate than the call it serves)
        key = ("nys", M, s, n_r)
        st = self._pin.get(key)
        if st is None:
            f64 = torch.float64
            nbytes = nat.nystrom_flags_bytes(n_r, niter)
            st = self._pin[key] = {
                "job": nat.NystromJob(), "C": torch.empty(M, M, dtype=f64, device=dev),
                "Y0": torch.empty(M, s, dtype=f64, device=dev), "Y1": torch.empty(M, s, dtype=f64, device=dev),
                "Gm": torch.empty(s, s, dtype=f64, device=dev),
                "xinv": torch.empty(((s + 31) // 32) * 1024, dtype=f64, device=dev),
                "flags": torch.empty(nbytes, dtype=torch.uint8, device=dev),
                "h_flags": torch.empty(nbytes, dtype=torch.uint8, pin_memory=True),
                "Ut": torch.empty(s, M, dtype=f64, device=dev),
            }
            skey = ("shifts", n_r)
            if skey not in self._pin:
                self._pin[skey] = torch.tensor([1e-5 * (2 ** k - 1) for k in range(n_r)], dtype=f64, device=dev)
            j = st["job"]
            j.M, j.s, j.n_rungs, j.niter = M, s, n_r, niter
            j.shifts, j.C = self._pin[skey].data_ptr(), st["C"].data_ptr()
            j.Y[0], j.Y[1], j.Gm, j.xinv = st["Y0"].data_ptr(), st["Y1"].data_ptr(), st["Gm"].data_ptr(), st["xinv"].data_ptr()
            j.flags_block, j.flags_bytes, j.h_flags_block = st["flags"].data_ptr(), nbytes, st["h_flags"].data_ptr()
            j.Ut = st["Ut"].data_ptr()
        j = st["job"]
        work = self._buf(p, "chol_work", n_r * M * M)
        j.G, j.chol_work = G.data_ptr(), work.data_ptr()
        # (eight workgroups per rung from a few panels on: 0.49 -> 0.16 ms at M = 500; a rung whose workgroups lost each
        #  other reports PROBE_NO_VERDICT and the step goes to the host route, this process then stays with one each)
        # (opt-in, SOBER_NYSTROM_SKIP=1: the range finder's intermediate CholeskyQR passes dropped behind the diagonal fallback
        #  with a mild spread -- faster, a subspace error of ~4e-11 instead of ~1e-15: csrc/nystrom_exec.cpp has the trade)
        j.skip_passes = 1 if os.environ.get("SOBER_NYSTROM_SKIP") else 0
        j.probe_mc = 1 if (M >= self.PROBE_MC_MIN and n_r <= 16 and self._probe_mc and M <= nat.chol_max_n()) else 0
        if j.probe_mc:
            pws = self._buf_u8(p, "chol_mc_ws", nat.cholesky_probe_mc_ws_bytes(M, n_r))
            j.probe_ws, j.probe_ws_bytes = pws.data_ptr(), pws.numel()
        elif M > nat.chol_max_n():                           # (the panel-by-panel probes: one inverted diagonal block per rung)
            pws = self._buf_u8(p, "chol_cb_ws", n_r * 8192)
            j.probe_ws, j.probe_ws_bytes = pws.data_ptr(), pws.numel()
        # the projection rides in the same call when the plan is a real one (it is simply redone should the flags
        # send the step to the host route)
        proj = hasattr(p, "weighted")
        if proj:
            T = p.T
            P = torch.empty(s, p.Mtot if T is not None else M, dtype=torch.float64, device=dev)
            j.T, j.n_obs = nat._ptr(T), (T.shape[1] if T is not None else 0)
            j.mean_nys, j.P = (p.mean_nys.data_ptr() if p.weighted else None), P.data_ptr()
        else:
            j.P = None
        # svd_lowrank's randn comes from the CPU generator (it is the next consumer of the generator in the
        # reference too: make_cov_psd draws nothing); should the host route have to decide, the generator is put back
        stream = torch.cuda.current_stream(dev)
        nat.nystrom_basis(j, 1, stream.cuda_stream)         # the probes run while the host steps its generator
        if early is not None:
            early()                                         # (short device work whose result the host wants soon)
        rng_state = torch.get_rng_state()
        R = _rng.device_randn(M, s, dev)
        j.R = R.data_ptr()
        nat.nystrom_basis(j, 2, stream.cuda_stream)
        ev = torch.cuda.Event()
        ev.record(stream)
        Ut = st["Ut"]
        if proj:
            p.P, p.n, p._proj_src = P, s, Ut
        if overlap is not None:
            overlap()                                       # (device work independent of U, behind the chain)
        ev.synchronize()
        hb = st["h_flags"]
        n8 = 8 * (n_r + 1 + n_orth2 + 1)
        f64s, i32s = hb[:n8].view(torch.float64), hb[n8:].view(torch.int32)
        piv_h, pivs_rf = f64s[:n_r + 1], f64s[n_r + 1:n_r + 1 + n_orth2]      # (then one double: passes skipped, 1.0 / 0.0)
        flags_h, infos_rf = i32s[:2 + n_r], i32s[2 + n_r:]
        if any(int(v) == nat.PROBE_NO_VERDICT for v in flags_h[2:]):
            self._probe_mc = False
            warnings.warn("sober_amd: the multi-CU Cholesky probe lost contact between its workgroups; "
                          "falling back to one workgroup per rung")
            torch.set_rng_state(rng_state)
            return None
        # a second CholeskyQR pass works on a nearly orthonormal block: its pivots must be ~1; a single pass is accepted
        # while min pivot / max diagonal of its Gram matrix (~ cond^-2) stays above ORTH1_MIN_RATIO
        last = 2 * niter
        if os.environ.get("SOBER_NYSTROM_DEBUG"):
            print("nystrom: pivot ratios of the range finder's blocks", [float(pivs_rf[k + 1]) for k in range(0, n_orth2, 2)],
                  "intermediate passes skipped:", float(f64s[n_r + 1 + n_orth2]))
        single = [k for k in range(0, n_orth2, 2) if k // 2 != last]
        rank_lost = bool((infos_rf != 0).any()) or float(pivs_rf[2 * last + 1]) < 0.5 \
            or any(not (float(pivs_rf[k + 1]) >= self.ORTH1_MIN_RATIO) for k in single)
        if int(flags_h[0]) == 0 or rank_lost or self.ladder_borderline(flags_h[2:], piv_h[:n_r], float(piv_h[n_r])):
            torch.set_rng_state(rng_state)                 # the host route draws the same randn again
            return None
        warnings.warn("Estimated covariance matrix was not positive semi-definite. Conveting...")
        return Ut, G

    # is_psd (SOBER/_utils.py:117-129) = LAPACK's Cholesky succeeds AND linalg.eig >= 0.  k_chol's verdict on a rung
    # can only differ from that where the rung is numerically singular: its smallest pivot (the failing one, <= 0,
    # for a rejected rung) within LADDER_GUARD x the largest diagonal entry of zero.  Only the two deciding rungs
    # matter (the first accepted one and the rejected one in front of it); a borderline ladder goes to the host's
    # LAPACK, as the host twin does (sober_amd/_utils.py:make_cov_psd).
    LADDER_GUARD = 1e-9
    PROBE_MC_MIN = 160         # Gram matrices from this size on are probed by eight workgroups per rung

    @classmethod
    def ladder_borderline(cls, info, min_pivot, dmax) -> bool:
        thr = cls.LADDER_GUARD * max(dmax, 0.0)
        ok = [int(v) == 0 for v in info]
        piv = [float(v) for v in min_pivot]
        if not (dmax == dmax) or any(v != v for v in piv):
            return True
        k = ok.index(True) if any(ok) else len(ok)
        if k < len(ok) and piv[k] < thr:                    # accepted, but numerically singular
            return True
        if k > 0 and piv[k - 1] > -thr:                     # rejected by a hair
            return True
        return False

    def _orth(self, Y, infos, pivs, slot, passes: int = 2):
        """The CholeskyQR building block of the range finder, by itself (csrc/nystrom_exec.cpp runs the same three
        calls per pass; this form serves the kernels' own tests).
        CholeskyQR2: orthonormal basis of range(Y) with the flag of Householder QR.
        passes=1 (the intermediate blocks of the power iteration): the same subspace to the same accuracy
        (the first triangular solve decides it), orthonormal only to cond(Y)^2 eps -- which is all the next
        product A Q needs; pivs[slot + 1] then holds min pivot / max diagonal of the Gram matrix (~ cond^-2)."""
        q = Y.shape[1]
        for it in range(passes):
            Gm = torch.empty(q, q, dtype=torch.float64, device=self.device)
            nat.dgemm(Y, Y, Gm, transa=True)
            # one-workgroup blocked Cholesky, in place; it leaves the inverted 32 x 32 diagonal blocks behind,
            # which turn Q = Y R^-1 into block-to-block MFMA work (sober_trsm_blocks) -- and, for a single pass,
            # min pivot / max diagonal of the Gram matrix in pivs[slot + 1]
            Lc = Gm
            xinv = torch.empty(((q + 31) // 32) * 1024, dtype=torch.float64, device=self.device)
            nat.cholesky_inv(Lc, 0.0, infos[slot + it:slot + it + 1], pivs[slot + it:slot + it + 1], xinv,
                             ratio=pivs[slot + 1:slot + 2] if passes == 1 else None)
            Q = torch.empty_like(Y)
            nat.trsm_blocks(Y, Lc, xinv, Q)                     # Q = Y R^-1
            Y = Q
        return Y

    # single-pass CholeskyQR is accepted for an intermediate block while min pivot / max diagonal of its Gram
    # matrix (~ cond(Y)^-2) stays above this: orthonormality then holds to ~1e-6 and the subspace to eps cond(Y)
    ORTH1_MIN_RATIO = 1e-10

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

    def _job(self, p: Plan, S: int, n: int = None) -> nat.LevelJob:
        """The level executor's job (include/sober_hip.h: sober_level_job): per-step constants and
        workspaces, built once per plan; only the position range changes from level to level.  The
        projection P may still be missing (set sums of the first level): it is attached when it exists."""
        n = p.n if n is None else n
        job = p.ws.get("job")
        if job is not None and job.S == S and job.n == n:
            if p.P is not None and job.P != p.P.data_ptr():
                if p.P.stride(0) != p.Mtot or p.P.stride(1) != 1:
                    p.P = p.P.contiguous()
                job.P = p.P.data_ptr()
            job.car_mode = min(self.car_mode, nat.CAR_SAFE)
            return job
        dev, f64 = self.device, torch.float64
        job = nat.LevelJob()
        Kmat = getattr(p, "Kmat", None)
        if Kmat is not None:
            job.variant, job.kind = nat.LEVEL_GATHER, 0
            job.cand, job.kmat_ld, job.n_rows = Kmat.data_ptr(), Kmat.stride(0), Kmat.shape[1]
            job.outputscale = 1.0
        elif p.da > 0:
            job.variant, job.kind = nat.LEVEL_MFMA, p.kind
            job.rows, job.cand, job.dim, job.n_rows = p.rows_aug.data_ptr(), p.cand_aug.data_ptr(), p.da, p.Mtot
            job.outputscale = float(p.spec.outputscale)
        else:
            # fingerprints of 512 .. 2048 bits: popcount(x & y) on the FP4 matrix cores (level_reduce_tani.hip)
            tani = p.kind == nat.KIND_TANIMOTO and self.use_mfma and bool(nat.load().sober_level_reduce_tani_supported(p.rows.dt))
            job.variant, job.kind = (nat.LEVEL_TANI if tani else nat.LEVEL_VALU), p.kind
            job.rows, job.rows_norm = p.rows.data.data_ptr(), nat._ptr(p.rows.norm)
            job.cand, job.cand_norm = p.cand.data.data_ptr(), nat._ptr(p.cand.norm)
            job.dim, job.n_rows = p.rows.dt, p.Mtot
            job.outputscale = float(p.spec.outputscale)
        job.wmul = nat._ptr(p.wmul)
        if p.P is not None and (p.P.stride(0) != p.Mtot or p.P.stride(1) != 1):
            p.P = p.P.contiguous()
        job.S, job.n, job.P = S, n, nat._ptr(p.P)
        nr, mc, xs = job.n_rows, nat.LEVEL_MAX_CHUNKS, nat.LEVEL_XS
        w = p.ws
        w["partG"], w["partTot"] = torch.empty(mc * nr * S, dtype=f64, device=dev), torch.empty(mc * S, dtype=f64, device=dev)
        w["extraG"], w["extraTot"] = torch.empty(mc * nr * xs, dtype=f64, device=dev), torch.empty(mc * xs, dtype=f64, device=dev)
        w["G"] = torch.empty(nr, S, dtype=f64, device=dev)
        w["XT"] = torch.empty(n * S + S, dtype=f64, device=dev)      # Xtr and tot back to back: one all-reduce message
        w["Xtr"], w["tot"] = w["XT"][:n * S].view(n, S), w["XT"][n * S:]
        w["X_tmp"] = torch.empty(S, n, dtype=f64, device=dev)
        w["keep_rank"] = torch.empty(S + 1, dtype=torch.int32, device=dev)
        w["w_star"], w["mu_out"] = torch.empty(S, dtype=f64, device=dev), torch.empty(S, dtype=f64, device=dev)
        w["h_flags"] = torch.empty(S + 1, dtype=torch.int32, pin_memory=True)
        w["h_flags_np"] = w["h_flags"].numpy()
        if job.variant in (nat.LEVEL_MFMA, nat.LEVEL_TANI) and self.queue_levels:
            # queued levels (csrc/level_exec.cpp): live positions per level, on the device and in pinned memory
            w["dR"] = torch.zeros(nat.LEVEL_QUEUE + 1, dtype=torch.int64, device=dev)
            w["h_dR"] = torch.zeros(nat.LEVEL_QUEUE + 1, dtype=torch.int64, pin_memory=True)
            job.dR, job.h_dR = w["dR"].data_ptr(), w["h_dR"].data_ptr()
        for k in ("partG", "partTot", "extraG", "extraTot", "G", "Xtr", "tot", "X_tmp", "keep_rank", "w_star", "mu_out",
                  "h_flags"):
            setattr(job, k, w[k].data_ptr())
        if nat.car_supported(S, n + 1):
            nbytes = nat.car_ws_bytes(S, n + 1)
            w["car_ws"] = torch.empty(max(nbytes // 8, 1), dtype=f64, device=dev)
            job.car_ws, job.car_ws_bytes = w["car_ws"].data_ptr(), nbytes
        job.car_mode = min(self.car_mode, nat.CAR_SAFE)
        w["job"] = job
        return job

    def _car_downgrade(self, mode: int, why: str):
        """Remember that the Caratheodory launches which depend on partner workgroups gave up on this device (another
        stream or process shares it, or it is partitioned): the following steps start on the rung that worked."""
        if mode > self.car_mode:
            self.car_mode = mode
            warnings.warn("sober_amd: a multi-workgroup Caratheodory launch gave up waiting for its partner workgroups "
                          f"({why}); this and the following steps use "
                          + ("the single-workgroup kernels" if mode == nat.CAR_SAFE else "the host LAPACK route")
                          + " for that step -- same result, slower")

    def level_moments(self, p: Plan, idx, pos0, count, S, E, mu, phase: int = 0, n: int = None):
        """Partial (n, S) projected set sums and (S,) set masses over the local list positions
        [pos0, pos0+count) of a level with E full elements (SOBER/_rchq.py:116-164 minus the
        division).  Q1: leftovers (p >= E*S) are summed into set p mod S AND into set S-1; only
        the latter reaches `tot`.  One call of the level executor (csrc/level_exec.cpp): level_reduce,
        the leftover launch, sum_partials and the projection P G.  The returned tensors are the plan's
        workspaces: they are overwritten by the next level.
        phase 1: only the set sums (they do not depend on the Nystrom basis; `n` = its size, for the
        workspaces); phase 2: only the projection of the sums left by a phase-1 call."""
        job = self._job(p, S, n)
        job.phase = phase
        if count <= 0 and phase != 2:                       # a rank without live positions contributes zeros
            p.ws["XT"].zero_(); p.ws["G"].zero_()
            return p.ws["Xtr"], p.ws["tot"]
        if phase == 2:
            nat.level_moments(job, nat._stream(p.ws["G"]))
            return p.ws["Xtr"], p.ws["tot"]
        nat._req(idx, torch.int32, "idx"); nat._req(mu, torch.float64, "mu")
        job.idx, job.pos0, job.count, job.E, job.mu = idx.data_ptr(), pos0, count, E, mu.data_ptr()
        pair = None
        if self.prof is not None:
            n_left = max(pos0 + count - max(pos0, E * S), 0)
            pair, pair2 = self._prof_pair(), (self._prof_pair() if n_left > 0 else None)
            job.ev[0], job.ev[1] = pair[0].cuda_event, pair[1].cuda_event
            job.ev[2], job.ev[3] = (pair2[0].cuda_event, pair2[1].cuda_event) if pair2 else (None, None)
        nat.level_moments(job, nat._stream(mu))
        if pair is not None:
            self.prof.append((pair[0], pair[1], int(count * job.n_rows), 1))
            if pair2 is not None:
                self.prof.append((pair2[0], pair2[1], int(n_left * job.n_rows), 1))
            for k in range(4):
                job.ev[k] = None
        return p.ws["Xtr"], p.ws["tot"]

    def level_loop(self, p: Plan, idx_cur, idx_new, R: int, S: int, mu, sums_ready: bool, row_offset: int = 0):
        """The whole halving loop of an unsharded pool while R > S (SOBER/_rchq.py:116-221) in ONE call of the
        level executor (csrc/level_exec.cpp: sober_level_loop) -- no trip through Python between a level's
        verdict and the next level's launches.  Returns (idx_cur, idx_new, R) for the terminal branch."""
        import math
        job = self._job(p, S)
        nat._req(mu, torch.float64, "mu"); nat._req(idx_cur, torch.int32, "idx"); nat._req(idx_new, torch.int32, "idx")
        job.mu = mu.data_ptr()
        events = pairs = None
        if self.prof is not None:
            n_max = min(nat.MAX_LEVELS, int(math.log2(max(R / S, 1.0))) + 3)
            pairs = [(self._prof_pair(), self._prof_pair()) for _ in range(n_max)]      # (main, leftover) per level
            events = [None] * (4 * nat.MAX_LEVELS)
            for l, (a, b) in enumerate(pairs):
                events[4 * l:4 * l + 4] = [a[0].cuda_event, a[1].cuda_event, b[0].cuda_event, b[1].cuda_event]
        # the final direct level rides in the same call when the loop ends on one (no visit to Python between the
        # loop's synchronisation and that level's launches); level_final() then finds its result waiting
        fin = None
        p.ws.pop("final_done", None)
        if not p.weighted and getattr(p, "Kmat", None) is None and self.car_mode != nat.CAR_HOST and job.car_ws:
            fin = self._final_job(p, S, mu, row_offset)
        level_R, R_final, in_b, gave_up = nat.level_loop(job, R, idx_cur, idx_new, sums_ready, events, nat._stream(mu), fin)
        if fin is not None and fin.done:
            n_keep = int(p.ws["h_flags_np"][S])
            if n_keep >= 0:
                lst = idx_new if in_b else idx_cur
                p.ws["final_done"] = (lst.data_ptr(), R_final, row_offset, p._fin_out[0][:n_keep], p._fin_out[1][:n_keep])
        if job.car_mode > min(self.car_mode, nat.CAR_SAFE):
            self._car_downgrade(nat.CAR_SAFE, "level loop")
        if gave_up:                                          # beyond the single-workgroup kernels: the host route is next
            self._car_downgrade(nat.CAR_HOST, "level loop")
        if pairs is not None:
            # which pairs a launch carried: job.ev_used (a queued level that the chain did not reach still launched --
            # and left at once: counted as a launch without entries)
            for l, (a, b) in enumerate(pairs):
                Rl = level_R[l] if l < len(level_R) else 0
                both = not ((job.ev_used[1] >> l) & 1)       # (the leftover workgroups rode in the main launch)
                for which, pr, ent in ((0, a, Rl + (Rl % S if both else 0)), (1, b, Rl % S)):
                    if (job.ev_used[which] >> l) & 1 and not (l == 0 and sums_ready):
                        self.prof.append((pr[0], pr[1], int(ent * job.n_rows), 1))
                    else:
                        self._ev_pool.append(pr)
        return (idx_new, idx_cur, R_final) if in_b else (idx_cur, idx_new, R_final)

    def level_loop_sharded(self, p: Plan, idx_cur, idx_new, bounds, S: int, mu, sums_ready: bool, comm, R_stop: int):
        """The halving loop of a ROW-SHARDED pool in one call of the level executor (sober_level_loop_sharded): per
        level one all-reduce of the flat (n S + S) buffer on the stream -- RCCL when the group's backend is nccl, the
        group's own all_reduce through a callback otherwise (the one-GPU tests) -- and no Python in between.
        Returns (idx_cur, idx_new, bounds) when the global list is down to max(S, R_stop)."""
        job = self._job(p, S)
        nat._req(mu, torch.float64, "mu"); nat._req(idx_cur, torch.int32, "idx"); nat._req(idx_new, torch.int32, "idx")
        job.mu = mu.data_ptr()
        fn_ptr, comm_ptr, keep = comm.native_allreduce(p.ws["XT"], self.device)
        _, new_bounds, in_b = nat.level_loop_sharded(job, comm.rank, comm.world, bounds, idx_cur, idx_new, sums_ready,
                                                     fn_ptr, comm_ptr, R_stop, nat._stream(mu))
        del keep
        if job.car_mode > min(self.car_mode, nat.CAR_SAFE):
            self._car_downgrade(nat.CAR_SAFE, "sharded level loop")
        return (idx_new, idx_cur, new_bounds) if in_b else (idx_cur, idx_new, new_bounds)

    def _final_job(self, p: Plan, S: int, mu, row_offset: int = 0):
        """The arguments of sober_level_final as a struct (the buffers live with the plan)."""
        dev = self.device
        f = nat.FinalJob()
        f.rows_sc, f.rows_norm = p.rows.data.data_ptr(), nat._ptr(p.rows.norm)
        f.cand_sc, f.cand_norm = p.cand.data.data_ptr(), nat._ptr(p.cand.norm)
        f.dt, f.done, f.N, f.row_offset = p.rows.dt, 0, mu.numel(), row_offset
        p._fin_out = (torch.empty(S, dtype=torch.int64, device=dev), torch.empty(S, dtype=torch.float64, device=dev))
        f.K, f.mu_live = self._buf(p, "K_final", p.Mtot * S).data_ptr(), self._buf(p, "mu_live", S).data_ptr()
        f.out_idx, f.out_w = p._fin_out[0].data_ptr(), p._fin_out[1].data_ptr()
        return f

    def level_final(self, p: Plan, idx_cur, R: int, S: int, mu, row_offset: int):
        """The final direct level of an unsharded pool (n + 1 < R <= S, SOBER/_rchq.py:77-114) without leaving the
        device: one executor call, one synchronisation (for the number of survivors).  -> (idx int64, w) or None
        when this plan needs the step-by-step route (weighted mode, resident kernel matrix, size)."""
        if p.weighted or getattr(p, "Kmat", None) is not None or not nat.car_supported(R, p.n + 1) \
                or self.car_mode == nat.CAR_HOST or (self.car_mode == nat.CAR_SAFE and not nat.car_safe_supported(R, p.n + 1)):
            return None
        done = p.ws.pop("final_done", None)
        if done is not None:                                 # (level_loop ran it already: mu holds the result)
            if done[:3] != (idx_cur.data_ptr(), R, row_offset):
                raise nat.SoberHipError("level_final: the final level that rode in level_loop was another one")
            return done[3], done[4]
        job = self._job(p, S)
        if not job.car_ws:
            return None
        dev, f64 = self.device, torch.float64
        job.mu = mu.data_ptr()
        K = self._buf(p, "K_final", job.n_rows * S)
        mu_live = self._buf(p, "mu_live", S)
        out_idx = torch.empty(S, dtype=torch.int64, device=dev)
        out_w = torch.empty(S, dtype=f64, device=dev)
        st = torch.cuda.current_stream(dev)
        nat.level_final(job, p.rows.data, p.rows.norm, p.cand.data, p.cand.norm, p.rows.dt, idx_cur, R, mu.numel(),
                        row_offset, K, mu_live, out_idx, out_w, st.cuda_stream)
        st.synchronize()
        n_keep = int(p.ws["h_flags_np"][S])
        if n_keep < 0:
            # the step gave up; the weights are untouched (sober_final_commit): once more on the next rung, or the
            # engine's step-by-step route with the host's LAPACK
            if job.car_mode == nat.CAR_DEFAULT and nat.car_safe_supported(R, p.n + 1):
                self._car_downgrade(nat.CAR_SAFE, "final level")
                return self.level_final(p, idx_cur, R, S, mu, row_offset)
            self._car_downgrade(nat.CAR_HOST, "final level")
            return None
        return out_idx[:n_keep], out_w[:n_keep]

    def level_flat(self, p: Plan):
        """The projected set sums and the set masses of the last `level_moments` as ONE flat tensor (n*S + S)."""
        return p.ws["XT"]

    def level_car(self, p: Plan, S: int):
        """Barycentres (SOBER/_rchq.py:151,166) + the on-chip Caratheodory step (:173-175) on the plan's Xtr / tot,
        then keep_rank and n_keep to the host -- one executor call, one stream synchronisation.
        Returns (keep_rank_d int32 (S,), w_star_d (S,), keep_rank host numpy int32 (S,), n_keep)."""
        job = self._job(p, S)
        st = torch.cuda.current_stream(self.device)
        nat.level_car(job, st.cuda_stream)
        st.synchronize()
        flags = p.ws["h_flags_np"]
        if int(flags[S]) < 0:
            # the launches gave up: the step alone again on the single-workgroup kernels; None = the size is beyond
            # them, the engine takes the barycentres (still in the plan's workspace) to the host route
            if not nat.level_car_retry(job, st.cuda_stream):
                self._car_downgrade(nat.CAR_HOST, "level")
                return None
            self._car_downgrade(nat.CAR_SAFE, "level")
        return p.ws["keep_rank"][:S], p.ws["w_star"], flags[:S].copy(), int(flags[S])

    def level_barycentres(self, p: Plan):
        """The barycentres and set masses the last `level_car` worked on (device)."""
        return p.ws["X_tmp"], p.ws["tot"]

    def level_trace(self, p: Plan):
        """Host copies of the last level's barycentres, set masses and kept weights (test traces)."""
        return self.to_host(p.ws["X_tmp"], p.ws["tot"], p.ws["w_star"])

    def direct_columns(self, p: Plan, idx, count):
        """(count, n) rows U @ kernel(pt_nys, samp[idx]) of the final direct level
        (SOBER/_rchq.py:78)."""
        dev = self.device
        K = torch.empty(p.Mtot, count, dtype=torch.float64, device=dev)
        nat.pairwise(p.kind, p.rows.data, p.rows.norm, p.cand.data, p.cand.norm, idx, count, p.rows.dt,
                     p.spec.outputscale, K)
        if p.weighted:
            K = K * p.wmul[idx[:count].long()].unsqueeze(0)
        Xtr = torch.empty(p.n, count, dtype=torch.float64, device=dev)
        nat.dgemm(p.P, K, Xtr)
        out = torch.empty(count, p.n, dtype=torch.float64, device=dev)
        nat.barycentres(Xtr, p.n, count, None, out)
        return out

    def barycentres(self, Xtr, tot):
        n, S = Xtr.shape
        out = torch.empty(S, n, dtype=torch.float64, device=self.device)
        nat.barycentres(Xtr, n, S, tot, out)
        return out

    def car_supported(self, N, m):
        """The step runs on the device: a kernel covers the size AND the rung this device is on still has one."""
        if not nat.car_supported(N, m) or self.car_mode == nat.CAR_HOST:
            return False
        return self.car_mode == nat.CAR_DEFAULT or nat.car_safe_supported(N, m)

    def car_device(self, X, mu_in, phi_out=None):
        """Tchernychova_Lyons_CAR on the device -> (keep_rank int32 (N,), w_star (N,), n_keep int32
        (1,), mu_out (N,)); nothing leaves the GPU.  n_keep = -1: the launches gave up (SOBER_CAR_DEFAULT only) and
        keep_rank / w_star are unwritten -- `car_device_checked` is the form that recovers."""
        N = X.shape[0]
        dev = self.device
        keep_rank = torch.empty(N, dtype=torch.int32, device=dev)
        w_star = torch.empty(N, dtype=torch.float64, device=dev)
        n_keep = torch.empty(1, dtype=torch.int32, device=dev)
        mu_out = torch.empty(N, dtype=torch.float64, device=dev)
        nat.car_device(X, mu_in, keep_rank, w_star, n_keep, mu_out, phi_out=phi_out, mode=min(self.car_mode, nat.CAR_SAFE))
        return keep_rank, w_star, n_keep, mu_out

    def car_device_checked(self, X, mu_in, phi_out=None, also=()):
        """`car_device` + the verdict on the host (one synchronisation, which every caller needs anyway) + recovery:
        a step that gave up is redone on the single-workgroup kernels when they cover the size.
        -> (keep_rank_d, w_star_d, keep_rank host, n_keep, host copies of `also`) or None: host route."""
        N, m = X.shape[0], X.shape[1] + 1
        for _ in range(2):
            if not self.car_supported(N, m):
                return None
            keep_rank, w_star, n_keep_d, _mu = self.car_device(X, mu_in, phi_out)
            host = self.to_host(keep_rank, n_keep_d, *also)
            n_keep = int(host[1][0])
            if n_keep >= 0:
                return keep_rank, w_star, host[0], n_keep, host[2:]
            self._car_downgrade(nat.CAR_SAFE if nat.car_safe_supported(N, m) and self.car_mode == nat.CAR_DEFAULT
                                else nat.CAR_HOST, "Caratheodory step")
        return None

    def car_obj_device(self, X, mu_in, obj_head=None):
        """The Caratheodory step of the acquisition-guided branch on the device: X (N, n + 1) carries the objective
        in its last column (SOBER/_rchq.py:79-81, :149-150), the step runs with n + 2 functions (:84, :173), then the
        extra elimination along the null vector of [X_p; 1] (:87-106, :177-196) -- that vector is the one-column
        null-space basis the same kernels produce for the n + 2 survivors.  `obj_head`: the final level's objective
        values by LIST POSITION (the reference's indexing quirk, :89); None: the last column (:179).
        -> (keep_rank_d, w_star_d, keep_rank host, n_keep, first-step trace) or None when the first step does not
        leave exactly n + 2 points (the reference then takes a singular vector of a full-rank matrix: host route)."""
        N, n1f = X.shape                                     # n1f = n + 1 functions incl. the objective
        dev = self.device
        n1 = n1f + 1                                         # what the first step leaves when the branch is regular
        for _ in range(2):
            if not self.car_supported(N, n1f + 1) or not self.car_supported(n1, n1f):
                return None
            # BOTH steps are enqueued before anything is read back (one synchronisation per level instead of three): the
            # second one assumes the regular outcome of the first -- exactly n + 2 survivors, whose rows go to their
            # rank's place (sober_rank_scatter; ranks outside 0..n1-1 are dropped) -- and is simply
            # discarded when the host then finds another count
            kr1, w1, nk1_d, _mu = self.car_device(X, mu_in)
            Xp = torch.zeros(n1, n1f - 1, dtype=torch.float64, device=dev)
            objp = torch.zeros(n1, dtype=torch.float64, device=dev)
            ocol = (X[:, n1f - 1] if obj_head is None else obj_head[:N]).contiguous()
            nat.rank_scatter(X, n1f - 1, ocol, kr1, n1, Xp, objp)
            phi = torch.empty(n1, 1, dtype=torch.float64, device=dev)
            scratch = [torch.empty(n1, dtype=t_, device=dev) for t_ in (torch.int32, torch.float64, torch.float64)]
            nkx = torch.empty(1, dtype=torch.int32, device=dev)
            # (phi_out is filled by the stand-alone bidiagonalisation + Phi launches: nothing here can give up)
            nat.car_device(Xp, w1[:n1].contiguous(), scratch[0], scratch[1], nkx, scratch[2], phi_out=phi)
            keep_rank = torch.empty(N, dtype=torch.int32, device=dev)
            w_star = torch.empty(N, dtype=torch.float64, device=dev)
            n_keep = torch.empty(1, dtype=torch.int32, device=dev)
            nat.second_elimination(phi, objp, w1, kr1, n1, keep_rank, w_star, n_keep)
            (kr1_h, nk1_h, keep_h, nk_h) = self.to_host(kr1, nk1_d, keep_rank, n_keep)
            nk1 = int(nk1_h[0])
            if nk1 < 0:                                      # the first step gave up: once more on the next rung
                self._car_downgrade(nat.CAR_SAFE if nat.car_safe_supported(N, n1f + 1) and self.car_mode == nat.CAR_DEFAULT
                                    else nat.CAR_HOST, "Caratheodory step")
                continue
            if nk1 != n1:
                return None
            return keep_rank, w_star, keep_h, int(nk_h[0]), (kr1_h, w1, n1)
        return None

    def level_update(self, idx_cur, pos0, count, S, E, keep_rank, w_star, tot, n_keep, mu, idx_new, new_pos0):
        nat.level_update(idx_cur, 0, pos0, count, S, E, keep_rank, w_star, tot, n_keep, mu, idx_new, new_pos0)

    def scatter_weights(self, idx_cur, sel, w, mu):
        out = torch.empty(sel.numel(), dtype=torch.int64, device=self.device)
        if sel.numel():
            nat.scatter_weights(idx_cur, sel, w, sel.numel(), mu, out)
        return out

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
            ws = torch.empty(nat.nonzero_ws_bytes(N), dtype=torch.uint8, device=self.device)
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
