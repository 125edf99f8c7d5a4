"""The step's device state (`Plan`) and the part of `HipOps` that builds it: the row table's side in one native call, the
pool's side at first use, the caches kept across calls (bit-packed fingerprint pool, the pool's posterior mean), the
Gram matrix and the projection.  Mixed into `sober_amd._ops_hip.HipOps`."""
from __future__ import annotations


import torch

from . import _native as nat
from ._kernel import KernelSpec, PointSet, posterior_mean, prepare_points, woodbury


class Plan:
    """Per-step device state shared by all levels.  The POOL's side of it (scaled / augmented candidates, the pool's
    posterior mean) is prepared at first use: the Nystrom chain needs the row table only, so the host enqueues that
    chain first and prepares the pool while the GPU is busy with it (build_plan: `_pool_prep`)."""
    _POOL_FIELDS = ("cand", "cand_raw", "cand_aug", "rows_aug", "wmul")

    def __getattr__(self, name):                                  # (only reached when the attribute is not set yet)
        if name in Plan._POOL_FIELDS:
            prep = self.__dict__.get("_pool_prep")
            if prep is not None:
                self.__dict__["_pool_prep"] = None
                prep()
                if name in self.__dict__:
                    return self.__dict__[name]
        raise AttributeError(name)



class _PlanOps:

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
            # The pool in SCALED form (x / lengthscale, padded) is what the VALU level kernel, the pool's posterior mean and
            # the final direct level read.  On the matrix-core path (augmented pool below) without the weighted mode only the
            # final level is left -- at most 2 b rows -- so the copy is not made (65 us and 192 MB of writes at 1M x 20, half
            # a millisecond at 8M): that level scales its own rows from the raw pool (sober_scale_points_idx, same quotients).
            Xc64 = X_cand if (X_cand.dtype == torch.float64 and X_cand.stride(-1) == 1) else X_cand.to(torch.float64).contiguous()
            if p.da > 0 and not p.weighted and p.kind != nat.KIND_TANIMOTO and self.lazy_scaled_pool:
                p.cand, p.cand_raw = None, Xc64
            else:
                p.cand = self._packed_pool(spec, X_cand, pool_owner)
                p.cand_raw = None
            p.wmul = self._pool_mean(spec, p.cand, X_cand, pool_owner) if p.weighted else None   # mu_y of SOBER/_kernel.py:41
            # matrix-core level kernel: augmented copies of the row table and the pool
            if p.da > 0:
                p.rows_aug = torch.empty(p.Mtot, p.da, dtype=torch.float64, device=dev)
                p.cand_aug = torch.empty(X_cand.shape[0], p.da, dtype=torch.float64, device=dev)
                Xo = spec.X_obs if corrected else None
                if Xn64.shape[1] <= 30 and (Xo is None or (Xo.dtype == torch.float64 and Xo.stride(-1) == 1)):
                    # centre (the Nystrom points' mean: any shift works, this one keeps |x~| small) + both tables: two launches
                    center = torch.empty(Xn64.shape[1], dtype=torch.float64, device=dev)
                    nat.augment_plan(Xn64 if Xn64.stride(-1) == 1 else Xn64.contiguous(), Xo, Xc64, spec.lengthscale, center,
                                     p.rows_aug, p.cand_aug)
                else:
                    st = stacked if stacked is not None else (torch.cat([Xn64, spec.X_obs], 0) if corrected else Xn64)
                    center = Xn64.mean(0).contiguous()
                    nat.augment_points(st.to(torch.float64).contiguous(), spec.lengthscale, center, 0, p.rows_aug)
                    nat.augment_points(Xc64, spec.lengthscale, center, 1, p.cand_aug)
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
