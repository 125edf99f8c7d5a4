"""Device operations of the recombination engine on the HIP library (the only product backend).

`HipOps` owns no algorithmic decision: it allocates workspaces with torch and enqueues the C-ABI
kernels on the current stream.  The level logic lives in `_engine.py`."""
from __future__ import annotations

import torch

from . import _native as nat
from ._kernel import KernelSpec, PointSet, posterior_mean, prepare_points, woodbury


class Plan:
    """Per-step device state shared by all levels."""
    pass


class HipOps:
    name = "hip"

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise nat.SoberHipError("sober_amd needs a HIP device (torch device 'cuda'); got "
                                    f"{self.device}.  There is no CPU fallback.")
        nat.load()
        self._pin = {}
        self.prof = None        # list -> (start_event, end_event, kernel entries) per level_reduce launch
        self.use_mfma = True    # False: VALU-only level kernel (direct differences), kept for A/B runs

    # ------------------------------------------------------------------ plan
    def build_plan(self, spec: KernelSpec, mode: str, X_nys, X_cand) -> Plan:
        p = Plan()
        p.spec, p.mode = spec, mode
        p.kind = nat.KIND_BY_NAME[spec.kind]
        p.M = X_nys.shape[0]
        dev = self.device
        corrected = mode != "kernel"
        p.n_obs = spec.X_obs.shape[0] if corrected else 0
        stacked = torch.cat([X_nys.to(torch.float64), spec.X_obs], 0) if corrected else X_nys
        p.rows = prepare_points(spec, stacked)                    # [X_nys; X_obs]
        p.cand = prepare_points(spec, X_cand)
        p.Mtot = len(p.rows)
        p.weighted = mode == "weighted_predictive_covariance"
        p.mean_nys = p.wmul = None
        if p.weighted:
            p.mean_nys = posterior_mean(spec, p.rows.rows(0, p.M))
            p.wmul = posterior_mean(spec, p.cand)                 # mu_y of SOBER/_kernel.py:41
        p.T = None
        if corrected:
            # T = KxX @ W with KxX = k(X_nys, X_obs) (SOBER/_gp.py:293,295)
            Kall = torch.empty(p.Mtot, p.M, dtype=torch.float64, device=dev)
            nys = p.rows.rows(0, p.M)
            nat.pairwise(p.kind, p.rows.data, p.rows.norm, nys.data, nys.norm, None, p.M, p.rows.dt,
                         spec.outputscale, Kall)
            p.Kall = Kall
            W = woodbury(spec)
            p.T = torch.empty(p.M, p.n_obs, dtype=torch.float64, device=dev)
            nat.dgemm(Kall[p.M:], W, p.T, transa=True)            # K_Xn^T W == KxX W (k symmetric)
        # matrix-core level kernel: augmented copies of the row table and the pool
        p.da = nat.aug_dim(X_nys.shape[1]) if (p.kind != nat.KIND_TANIMOTO and self.use_mfma) else -1
        if p.da > 0:
            Xn64 = X_nys.to(torch.float64)
            center = Xn64.mean(0).contiguous()                    # any shift works; this one keeps |x~| small
            p.rows_aug = torch.empty(p.Mtot, p.da, dtype=torch.float64, device=dev)
            p.cand_aug = torch.empty(X_cand.shape[0], p.da, dtype=torch.float64, device=dev)
            nat.augment_points(stacked.to(torch.float64).contiguous(), spec.lengthscale, center, 0, p.rows_aug)
            Xc = X_cand if (X_cand.dtype == torch.float64 and X_cand.stride(-1) == 1) else \
                X_cand.to(torch.float64).contiguous()
            nat.augment_points(Xc, spec.lengthscale, center, 1, p.cand_aug)
        p.P = None
        p.ws = {}
        return p

    def gram(self, p: Plan):
        """kernel(pt, pt) of SOBER/_rchq.py:35 for the plan's mode."""
        dev = self.device
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
        U = U.to(self.device, torch.float64).contiguous()
        p.n = U.shape[0]
        P1 = U * p.mean_nys.unsqueeze(0) if p.weighted else U
        if p.T is None:
            p.P = P1.contiguous()
            return
        P = torch.empty(p.n, p.Mtot, dtype=torch.float64, device=self.device)
        P[:, :p.M] = P1
        nat.dgemm(P1.contiguous(), p.T, P[:, p.M:], alpha=-1.0)
        p.P = P

    # ------------------------------------------------------------------ Nystrom basis on the device
    def nystrom_basis_device(self, p: Plan, s: int, max_iter: int = 10):
        """ker_svd_sparsify (SOBER/_rchq.py:34-39) with the N_nys x N_nys work on the GPU:
          make_cov_psd: |cov| and the symmetry test in one kernel; the jitter rung is found by bisection
                        with the one-workgroup Cholesky (PD-ness is monotone in the shift), then the
                        reference's own sequence of diagonal additions is applied;
          svd_lowrank : randn from the CPU generator (same draw as the reference); range finder with
                        MFMA GEMMs + CholeskyQR2; only the q x M matrix B goes to the host for LAPACK's
                        SVD (2 ms), U = Q U_B back on the device.
        Returns (U (s, M) on the device, the Gram matrix) or None when the literal host path must decide
        (exactly symmetric Gram, sizes beyond the kernels, ill-conditioned range finder)."""
        import warnings
        dev, M = self.device, p.M
        G = self.gram(p)
        if M > nat.chol_max_n() or s > 256 or s >= M:
            return None
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        C = torch.empty_like(G)
        nat.abs_sym(G, C, flag)
        if int(flag.item()) == 0:
            return None                                    # symmetric input: is_psd(cov) itself has to run
        warnings.warn("Estimated covariance matrix was not positive semi-definite. Conveting...")
        # every rung of the ladder (after k = 0 .. max_iter jitter additions) is probed at once: one
        # workgroup per rung, one launch; the CPU draws svd_lowrank's randn meanwhile (it is the next
        # consumer of the generator in the reference too: make_cov_psd draws nothing)
        n_r = max_iter + 1
        shifts = torch.tensor([1e-5 * (2 ** k - 1) for k in range(n_r)], dtype=torch.float64, device=dev)
        info = torch.zeros(n_r, dtype=torch.int32, device=dev)
        work = self._buf(p, "chol_work", n_r * M * M)
        nat.cholesky_probe(C, shifts, work, info)
        R = torch.randn(M, s, dtype=torch.float64)
        (info_h,) = self.to_host(info)
        ok = (info_h == 0).tolist()
        k_first = ok.index(True) if any(ok) else max_iter + 1
        # SOBER/_utils.py:150-156: k_first additions (max_iter + 1 of them before the diagonal fallback), one launch
        nat.jitter_ladder(C, k_first)
        if k_first > max_iter:
            C = torch.diag(C.diagonal().clone())           # :155
        U = self._svd_lowrank_device(C, s, R)
        if U is None:
            return None
        return U, G

    def _orth(self, Y, infos, pivs, slot, passes: int = 2, want_q: bool = True):
        """CholeskyQR2: orthonormal basis of range(Y) with the flag of Householder QR.
        passes=1 (the intermediate blocks of the power iteration): the same subspace to the same accuracy
        (the first triangular solve decides it), orthonormal only to cond(Y)^2 eps -- which is all the next
        product A Q needs; pivs[slot + 1] then holds min pivot / max diagonal of the Gram matrix (~ cond^-2).
        want_q=False: only the triangular factors are needed; returns (L1, L2) with Y = Q (L2 L1)^T."""
        q = Y.shape[1]
        Ls = []
        for it in range(passes):
            Gm = torch.empty(q, q, dtype=torch.float64, device=self.device)
            nat.dgemm(Y, Y, Gm, transa=True)
            dmax = Gm.diagonal().amax() if passes == 1 else None
            if q <= 128:                                    # LDS-resident factorisation
                Lc = torch.zeros(q, q, dtype=torch.float64, device=self.device)
                nat.chol_small(Gm, Lc, infos[slot + it:slot + it + 1], pivs[slot + it:slot + it + 1])
            else:                                           # blocked one-workgroup Cholesky, in place
                Lc = Gm
                nat.cholesky(Lc, 0.0, infos[slot + it:slot + it + 1], pivs[slot + it:slot + it + 1])
                Lc = Lc.tril_()
            if passes == 1:
                pivs[slot + 1] = pivs[slot] / dmax
            Ls.append(Lc)
            if not want_q and it == passes - 1:
                return Ls
            Q = torch.empty_like(Y)
            nat.trsm_rows(Y, Lc, Q)                             # Q = Y R^-1, one wave per row
            Y = Q
        return Y

    # single-pass CholeskyQR is accepted for an intermediate block while min pivot / max diagonal of its Gram
    # matrix (~ cond(Y)^-2) stays above this: orthonormality then holds to ~1e-6 and the subspace to eps cond(Y)
    ORTH1_MIN_RATIO = 1e-10

    def _svd_lowrank_device(self, A, q, R_host, niter: int = 2):
        """torch.svd_lowrank(A, q) (Halko et al. Alg. 4.4 / 5.1, as in torch/_lowrank.py) for a square
        device matrix; returns -U^T (q, M) like SOBER/_rchq.py:38, or None if CholeskyQR lost rank."""
        from ._engine import host_lapack_threads
        dev, M = self.device, A.shape[0]
        R = R_host.to(dev)                                           # CPU generator: the reference's draw
        n_orth = 1 + 2 * niter
        infos = torch.zeros(2 * n_orth, dtype=torch.int32, device=dev)
        pivs = torch.zeros(2 * n_orth, dtype=torch.float64, device=dev)
        Y = torch.empty(M, q, dtype=torch.float64, device=dev)
        nat.dgemm(A, R, Y)
        # torch/_lowrank.py orthonormalises after every product; only range(Q) of the LAST block enters the
        # result, the earlier QRs are there for conditioning -- one CholeskyQR pass each does that
        last = 2 * niter
        Q = self._orth(Y, infos, pivs, 0, passes=1 if last > 0 else 2)
        slot, k = 2, 0
        for _ in range(niter):
            k += 1
            nat.dgemm(A, Q, Y, transa=True)                          # A^H Q
            Q = self._orth(Y, infos, pivs, slot, passes=1); slot += 2
            Y = torch.empty(M, q, dtype=torch.float64, device=dev)
            k += 1
            nat.dgemm(A, Q, Y)
            Q = self._orth(Y, infos, pivs, slot, passes=2 if k == last else 1); slot += 2
            Y = torch.empty(M, q, dtype=torch.float64, device=dev)
        # B = Q^H A (q x M).  Its left singular vectors are those of the q x q factor of B^T = Qb Rb:
        # B = Rb^T Qb^T  =>  U_B = left singular vectors of Rb^T.  CholeskyQR2 of B^T on the device
        # (B^T = Q1 L1^T, Q1 = Q2 L2^T  =>  Rb^T = L1 L2; Q2 itself is never formed), so only q x q goes to
        # the host for LAPACK's SVD (0.6 ms instead of 2.2 ms for q x M).
        Bt = torch.empty(M, q, dtype=torch.float64, device=dev)
        nat.dgemm(A, Q, Bt, transa=True)                             # (Q^H A)^T = A^T Q
        infos_b = torch.zeros(2, dtype=torch.int32, device=dev)
        pivs_b = torch.zeros(2, dtype=torch.float64, device=dev)
        L1, L2 = self._orth(Bt, infos_b, pivs_b, 0, want_q=False)
        RbT = torch.empty(q, q, dtype=torch.float64, device=dev)
        nat.dgemm(L1, L2, RbT)                                       # Rb^T = L1 L2 (lower triangular)
        RbT_h, infos_h, pivs_h, infos_bh, pivs_bh = self.to_host(RbT, infos, pivs, infos_b, pivs_b)
        # a second CholeskyQR pass works on a nearly orthonormal block: its pivots must be ~1
        single = [s for s in range(0, 2 * n_orth, 2) if s // 2 != last]
        if bool((infos_h != 0).any()) or bool((infos_bh != 0).any()) or float(pivs_bh[1]) < 0.5 \
                or float(pivs_h[2 * last + 1]) < 0.5 \
                or any(not (float(pivs_h[s + 1]) >= self.ORTH1_MIN_RATIO) for s in single):
            return None
        with host_lapack_threads(M):
            Ub, _, _ = torch.linalg.svd(RbT_h, full_matrices=False)
        U = torch.empty(M, q, dtype=torch.float64, device=dev)
        nat.dgemm(Q, self.from_host(Ub.contiguous()), U)
        return (-1 * U.T).contiguous()

    # ------------------------------------------------------------------ levels
    def _prof_begin(self):
        if self.prof is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(self.device))      # the stream the kernel is launched on
        return ev

    def _prof_end(self, ev, entries):
        if ev is None:
            return
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record(torch.cuda.current_stream(self.device))
        self.prof.append((ev, ev1, int(entries)))

    def _buf(self, p, name, numel):
        t = p.ws.get(name)
        if t is None or t.numel() < numel:
            t = torch.empty(numel, dtype=torch.float64, device=self.device)
            p.ws[name] = t
        return t

    def _level_reduce(self, p, idx, idx_off, pos0, count, S, mu, n_chunks, partG, ldg, partTot, tot_limit):
        if p.da > 0:
            nat.level_reduce_mfma(p.kind, p.rows_aug, p.cand_aug, p.da, idx, idx_off, pos0, count, S, mu,
                                  p.wmul, p.spec.outputscale, n_chunks, partG, ldg, 0, partTot, tot_limit)
        else:
            nat.level_reduce(p.kind, p.rows.data, p.rows.norm, p.cand.data, p.cand.norm, p.rows.dt, idx,
                             idx_off, pos0, count, S, mu, p.wmul, p.spec.outputscale, n_chunks, partG, ldg, 0,
                             partTot, tot_limit)

    def level_moments(self, p: Plan, idx, pos0, count, S, E, mu):
        """Partial (n, S) projected set sums and (S,) set masses over the local list positions
        [pos0, pos0+count) of a level with E full elements (SOBER/_rchq.py:116-164 minus the
        division).  Q1: leftovers (p >= E*S) are summed into set p mod S AND into set S-1; only
        the latter reaches `tot`."""
        dev = self.device
        ES = E * S
        n_chunks = nat.level_chunks(p.Mtot, pos0, count, S)
        partG = self._buf(p, "partG", n_chunks * p.Mtot * S)
        partTot = self._buf(p, "partTot", n_chunks * S)
        ev = self._prof_begin()
        self._level_reduce(p, idx, 0, pos0, count, S, mu, n_chunks, partG, S, partTot, ES)
        self._prof_end(ev, count * p.Mtot)
        extraG = extraTot = None
        n_xchunks, XS = 0, 16
        lo = max(pos0, ES)                                   # first local leftover position
        n_left = pos0 + count - lo
        if n_left > 0:
            # second placement of the leftovers (:153-164): the same kernel over the leftover
            # positions alone, spread over XS pseudo-sets that sum_partials folds into set S-1
            n_xchunks = nat.level_chunks(p.Mtot, 0, n_left, XS)
            extraG = self._buf(p, "extraG", n_xchunks * p.Mtot * XS)
            extraTot = self._buf(p, "extraTot", n_xchunks * XS)
            ev = self._prof_begin()
            self._level_reduce(p, idx, lo - pos0, 0, n_left, XS, mu, n_xchunks, extraG, XS, extraTot, n_left)
            self._prof_end(ev, n_left * p.Mtot)
        G = self._buf(p, "G", p.Mtot * S).view(p.Mtot, S)
        tot = torch.empty(S, dtype=torch.float64, device=dev)
        nat.sum_partials(partG, partTot, n_chunks, p.Mtot, S, S, extraG, extraTot, n_xchunks, XS, G, tot)
        Xtr = torch.empty(p.n, S, dtype=torch.float64, device=dev)
        nat.dgemm(p.P, G, Xtr)
        return Xtr, tot

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
        return nat.car_supported(N, m)

    def car_device(self, X, mu_in):
        """Tchernychova_Lyons_CAR on the device -> (keep_rank int32 (N,), w_star (N,), n_keep int32
        (1,), mu_out (N,)); nothing leaves the GPU."""
        N = X.shape[0]
        dev = self.device
        keep_rank = torch.empty(N, dtype=torch.int32, device=dev)
        w_star = torch.empty(N, dtype=torch.float64, device=dev)
        n_keep = torch.empty(1, dtype=torch.int32, device=dev)
        mu_out = torch.empty(N, dtype=torch.float64, device=dev)
        nat.car_device(X, mu_in, keep_rank, w_star, n_keep, mu_out)
        return keep_rank, w_star, n_keep, mu_out

    def level_update(self, idx_cur, pos0, count, S, E, keep_rank, w_star, tot, n_keep, mu, idx_new, new_pos0):
        nat.level_update(idx_cur, 0, pos0, count, S, E, keep_rank, w_star, tot, n_keep, mu, idx_new, new_pos0)

    def scatter_weights(self, idx_cur, sel, w, mu):
        out = torch.empty(sel.numel(), dtype=torch.int64, device=self.device)
        if sel.numel():
            nat.scatter_weights(idx_cur, sel, w, sel.numel(), mu, out)
        return out

    # ------------------------------------------------------------------ plumbing
    def nonzero_i32(self, mu):
        nz = torch.nonzero(mu != 0).flatten()
        out = torch.empty(max(nz.numel(), 1), dtype=torch.int32, device=self.device)
        if nz.numel():
            nat.i64_to_i32(nz, out)
        return out, int(nz.numel())

    def empty_i32(self, n):
        return torch.empty(max(n, 1), dtype=torch.int32, device=self.device)

    def to_host(self, *tensors):
        outs = []
        for i, t in enumerate(tensors):
            key = (i, t.dtype, tuple(t.shape))
            buf = self._pin.get(key)
            if buf is None:
                buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                self._pin[key] = buf
            buf.copy_(t, non_blocking=True)
            outs.append(buf)
        torch.cuda.current_stream(self.device).synchronize()
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

    def build_plan(self, kernel_fn, mode, X_nys, X_cand) -> Plan:
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

    def _level_reduce(self, p, idx, idx_off, pos0, count, S, mu, n_chunks, partG, ldg, partTot, tot_limit):
        nat.level_gather(p.Kmat, idx, idx_off, pos0, count, S, mu, None, n_chunks, partG, ldg, 0, partTot, tot_limit)

    def direct_columns(self, p, idx, count):
        Kc = p.Kmat[idx[:count].long()].contiguous()                      # (count, M)
        Xtr = torch.empty(p.n, count, dtype=torch.float64, device=self.device)
        nat.dgemm(p.P, Kc, Xtr, transb=True)
        out = torch.empty(count, p.n, dtype=torch.float64, device=self.device)
        nat.barycentres(Xtr, p.n, count, None, out)
        return out


CallableKernelOps = MatrixKernelOps
