"""The Nystrom side of `HipOps`: make_cov_psd's jitter ladder + torch.svd_lowrank's range finder as ONE native job
(csrc/nystrom_exec.cpp), its verdicts read back, the hand-over to the host route.  Mixed into `sober_amd._ops_hip.HipOps`."""
from __future__ import annotations

import os
import warnings

import torch

from . import _native as nat
from . import _rng
from ._ops_plan import Plan


class _NystromOps:

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
        if M > nat.nystrom_max_n() or s > 2 * nat.chol_max_n() or s >= M:
            if s < M:                                             # (s >= M is the reference's own degenerate case, not a size limit)
                self.size_cliff("nystrom", f"N_nys = {M}, batch = {s + 1}: beyond the device Nystrom route (N_nys <= "
                                           f"{nat.nystrom_max_n()}, batch <= {2 * nat.chol_max_n() + 1}); make_cov_psd and svd_lowrank run on host LAPACK "
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
        n8 = 8 * (n_r + 1 + n_orth2)
        f64s, i32s = hb[:n8].view(torch.float64), hb[n8:].view(torch.int32)
        piv_h, pivs_rf = f64s[:n_r + 1], f64s[n_r + 1:n_r + 1 + n_orth2]
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
            print("nystrom: pivot ratios of the range finder's blocks", [float(pivs_rf[k + 1]) for k in range(0, n_orth2, 2)])
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
