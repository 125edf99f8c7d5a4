"""The level side of `HipOps`: the level executor's job struct, the queued / synchronised / sharded level loops and the
final level behind one native call each, the step-by-step forms the engine falls back to, the Caratheodory step and its
downgrade ladder (fused / multi-CU launches -> single-workgroup kernels -> host).  Mixed into `sober_amd._ops_hip.HipOps`."""
from __future__ import annotations

import warnings

import torch

from . import _native as nat
from ._ops_plan import Plan


class _LevelOps:

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
            w["dR"] = torch.empty(nat.LEVEL_QUEUE + 1, dtype=torch.int64, device=dev)   # (every entry a chain reads, the chain wrote)
            w["h_dR"] = torch.zeros(nat.LEVEL_QUEUE + 1, dtype=torch.int64, pin_memory=True)
            job.dR, job.h_dR = w["dR"].data_ptr(), w["h_dR"].data_ptr()
            if S % 2 == 0:
                # levels derived from level 0's class sums (csrc/level_class.hip): ping-pong class sums, 2^D S columns at most
                cl = 1 << nat.CLASS_MAX_DEPTH
                w["Gc"] = [torch.empty(nr * S * (cl >> k), dtype=f64, device=dev) for k in (0, 1)]
                w["totc"] = [torch.empty(S * (cl >> k), dtype=f64, device=dev) for k in (0, 1)]
                w["cls_scale"], w["cls_sof"] = torch.empty(S, dtype=f64, device=dev), torch.empty(S, dtype=torch.int32, device=dev)
                for k in (0, 1):
                    job.Gc[k], job.totc[k] = w["Gc"][k].data_ptr(), w["totc"][k].data_ptr()
                job.cls_scale, job.cls_sof = w["cls_scale"].data_ptr(), w["cls_sof"].data_ptr()
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
                          + ("the kernels without such waits (one workgroup, or a launch per dependency beyond batch 100)" if mode == nat.CAR_SAFE
                             else "the host LAPACK route")
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
        # the first level's sums of an unsharded pool without leftovers: by 2^D element classes, from which the queued
        # loop forms levels 1 .. D without evaluating the kernel again (csrc/level_class.hip; D = 0: as ever)
        job.class_depth = 0
        if phase == 1 and pos0 == 0 and count == E * S and job.Gc[0] and job.dR and self.level_classes:
            job.class_depth = int(nat.load().sober_level_class_depth(job.variant, job.n_rows, count, S))
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

    def level_loop_obj(self, p: Plan, idx_cur, idx_new, R: int, S: int, mu, obj, sums_ready: bool):
        """The acquisition-guided branch's levels (SOBER/_rchq.py:116-221 with :138-150, :173-196) as ONE queued chain of the level
        executor (csrc/level_exec.cpp: sober_level_loop_obj): set sums, the objective's column, the step with n + 2 functions, the
        second elimination and the update of every level that certainly exists, one synchronisation -- instead of a visit to
        Python (and a read-back) per level.  Returns (idx_cur, idx_new, R, levels done) for the engine's synchronised loop, which
        takes whatever the chain did not complete (an irregular level, the last levels, the final one); None: not applicable."""
        job = self._job(p, S)
        n = job.n
        if not self.queue_obj_levels or not self.queue_levels or not job.dR or job.variant not in (nat.LEVEL_MFMA, nat.LEVEL_TANI):
            return None
        if self.car_mode == nat.CAR_HOST or not self.obj_null_kernel or not self.car_supported(S, n + 2) \
                or not bool(nat.load().sober_null_vector_supported(n)):
            return None
        nat._req(mu, torch.float64, "mu"); nat._req(idx_cur, torch.int32, "idx"); nat._req(idx_new, torch.int32, "idx")
        nat._req(obj, torch.float64, "obj")
        dev, f64, w = self.device, torch.float64, p.ws
        if "obj_job" not in w:
            oj = nat.ObjJob()
            w["obj_X"] = torch.empty(S, n + 1, dtype=f64, device=dev)
            w["obj_f"] = torch.empty(3, S, dtype=f64, device=dev)              # ocol | w1 | null_row
            w["obj_i"] = torch.empty(S + 2, dtype=torch.int32, device=dev)     # kr1 | nk1 | status
            nbytes = nat.car_ws_bytes(S, n + 2)
            w["obj_car_ws"] = (torch.empty(max(nbytes // 8, 1), dtype=f64, device=dev), nbytes)
            oj.X_tmp, oj.ocol, oj.w1, oj.null_row = (w["obj_X"].data_ptr(), w["obj_f"][0].data_ptr(), w["obj_f"][1].data_ptr(),
                                                       w["obj_f"][2].data_ptr())
            oj.kr1, oj.nk1, oj.status = w["obj_i"].data_ptr(), w["obj_i"][S:].data_ptr(), w["obj_i"][S + 1:].data_ptr()
            w["obj_job"] = oj
        oj = w["obj_job"]
        oj.obj = obj.data_ptr()
        job.mu = mu.data_ptr()
        saved = (job.car_ws, job.car_ws_bytes, job.class_depth)
        job.car_ws, job.car_ws_bytes, job.class_depth = w["obj_car_ws"][0].data_ptr(), w["obj_car_ws"][1], 0
        try:
            level_R, R_final, in_b = nat.level_loop_obj(job, oj, R, idx_cur, idx_new, sums_ready, nat._stream(mu))
        finally:
            job.car_ws, job.car_ws_bytes, job.class_depth = saved
        if in_b:
            idx_cur, idx_new = idx_new, idx_cur
        return idx_cur, idx_new, R_final, len(level_R)

    def level_loop(self, p: Plan, idx_cur, idx_new, R: int, S: int, mu, sums_ready: bool, row_offset: int = 0):
        """The whole halving loop of an unsharded pool while R > S (SOBER/_rchq.py:116-221) in ONE call of the
        level executor (csrc/level_exec.cpp: sober_level_loop) -- no trip through Python between a level's
        verdict and the next level's launches.  Returns (idx_cur, idx_new, R) for the terminal branch."""
        import math
        job = self._job(p, S)
        nat._req(mu, torch.float64, "mu"); nat._req(idx_cur, torch.int32, "idx"); nat._req(idx_new, torch.int32, "idx")
        job.mu = mu.data_ptr()
        if not sums_ready:
            job.class_depth = 0                                 # (class sums come from the phase-1 call only)
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
        self.last_levels = {"R": [int(v) for v in level_R], "derived": int(job.class_depth) if sums_ready else 0}
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
        job.class_depth = 0
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
        if p.cand is not None:
            f.cand_sc, f.cand_norm = p.cand.data.data_ptr(), nat._ptr(p.cand.norm)
        else:                                                # (the raw pool: the final level scales the rows it touches)
            Xr, ls = p.cand_raw, p.spec.lengthscale
            f.cand_sc, f.cand_norm = None, None
            f.cand_raw, f.ld_raw, f.d_raw = Xr.data_ptr(), Xr.stride(0), Xr.shape[1]
            f.ls, f.ls_len = ls.data_ptr(), ls.numel()
            f.sc_buf = self._buf(p, "sc_final", S * p.rows.dt).data_ptr()
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
        dev = self.device
        job.mu = mu.data_ptr()
        fin = self._final_job(p, S, mu, row_offset)
        out_idx, out_w = p._fin_out
        st = torch.cuda.current_stream(dev)
        nat.level_final_job(job, fin, idx_cur, R, st.cuda_stream)
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
        if p.cand is not None:
            nat.pairwise(p.kind, p.rows.data, p.rows.norm, p.cand.data, p.cand.norm, idx, count, p.rows.dt,
                         p.spec.outputscale, K)
        else:                                                # (the raw pool: these rows scaled here)
            sc = torch.empty(count, p.rows.dt, dtype=torch.float64, device=dev)
            nat.scale_points_idx(p.cand_raw, idx, count, p.spec.lengthscale, sc)
            nat.pairwise(p.kind, p.rows.data, p.rows.norm, sc, None, None, count, p.rows.dt, p.spec.outputscale, K)
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
        # Round 6: the second elimination's direction by ONE launch on the survivors' own matrix (csrc/null_vector.hip: Gauss-
        # Jordan with partial pivoting) instead of a second run of the Caratheodory kernels; the first step's verdict is read
        # on the device, so nothing is read back between the launches.  Sizes beyond that kernel keep the second step below.
        if self.obj_null_kernel and bool(nat.load().sober_null_vector_supported(n1f - 1)) and self.car_supported(N, n1f + 1):
            for _ in range(2):
                # every verdict of the level in ONE int32 block -- [keep_rank N | n_keep | first step's n_keep | status | first
                # step's ranks N] -- so that one copy brings them to the host (they were five copies and their five launches)
                flags = torch.empty(2 * N + 3, dtype=torch.int32, device=dev)
                keep_rank, n_keep, nk1_d, status, kr1 = flags[:N], flags[N:N + 1], flags[N + 1:N + 2], flags[N + 2:N + 3], flags[N + 3:]
                w1 = torch.empty(N, dtype=torch.float64, device=dev)
                mu1 = torch.empty(N, dtype=torch.float64, device=dev)
                nat.car_device(X, mu_in, kr1, w1, nk1_d, mu1, mode=min(self.car_mode, nat.CAR_SAFE))
                null_row = torch.empty(N, dtype=torch.float64, device=dev)
                nat.null_vector(X, n1f - 1, kr1, nk1_d, n1, null_row, status)
                ocol = (X[:, n1f - 1] if obj_head is None else obj_head[:N]).contiguous()
                w_star = torch.empty(N, dtype=torch.float64, device=dev)
                nat.second_elimination_rows(null_row, ocol, w1, kr1, nk1_d, n1, keep_rank, w_star, n_keep, status=status)
                (fl_h,) = self.to_host(flags)
                keep_h, nk_h, nk1, st_h, kr1_h = fl_h[:N], int(fl_h[N]), int(fl_h[N + 1]), int(fl_h[N + 2]), fl_h[N + 3:]
                if nk1 == n1 and st_h == 0:
                    return keep_rank, w_star, keep_h, nk_h, (kr1_h, w1, n1)
                if nk1 >= 0:
                    return None                              # irregular first step or a rank-deficient A2: the host route
                self._car_downgrade(nat.CAR_SAFE if nat.car_safe_supported(N, n1f + 1) and self.car_mode == nat.CAR_DEFAULT
                                    else nat.CAR_HOST, "Caratheodory step")
                if not self.car_supported(N, n1f + 1):
                    return None
            return None
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

    def obj_set_sums(self, obj, mu, idx_cur, pos0, count, S, E):
        """The objective's row of a level's set sums (SOBER/_rchq.py:138-146, :157-163) in one launch."""
        out = torch.empty(S, dtype=torch.float64, device=self.device)
        nat.obj_set_sums(obj, mu, idx_cur, pos0, count, S, E, out)
        return out

    def level_update(self, idx_cur, pos0, count, S, E, keep_rank, w_star, tot, n_keep, mu, idx_new, new_pos0):
        nat.level_update(idx_cur, 0, pos0, count, S, E, keep_rank, w_star, tot, n_keep, mu, idx_new, new_pos0)

    def scatter_weights(self, idx_cur, sel, w, mu):
        out = torch.empty(sel.numel(), dtype=torch.int64, device=self.device)
        if sel.numel():
            nat.scatter_weights(idx_cur, sel, w, sel.numel(), mu, out)
        return out
