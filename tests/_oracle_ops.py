"""A CPU test double for `sober_amd._ops_hip.HipOps`, built on the oracle's kernels.

TEST INFRASTRUCTURE ONLY: it lets the `-m "not gpu"` suite drive the product's HOST logic
(`sober_amd._engine.RecombinationEngine`: grouping, leftovers, compaction, sharding, all-reduce
placement) without a GPU.  The product never imports this (nor the oracle)."""
import numpy as np
import torch

from oracle import sober_oracle as O


def to_oracle_spec(spec):
    return O.GPSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise,
                    spec.mean_const, spec.alpha)


class _Plan:
    pass


class OracleOps:
    name = "oracle-test-double"

    def __init__(self):
        self.device = torch.device("cpu")

    def build_plan(self, spec, mode, X_nys, X_cand, pool_owner=None):
        p = _Plan()
        p.kernel = O.Kernel(to_oracle_spec(spec), mode)
        p.X_nys, p.X_cand = X_nys, X_cand
        p.M = X_nys.shape[0]
        return p

    def gram(self, p):
        return p.kernel(p.X_nys, p.X_nys)

    def set_projection(self, p, U):
        p.U, p.n = U, U.shape[0]

    def level_moments(self, p, idx, pos0, count, S, E, mu):
        ES = E * S
        G = torch.zeros(p.M, S, dtype=torch.float64)
        tot = torch.zeros(S, dtype=torch.float64)
        if count > 0:
            c = idx[:count].long()
            pos = pos0 + torch.arange(count)
            K = p.kernel(p.X_nys, p.X_cand[c]) * mu[c].unsqueeze(0)
            G.index_add_(1, pos % S, K)                       # incl. first leftover placement (Q1)
            main = pos < ES
            tot.index_add_(0, (pos % S)[main], mu[c][main])
            if (~main).any():                                 # second placement -> set S-1
                G[:, S - 1] += K[:, ~main].sum(1)
                tot[S - 1] += mu[c][~main].sum()
        return p.U @ G, tot

    def direct_columns(self, p, idx, count):
        c = idx[:count].long()
        return (p.U @ p.kernel(p.X_nys, p.X_cand[c])).T.contiguous()

    def barycentres(self, Xtr, tot):
        return (Xtr / tot.unsqueeze(0)).T.contiguous()

    def level_update(self, idx_cur, pos0, count, S, E, keep_rank, w_star, tot, n_keep, mu, idx_new, new_pos0):
        ES = E * S
        for t in range(count):
            p = pos0 + t
            c = int(idx_cur[t])
            if p < ES:
                s = p % S
                k = int(keep_rank[s])
                dst = (p // S) * n_keep + k if k >= 0 else -1
            else:
                s = S - 1
                k = int(keep_rank[s])
                dst = E * n_keep + (p - ES) if k >= 0 else -1
            if dst >= 0:
                mu[c] = (mu[c] * w_star[k]) / tot[s]
                idx_new[dst - new_pos0] = c
            else:
                mu[c] = 0.0

    def scatter_weights(self, idx_cur, sel, w, mu):
        c = idx_cur[sel.long()].long()
        mu[c] = w
        return c

    def nonzero_i32(self, mu):
        nz = torch.nonzero(mu != 0).flatten().to(torch.int32)
        n = int(nz.numel())
        return (nz if n else torch.zeros(1, dtype=torch.int32)), n

    def empty_i32(self, n):
        return torch.zeros(max(n, 1), dtype=torch.int32)

    def to_host(self, *tensors):
        return [t.clone() for t in tensors]

    def from_host(self, t, dtype=None):
        return t if dtype is None else t.to(dtype)
