#!/usr/bin/env python3
"""Benchmark of the kernel-recombination step (BASELINE.json metric) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W          (N > 1)

A "step" is ONE `sampling_recombination` call -- Gram + Nystrom basis + every halving level +
the final direct level -- on a synthetic pool that is already resident in HBM:
  N=1   BASELINE.json config 2: Ackley-shaped d=10, RBF, N_rec=100k, N_nys=500, batch=100,
        n_obs=200 (inputs = tests/golden/synth.py seed 0, the cfg2 golden's inputs);
  N>1   weak scaling: every rank owns a 100k-row shard of an N*100k pool; one recombination over
        the whole pool, one small all-reduce (RCCL) per level.
Rank 0 prints ONE JSON line.  `value` = candidates reduced per second, whole job.
`roofline` is for the dominant kernel (k_level_reduce): algorithmic FP64 flop (SURVEY.md 8d:
(2d + 2 + C_k) per kernel entry, C_k = 28 for the software FP64 exp) / HIP-event time of the
launches, against the FP64 peak.  `cpu_baseline` = the oracle (a torch-CPU port of the reference's
own arithmetic, reference-shaped: it materialises the (E, M, S) tensor) on this box's host cores.
"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import sober_amd  # noqa: E402
from tests.golden.synth import SEED_CALL, build_spec, synth  # noqa: E402

CFG2 = dict(kind="rbf", mode="predictive_covariance", N=100000, M=500, d=10, b=100, n_obs=200, seed=0)
FP64_PEAK_TFLOPS = 78.6          # MI355X FP64 vector = FP64 matrix (vendor)
CK = {"rbf": 28, "matern52": 40}
PMC_TRAFFIC_BYTES_PER_LAUNCH = 23.4e6   # measured, see the comment at "traffic" below


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    # one rank per GPU over RCCL.  (Test hook: SOBER_BENCH_BACKEND=gloo lets several ranks share the GPUs that
    # exist -- the N > 1 code path on a one-GPU box; never used for a measurement.)
    backend = os.environ.get("SOBER_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    group = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    case = dict(CFG2)
    case["seed"] = CFG2["seed"] + rank                 # every rank synthesises its own shard
    inp = synth(case)
    base = synth(CFG2) if rank else inp                # X_nys / X_obs come from rank 0's stream
    spec = build_spec(CFG2, base)
    ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache,
                              spec.noise, spec.mean_const, spec.alpha)
    kernel = sober_amd.Kernel(ks, CFG2["mode"])
    sober_amd.setting_parameters(device=dev, dtype=torch.double)
    sampler = sober_amd.RecombinationSampler(kernel)

    X_cand = t(inp["X_cand"]).to(dev)
    X_nys = t(base["X_nys"]).to(dev)
    mu0 = t(inp["mu0"] / world).to(dev)                # global weights sum to 1
    mu = mu0.clone()
    N_loc, b = CFG2["N"], CFG2["b"]

    from sober_amd._ops_hip import HipOps
    ops = HipOps(dev)
    timers = {}

    def step(tm=None):
        mu.copy_(mu0)
        torch.manual_seed(SEED_CALL)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return sober_amd.recombination(X_cand, X_nys, b, kernel, dev, torch.double, init_weights=mu,
                                           group=group, row_offset=rank * N_loc, _ops=ops, _timers=tm)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    # the warm-up steps run exactly what the timed steps run, event brackets included (their first use costs
    # tens of ms in a fresh process)
    ops.prof = []
    ops.prof_reserve(16 * (args.steps + args.warmup + 1))    # event pairs for the level_reduce launches, created up front
    idx, w = step()                                      # initialisation (library load, workspaces, first-use paths)
    import gc
    gc.collect(); gc.disable()                           # no collector pauses inside the timed region
    for _ in range(args.warmup):
        idx, w = step()
    torch.cuda.synchronize()
    ops.prof = []
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    per_step = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        idx, w = step(timers)
        per_step.append(time.perf_counter() - ts)          # (host clock at return: the result tensors are final)
    torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # dominant kernel: HIP-event time of every k_level_reduce launch in the timed region, minus the
    # cost of an empty event pair on the same stream (calibrated here; ~5 us, comparable to the deep
    # levels' launches -- without it the event sum would not agree with rocprofv3's kernel durations)
    prof, ops.prof = ops.prof, None
    cal = []
    from sober_amd import _native as nat
    ops.prof_reserve(200)
    st = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(200):                               # same mechanism as the executor's brackets
        e0, e1 = ops._prof_pair()
        nat.record_event_pair(e0, e1, st); cal.append((e0, e1))
    torch.cuda.synchronize()
    ev_overhead = float(np.median([a.elapsed_time(b_) for a, b_ in cal]))
    kern_ms = sum(max(a.elapsed_time(b_) - ev_overhead, 0.0) for a, b_, _ in prof)
    entries = sum(e for _, _, e in prof)
    flop_per_entry = 2 * CFG2["d"] + 2 + CK[CFG2["kind"]]
    achieved = entries * flop_per_entry / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0

    if rank != 0:
        return

    # parity of the timed configuration against the reference golden (N=1 only)
    parity = None
    gpath = os.path.join(ROOT, "tests", "golden", "recomb_cfg2_rbf.npz")
    if world == 1 and os.path.exists(gpath):
        z = np.load(gpath)
        same = bool(np.array_equal(idx.cpu().numpy(), z["idx"]))
        relw = float(np.max(np.abs(w.cpu().numpy() - z["w"]) / np.abs(z["w"]))) if same else None
        parity = {"idx_equal_reference": same, "max_rel_w_vs_reference": relw}

    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import sober_oracle as O
        Xc, Xn = t(inp["X_cand"]), t(inp["X_nys"])
        ok = O.Kernel(spec, CFG2["mode"])

        def cpu_step():
            m = t(inp["mu0"].copy())
            torch.manual_seed(SEED_CALL)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                return O.recombination(Xc, Xn, b, ok, init_weights=m)
        # the reference-shaped CPU path barely scales with cores (memory-bound temporaries): time it at the
        # host's full thread count AND at 8 threads, report the faster one
        runs = {}
        for th in sorted({os.cpu_count() // 2 or 1, 8}):
            torch.set_num_threads(th)
            cpu_step()                                        # warm-up
            c0 = time.perf_counter()
            for _ in range(args.cpu_steps):
                cpu_step()
            runs[th] = (time.perf_counter() - c0) / args.cpu_steps
        best = min(runs, key=runs.get)
        cpu_s = runs[best]
        cpu_baseline = {"value": N_loc / cpu_s, "unit": "candidates/s", "cores": best,
                        "kind": "port", "ms_per_step": cpu_s * 1e3,
                        "ms_per_step_by_threads": {str(k): v * 1e3 for k, v in runs.items()},
                        "sample": f"{args.cpu_steps} full recombination steps of the same workload "
                                  f"(N_rec=100k, N_nys=500, d=10, batch=100) after 1 warm-up, oracle "
                                  f"(torch CPU FP64, reference-shaped), best of {sorted(runs)} threads"}

    ms_per_step = elapsed / args.steps * 1e3
    out = {
        "metric": "recombination-step candidates/sec (N_rec=100k, N_nys=500, d=10, batch=100)",
        "value": world * N_loc / (elapsed / args.steps),
        "unit": "candidates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "Ackley-shaped d=10 RBF posterior covariance, N_rec=100k per GPU, "
                               "N_nys=500, batch=100, n_obs=200 (BASELINE.json configs[1])",
                   "parallelism": f"pool row-sharded x{world}, one all-reduce of (n*S+S) f64 per level"
                                  if world > 1 else "single GPU"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP64_PEAK_TFLOPS,
                     # HBM bytes per launch (mean over the 11 launches of a step) from rocprofv3 PMC passes of
                     # THIS workload: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction of
                     # MI355X_MICROARCH.md; separate --pmc passes; profiles/r01_pmc_level_reduce.csv).
                     # Algorithmic bytes are ~1.7 MB/launch: the excess is the per-chunk partial-sum
                     # buffers (n_chunks * Mtot * S * 8 B written, then re-read by k_sum_partials).
                     "traffic": PMC_TRAFFIC_BYTES_PER_LAUNCH if world == 1 else None,
                     "kernel": "k_level_reduce_mfma", "launches": len(prof), "kernel_ms_per_step": kern_ms / args.steps,
                     "event_pair_overhead_ms": ev_overhead,
                     "note": "FP64 compute-bound: -|x-y|^2/2 on v_mfma_f64_16x16x4 (augmented GEMM), "
                             "table-driven FP64 exp on the VALU; MI355X FP64 vector and matrix peaks are "
                             "both 78.6 TFLOP/s and share the DP units (scripts/fp64_pipes_probe.hip); "
                             f"algorithmic flop = entries * (2d+2+C_k) = entries * {flop_per_entry}; all "
                             "launches of a step are averaged (level 0 alone runs ~2x the average)"},
        "cpu_baseline": cpu_baseline,
        "parity": parity,
        "phases_ms_per_step": {k: v / args.steps * 1e3 for k, v in timers.items()},
        "ms_each_step": [round(v * 1e3, 3) for v in per_step],
        "n_selected": int(idx.numel()),
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
