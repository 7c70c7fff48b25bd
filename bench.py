#!/usr/bin/env python
"""bench.py -- the headline benchmark of BASELINE.json on MI355X.

Workload (config 3 / 4): multipathfinder hot path, npaths = 64, d = 1000 correlated Gaussian
(low-rank + diagonal true covariance, r = 8), history_length = 6, ndraws_elbo = 1000, ndraws = 1000.
One "step" = one pass of the hot path over the batch of 64 optimisation traces that are already
resident in HBM:  fit_batch (L-BFGS inverse-Hessian reconstruction + Woodbury factorisation of every
trace point) -> elbo_batch (1000 draws x every fit, logq, logp, mean/SE, argmax per path) ->
pool_build (draws of the winning fits + log ratios) -> [RCCL all-gather of the log ratios] -> PSIS ->
resample indices -> gather of the selected columns [-> xGMI all-reduce].

value = ELBO draws per second of the whole job = (sum over fits of ndraws_elbo) / step time (the step time
includes fit, PSIS and resampling, i.e. it is also the hot-path part of the multipathfinder wall-clock).

Run:  python bench.py [--gpus N --steps K --warmup W]
A step is ENQUEUED as one pipeline (round 3): fit -> scan -> reduce / argmax -> winners picked on the device -> pool -> [all-gather] ->
PSIS -> index selection -> owner gather [-> all-reduce] with ONE host synchronisation at the end (pfmi_*_enqueue,
pfmi_pool_build_best, pfmi_comm_psis_resample); the results every step downloads: elbo / se / best_iter, k-hat, indices, draws.
N > 1: one rank per GPU (or --single-process: ONE process drives all N GPUs through pfmi_comm_init_all).  Either the caller launches the ranks (python -m torch.distributed.run ... bench.py --gpus N:
WORLD_SIZE is then set and must equal N), or bench.py launches them itself: `python bench.py --gpus N` without
WORLD_SIZE in the environment re-executes itself under torch.distributed.run on 127.0.0.1.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# The engine's streaming pipeline runs the optimiser, the fits and two scan launches on four streams beside the context's own; torch and
# RCCL create streams too.  With the runtime's default of 4 hardware queues they share queues and serialise one another (measured in the
# one-rank-per-GPU mode: packed route 5.0 -> 6.1 ms per step at 8 paths); 8 queues keep them apart.  Must be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "pathfinder.jl_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402


def _blocks(K, G):
    """contiguous blocks of paths, one per GPU, the first K % G one path longer (any npaths over any GPU count, like the reference's
    nruns over tasks, src/multipath.jl:131-146); returns the G + 1 block boundaries"""
    if K < G:
        sys.exit(f"bench.py: npaths={K} is smaller than the number of GPUs {G}")
    base, rem = divmod(K, G)
    b = [0]
    for g in range(G):
        b.append(b[-1] + base + (1 if g < rem else 0))
    return b


def _self_launch(ngpus):
    """`python bench.py --gpus N` outside a torch.distributed.run world: launch the N ranks (one per GPU) ourselves and
    relay their output; returns the launcher's exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _launch_only(world, rank):
    """PFMI_BENCH_LAUNCH_ONLY=1 (CPU test of the launcher, tests/test_bench_launch.py): every rank joins a `gloo` world,
    one all-reduce counts the ranks, rank 0 prints the JSON skeleton.  No engine, no GPU."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.ones(1, dtype=torch.float64)
    dist.all_reduce(t)
    n = dist.get_world_size()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_only": True, "n_gpus": world, "ranks_in_collective": int(t[0]), "world_size": n}), flush=True)


def _demo_device_target(pfmi, tg):
    """`tg` as a PFMI_TARGET_DEVICE_CALLBACK closure: the user-side HIP kernel of examples/device_logp reads the draws the library
    materialises in HBM (the route an arbitrary device-resident `logp` takes; the built-in targets never form x)."""
    import ctypes as C
    pfmi.lib()
    L = C.CDLL(os.environ.get("PFMI_DEMO_CLOSURE_LIB") or os.path.join(ROOT, "examples", "device_logp", "liblogp_demo.so"))
    dp = C.POINTER(C.c_double)
    if tg.kind == 1:
        return pfmi.DeviceCallbackTarget(tg.d, C.cast(L.pfx_funnel_logp, C.c_void_p).value, None, host=tg, keepalive=L)
    L.pfx_gauss_create.restype = C.c_void_p
    L.pfx_gauss_create.argtypes = [C.c_int32, C.c_int32, dp, dp, dp, dp, C.c_double]
    h = L.pfx_gauss_create(tg.d, tg.r, tg.mean.ctypes.data_as(dp), tg.a.ctypes.data_as(dp),
                           tg.Wd.ctypes.data_as(dp) if tg.r else None, tg.G.ctypes.data_as(dp) if tg.r else None, tg.offset)
    if not h:
        raise RuntimeError("pfx_gauss_create failed")
    return pfmi.DeviceCallbackTarget(tg.d, C.cast(L.pfx_gauss_logp, C.c_void_p).value, C.c_void_p(h), host=tg, keepalive=(L, h))


def _pmc_traffic(argv_tail, kernels):
    """HBM bytes per launch of the kernels in `kernels` ({label: SQL LIKE pattern}), measured IN THIS RUN: one step of this very command
    (+ one device-closure scan) re-run twice under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `... WRITE_SIZE` (separate passes: the
    TCC block cannot hold both counters), units and the gfx950 correction as /opt/skills/guides/MI355X_MICROARCH.md prescribes (KiB;
    FETCH_SIZE reports half of a wide coalesced read).  Returns ({label: {...}}, None) or (None, reason)."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    raw = {}
    tmp = tempfile.mkdtemp(prefix="pfmi_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            dd = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "-d", dd, "-o", "p", "--", sys.executable, os.path.abspath(__file__)] + argv_tail + \
                  ["--steps", "1", "--warmup", "0", "--minimal", "--with-devcb", "--no-cpu-baseline", "--no-pmc"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
            dbs = glob.glob(os.path.join(dd, "**", "*_results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {r.returncode}): {r.stderr[-300:]}"
            con = sqlite3.connect(dbs[0])
            for label, like in kernels.items():
                rows = list(con.execute(
                    "select grid_size, sum(value), count(distinct dispatch_id), avg(duration) from counters_collection "
                    "where kernel_name like ? and counter_name = ? group by grid_size order by sum(value) desc", (like, ctr)))
                if rows:
                    g, v, n, dur = rows[0]                          # the launch shape that moves the most bytes (the main launch)
                    raw.setdefault(label, {})[ctr] = (v / n * 1024.0, int(n), dur / 1e6)
            con.close()
    except Exception as ex:
        return None, repr(ex)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for label, rr in raw.items():
        if "FETCH_SIZE" not in rr or "WRITE_SIZE" not in rr:
            continue
        fetch, write = 2.0 * rr["FETCH_SIZE"][0], rr["WRITE_SIZE"][0]
        out[label] = {"traffic_bytes_per_launch": fetch + write, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                      "fetch_correction": 2.0, "launches_profiled": rr["FETCH_SIZE"][1], "avg_duration_ms_under_pmc": round(rr["FETCH_SIZE"][2], 3),
                      "source": "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, two separate passes of one step"}
    if not out:
        return None, "no counter rows for the requested kernels"
    return out, None


def _single_gpu_reference(pfmi, tg, x0_all, run_seeds, J, maxiters, N_e, N_r, ndraws, master, device, free_bytes=None):
    """ALL K paths on ONE context (a world of one: no collective), the answer a sharded run must reproduce bit for bit
    (test/multipath.jl:107-140 extended from `ntasks` to the GPU count).  Returns (fingerprint | None, note)."""
    from pfmi.distributed import result_fingerprint
    from pfmi.hostrng import rand_u64
    K, d = x0_all.shape
    kc = next(o for o in (4, 8, 12, 16, 20, 32) if 2 * J <= o)
    need = 8.0 * d * (K * (maxiters + 1) * (kc + 9))               # factors + alpha / sqrt(alpha) / mu + traces and their staging, upper bound
    if free_bytes is not None and need > 0.8 * free_bytes:
        return None, f"all {K} paths need up to {need / 1e9:.0f} GB on one GPU ({free_bytes / 1e9:.0f} GB free): not verifiable on one device"
    e = pfmi.Engine(device)
    try:
        e.set_target(tg)
        npts = e.optimize_batch(x0_all, J, maxiters)
        seeds = np.concatenate([rand_u64(int(run_seeds[k]), np.arange(n, dtype=np.uint64), 10) for k, n in enumerate(npts)])
        e.fit_batch(J)
        e.elbo_batch_enqueue(N_e, seeds)
        e.pool_build_best(N_r)
        cm = pfmi.Comm.init_all([e])
        res, idx, draws = cm.psis_resample(ndraws, seed=master)
        e.elbo_batch_wait()
        cm.close()
        return result_fingerprint(res["pareto_shape"], res["tail_length"], idx, draws), f"all {K} paths recomputed on GPU {device} (no collective)"
    finally:
        e.close()


def main_single_process(args):
    """--gpus N --single-process: ONE host process (thread) drives N contexts, one per GPU -- what a single Julia caller of
    multipathfinder does (north star).  pfmi_comm_init_all; every stage is enqueued on all GPUs before the first wait."""
    import pfmi
    from pfmi.hostrng import rand_u64
    K, d, N_e, J, ndraws, G = args.npaths, args.dim, args.ndraws_elbo, args.history, args.ndraws, args.gpus
    bk = _blocks(K, G)
    N_r = max(N_e, -(-ndraws // K))
    master = 20260928
    tg = {"lowrank": lambda: pfmi.t_lowrank(d, r=8, seed=2), "diag": lambda: pfmi.t_diag(d, seed=1), "iso": lambda: pfmi.t_iso(d),
          "funnel": lambda: pfmi.t_funnel(d)}[args.target]()
    run_seeds = rand_u64(master, np.arange(K, dtype=np.uint64), 9)
    sc = args.init_scale
    x0 = np.stack([pfmi.HostRNG(int(s)).rand(d) * 2 * sc - sc for s in run_seeds])
    import ctypes as C
    ndev = C.c_int32()
    pfmi.lib().pfmi_device_count(C.byref(ndev))
    engs = [pfmi.Engine(g % max(ndev.value, 1)) for g in range(G)]     # fewer GPUs than ranks: test hook (PFMI_COMM_ALLOW_SHARED_GPU)
    for e in engs:
        e.set_target(tg)
    for g, e in enumerate(engs):
        e.optimize_batch_enqueue(x0[bk[g]:bk[g + 1]], J, args.maxiters)
    seeds = []
    for g, e in enumerate(engs):
        npts = e.optimize_batch_wait()
        seeds.append(np.concatenate([rand_u64(int(run_seeds[bk[g] + i]), np.arange(n, dtype=np.uint64), 10) for i, n in enumerate(npts)]))
    comm = pfmi.Comm.init_all(engs)
    info = comm.info()
    total_draws = float(sum((e.P - (bk[g + 1] - bk[g])) * N_e for g, e in enumerate(engs)))
    state = {}

    def step():
        for e, sd in zip(engs, seeds):
            e.fit_batch(J)
            e.elbo_batch_enqueue(N_e, sd)
            e.pool_build_best(N_r)
        res, idx, state["draws"] = comm.psis_resample(ndraws, seed=master)
        state["best"] = [e.elbo_batch_wait()[2] for e in engs]
        state.update(pareto_k=res["pareto_shape"], tail=res["tail_length"], idx=idx)

    # the end-to-end call as pfmi.multipathfinder(engines=[...]) makes it: ONE host thread schedules G streaming pipelines (optimise + fit +
    # scan per engine, pfmi_stream_*) and the pooled stage.  What that thread spends per step is measured on the HOST clock, split into
    # the calls that enqueue, the scheduling passes that launched a segment (counted inside libpfmi: "stream_host_schedule"; the idle
    # polls between them cost nothing but the thread's own time) and the pooled stage's enqueue -- VERDICT r5 weak #7: the per-pump cost
    # x G against a ~4 ms step.  (With the G contexts on fewer GPUs than G -- the stand-in test box -- the GPU side is G times slower
    # than a real node; the host figures are what the host does either way.)
    cap = args.maxiters + 1
    seed_tabs = [np.concatenate([rand_u64(int(run_seeds[k]), np.arange(1, cap + 1, dtype=np.uint64), 10) for k in range(bk[g], bk[g + 1])])
                 for g in range(G)]
    host = {"enqueue_ms": 0.0, "pump_loop_ms": 0.0, "pooled_enqueue_ms": 0.0, "wait_ms": 0.0, "steps": 0}

    def step_streamed():
        t0 = time.perf_counter()
        for g, e in enumerate(engs):
            e.stream_enqueue(x0[bk[g]:bk[g + 1]], N_e, seed_tabs[g], J, args.maxiters)
        t1 = time.perf_counter()
        active = list(engs)
        while active:
            active = [e for e in active if not e.stream_pump()]
        for e in engs:
            e.stream_wait()
        t2 = time.perf_counter()
        for e in engs:
            e.pool_build_best(N_r)
        comm.psis_resample_enqueue(ndraws, seed=master)
        tabs = []
        for e in engs:
            e.defer(1)
            tabs.append(e.elbo_batch_wait())
            e.defer(0)
        t3 = time.perf_counter()
        res, idx, state["draws_s"] = comm.psis_resample_wait()
        t4 = time.perf_counter()
        state.update(pareto_k_s=res["pareto_shape"], idx_s=idx)
        host["enqueue_ms"] += (t1 - t0) * 1e3; host["pump_loop_ms"] += (t2 - t1) * 1e3
        host["pooled_enqueue_ms"] += (t3 - t2) * 1e3; host["wait_ms"] += (t4 - t3) * 1e3; host["steps"] += 1

    for _ in range(args.warmup):
        step()
    for e in engs:
        e.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    for e in engs:
        e.sync()
    ms_per_step = (time.perf_counter() - t0) / args.steps * 1e3
    host_line = None
    if not args.host_traces:
        import gc
        try:
            for _ in range(2):
                step_streamed()
            for k_ in host:
                host[k_] = 0.0 if k_ != "steps" else 0
            for e in engs:
                e.profile(2)
            gc.collect(); gc.disable()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step_streamed()
            for e in engs:
                e.sync()
            e2e = (time.perf_counter() - t0) / args.steps * 1e3
            gc.enable()
            sched = [e.kernel_time("stream_host_schedule") for e in engs]
            for e in engs:
                e.profile(0)
            n = max(host["steps"], 1)
            launching = sum(ms_ for ms_, _ in sched) / n
            host_line = {"step_ms_end_to_end_streamed": round(e2e, 3),
                         "host_schedule_ms_per_step": round(host["enqueue_ms"] / n + launching + host["pooled_enqueue_ms"] / n, 4),
                         "stream_enqueue_calls_ms": round(host["enqueue_ms"] / n, 4),
                         "segment_launching_passes_ms": round(launching, 4), "segment_launches_per_step": round(sum(c_ for _, c_ in sched) / n, 1),
                         "pooled_stage_enqueue_ms": round(host["pooled_enqueue_ms"] / n, 4),
                         "pump_loop_wall_ms": round(host["pump_loop_ms"] / n, 4), "final_wait_ms": round(host["wait_ms"] / n, 4),
                         "streamed_equals_packed": bool(state["pareto_k_s"] == state["pareto_k"] and np.array_equal(state["idx_s"], state["idx"])
                                                        and np.array_equal(state["draws_s"], state["draws"])),
                         "note": "ONE host thread drives all contexts: host_schedule = stream_enqueue calls + the scheduling passes that launched a segment "
                                 "(timed inside libpfmi) + the pooled stage's enqueue; pump_loop_wall is the span during which the thread polls (it "
                                 "overlaps the GPUs' work)"}
        except Exception as ex:  # pragma: no cover
            host_line = {"error": repr(ex)}
            for e in engs:
                e.stream_cancel()
    verdict, vnote = None, "not requested"
    if args.verify_sharding or (G > 1 and not args.no_verify_sharding):
        from pfmi.distributed import result_fingerprint, sharded_equals_single
        ref, vnote = _single_gpu_reference(pfmi, tg, x0, run_seeds, J, args.maxiters, N_e, N_r, ndraws, master, engs[0].device)
        verdict, bad = sharded_equals_single(None, result_fingerprint(state["pareto_k"], state["tail"], state["idx"], state["draws"]), ref)
        if bad:
            vnote += f"; MISMATCH in {bad}"
    line = {"metric": "ELBO draws/sec (multipathfinder hot path: fit + ELBO + pool + PSIS + resample)",
            "value": round(total_draws / (ms_per_step * 1e-3), 1), "unit": "ELBO draws/s", "n_gpus": G, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"multipathfinder npaths={K} d={d} target={args.target}, history_length={J}, ndraws_elbo={N_e}, ndraws={ndraws}",
                       "npaths": K, "paths_per_gpu": [bk[g + 1] - bk[g] for g in range(G)], "fits_total": int(total_draws // N_e),
                       "elbo_draws_per_step": int(total_draws),
                       "parallelism": f"paths sharded x{G}, ONE host process / thread (pfmi_comm_init_all)",
                       "ranks_in_collective": info["world"], "rccl_version": info["rccl_version"]},
            "pareto_k": state.get("pareto_k"), "sharded_equals_single": verdict, "sharded_equals_single_note": vnote,
            "single_thread_scheduler": host_line,
            "rccl_version": info["rccl_version"], "roofline": None, "cpu_baseline": None}
    comm.close()
    for e in engs:
        e.close()
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--npaths", type=int, default=64)
    ap.add_argument("--dim", type=int, default=1000)
    ap.add_argument("--ndraws-elbo", type=int, default=1000)
    ap.add_argument("--ndraws", type=int, default=1000)
    ap.add_argument("--history", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--host-traces", action="store_true", help="make the input traces with the host numpy driver")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-GPU code path (RCCL all-gather / all-reduce) even in a single-rank world")
    ap.add_argument("--target", default="lowrank", choices=["lowrank", "diag", "iso", "funnel"],
                    help="synthetic target of SURVEY 8(d); the headline config uses lowrank (r = 8)")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N driven by ONE host process (N contexts, pfmi_comm_init_all) instead of one rank per GPU")
    ap.add_argument("--minimal", action="store_true", help="timed steps only (the rocprofv3 counter passes re-run bench.py this way)")
    ap.add_argument("--with-devcb", action="store_true", help="with --minimal: also one device-closure scan (counter passes)")
    ap.add_argument("--no-pmc", action="store_true", help="do not re-run a step under rocprofv3 --pmc for roofline.traffic")
    ap.add_argument("--verify-sharding", action="store_true",
                    help="after the timed steps rank 0 recomputes ALL paths on one GPU and every rank compares k-hat, the indices and a hash "
                         "of the d x ndraws result with its sharded answer (default when --gpus > 1; this flag forces it at --gpus 1)")
    ap.add_argument("--no-verify-sharding", action="store_true")
    ap.add_argument("--maxiters", type=int, default=1000)
    ap.add_argument("--init-scale", type=float, default=2.0)
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.single_process:
        return main_single_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the torch.distributed.run world must have one rank "
                 "per requested GPU")
    if os.environ.get("PFMI_BENCH_LAUNCH_ONLY") == "1":
        return _launch_only(world, int(os.environ.get("RANK", "0")))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import pfmi
    from pfmi.hostrng import rand_u64

    K, d, N_e, J, ndraws = args.npaths, args.dim, args.ndraws_elbo, args.history, args.ndraws
    G = world
    bk = _blocks(K, G)
    k0, Kl = bk[rank], bk[rank + 1] - bk[rank]
    N_r = max(N_e, -(-ndraws // K))                                 # reference src/multipath.jl:138
    master = 20260928

    # ---- synthetic inputs: target T_lr(d, r=8, seed=2), one L-BFGS trace per path ----------
    tg = {"lowrank": lambda: pfmi.t_lowrank(d, r=8, seed=2), "diag": lambda: pfmi.t_diag(d, seed=1), "iso": lambda: pfmi.t_iso(d),
          "funnel": lambda: pfmi.t_funnel(d)}[args.target]()
    run_seeds = rand_u64(master, np.arange(K, dtype=np.uint64), 9)
    sc = args.init_scale                                             # U[-2, 2] by default (src/singlepath.jl:158-159)
    x0s = np.stack([pfmi.HostRNG(int(run_seeds[k])).rand(d) * 2 * sc - sc for k in range(k0, k0 + Kl)])
    eng = pfmi.Engine(local_rank)
    eng.set_target(tg)
    traces = None
    if args.host_traces:
        traces = [pfmi.optimize_with_trace(tg, x0, history_length=J, maxiters=args.maxiters) for x0 in x0s]
        eng.set_traces([t.points for t in traces], [t.gradients for t in traces])  # H2D: outside the timed region
        npts = np.array([len(t) for t in traces])
    else:                                                           # device L-BFGS: traces are born in HBM
        npts = eng.optimize_batch(x0s, J, args.maxiters)
    P = eng.P
    seeds = np.concatenate([rand_u64(int(run_seeds[k0 + i]), np.arange(n, dtype=np.uint64), 10)
                            for i, n in enumerate(npts)])
    nfits_local = P - Kl
    draws_local = nfits_local * N_e

    comm = None
    if not use_dist:
        comm = pfmi.Comm.init_all([eng])                            # a world of one context: same entry points, no RCCL involved
    if use_dist:
        # the data-path collectives go through the C ABI (pfmi_comm_*: ncclAllGather of the log-ratio shards, ncclSend / ncclRecv of the selected
        # columns to rank 0, on the engine's stream, csrc/comm_rccl.hip); torch.distributed only ships the 128-byte RCCL id, the barrier and
        # the timing reduction.  There is no second data path: if the group cannot be formed the run fails on every rank.
        comm_note = "RCCL through the C ABI (pfmi_comm_*), torch nccl group for barrier / timing"
        uid = [None]
        if rank == 0:
            try:
                uid = [pfmi.Comm.unique_id()]
            except Exception as ex0:                            # every rank must still reach the broadcast
                uid = [("error", repr(ex0))]
        dist.broadcast_object_list(uid, src=0)
        if not isinstance(uid[0], (bytes, bytearray)):
            sys.exit(f"bench.py: rank 0 could not create an RCCL id: {uid[0]}")
        comm = pfmi.Comm.init_rank(eng, world, rank, uid[0])
        info = comm.info()
        assert info["world"] == world, info

    state = {}
    self_check_failures = []

    # the streaming pipeline (pfmi_stream_enqueue) draws maxiters + 1 seeds per run up front; fit l of a run takes value l - 1 of that stream:
    # the same seeds as `seeds` above (whose entry l of a run is counter l)
    cap = args.maxiters + 1
    seed_tab = np.concatenate([rand_u64(int(run_seeds[k0 + i]), np.arange(1, cap + 1, dtype=np.uint64), 10) for i in range(Kl)])

    def step(streamed=False):
        # everything up to the last line of the `comm` branch only ENQUEUES work on the engine's stream
        if streamed:
            # optimise + fit + scan as one dataflow: the calling thread launches each segment of trace positions as soon as every path has
            # produced it (csrc/pfmi_api.hip: pfmi_stream_pump); stream_wait returns when the last segment is OUT, not when it is done
            eng.stream_enqueue(x0s, N_e, seed_tab, J, args.maxiters)
            eng.stream_wait()
        else:
            eng.fit_batch(J)
            eng.elbo_batch_enqueue(N_e, seeds)
        eng.pool_build_best(N_r)                                    # winners (fit_iteration per path) picked on the device
        if comm is not None:
            # [ONE RCCL all-gather of the log-ratio shards] + replicated PSIS + replicated indices + owner-only transfer of the selected
            # columns (one process per GPU: ncclSend / ncclRecv to rank 0, which alone holds the d x ndraws result), D2H of k-hat / indices / draws
            comm.psis_resample_enqueue(ndraws, seed=master)
            eng.defer(1)                                            # the ELBO tables' downloads are queued behind the pooled stage ...
            elbo, se, best = eng.elbo_batch_wait()
            eng.defer(0)
            res, idx, state["draws"] = comm.psis_resample_wait(want_draws=(rank == 0))    # ... and this ONE wait delivers everything
            state.update(elbo=elbo, best=best, pareto_k=res["pareto_shape"], tail=res["tail_length"], idx=idx)
            return
        raise RuntimeError("no communicator")

    def barrier():
        eng.sync()
        if use_dist:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    # Like timeit: no cyclic garbage collection inside the timed loops.  A full collection of this process (~50 ms: every ~120 steps at 8 paths)
    # stops the host, and in the streamed call the host is the scheduler of the pipeline: one such step took 46.5 ms instead of 3.9
    # (profiles/r05_experiments.md section 5).  Nothing here creates reference cycles; memory is reclaimed by reference counting as before.
    import gc
    gc.collect()
    gc.disable()
    for _ in range(args.warmup):
        step()
    barrier()
    # stage timers of the TIMED steps: hipEvent pairs left in the engine's stream (pfmi_profile mode 2: no host synchronisation, the
    # pipeline runs exactly as it does unprofiled; ~20 event records per step on the host, hidden behind the GPU work), read afterwards
    timed_stages = rank == 0 and not args.minimal
    if timed_stages:
        eng.profile(2)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    stages, scan_ms, scan_n = {}, 0.0, 0
    if timed_stages:
        for name in ("history", "fit", "elbo_draws", "elbo_draws_x", "elbo_reduce", "psis", "resample", "comm_handshake_host", "comm_allgather",
                     "comm_sendrecv"):
            ms_, n_ = eng.kernel_time(name)
            stages[name] = {"ms": round(ms_ / max(n_, 1), 4), "launches": int(n_),      # average per launch
                            "ms_per_step": round(ms_ / max(args.steps, 1), 4)}
        scan_ms, scan_n = eng.kernel_time("elbo_draws")                                # the ELBO-scan launches only (total, count)
        eng.profile(2 if not args.host_traces else 0)                                 # reset: the end-to-end loop adds "optimize"
    if use_dist:
        import torch
        tt = torch.tensor([dt, float(draws_local), 1.0], dtype=torch.float64, device=f"cuda:{local_rank}")
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        total_draws = float(tt[1])
        ranks_seen = int(round(float(tt[2])))                           # ranks that took part in the RCCL all-reduce
        assert ranks_seen == dist.get_world_size() == world, (ranks_seen, dist.get_world_size(), world)
    else:
        total_draws = float(draws_local)
        ranks_seen = 1
    ms_per_step = dt / args.steps * 1e3
    value = total_draws / (ms_per_step * 1e-3)

    # ---- self-verification of the sharded run: the first real multi-GPU run says whether it is CORRECT, not only how
    #      fast.  Rank 0 recomputes all K paths on its own GPU (a world of one context, no collective); every rank compares k-hat, the
    #      tail length, the resample indices and a hash of the d x ndraws result of ITS copy of the sharded answer with it.
    verdict, vnote, rccl_version = None, "not requested", None
    if comm is not None:
        try:
            rccl_version = comm.info()["rccl_version"]
        except Exception:
            pass
    if (args.verify_sharding or (world > 1 and not args.no_verify_sharding)) and not args.minimal and not args.host_traces:
        from pfmi.distributed import result_fingerprint, sharded_equals_single
        mine = result_fingerprint(state["pareto_k"], state["tail"], state["idx"], state["draws"])    # (draws: rank 0 only, the result lives there)
        ref = None
        if rank == 0:
            x0_all = np.stack([pfmi.HostRNG(int(run_seeds[k])).rand(d) * 2 * sc - sc for k in range(K)])
            free_b = None
            try:
                import torch
                free_b = torch.cuda.mem_get_info(local_rank)[0]
            except Exception:
                pass
            try:
                ref, vnote = _single_gpu_reference(pfmi, tg, x0_all, run_seeds, J, args.maxiters, N_e, N_r, ndraws, master, local_rank, free_b)
            except Exception as ex:                                     # the other ranks are waiting in the broadcast: never raise here
                ref, vnote = None, f"reference run failed: {ex!r}"
        dev = None
        if use_dist:
            import torch
            dev = torch.device("cuda", local_rank)
        verdict, bad = sharded_equals_single(dist if use_dist else None, mine, ref, device=dev)
        if bad:
            vnote = (vnote or "") + f"; rank {rank} MISMATCH in {bad}"

    # ---- metric (ii): end-to-end wall-clock incl. trajectory generation (x0 on the host -> resampled draws on the host)
    wall_e2e = wall_e2e_packed = None
    streamed_equal, stream_note = None, None
    if not args.host_traces and not args.minimal:
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.optimize_batch(x0s, J, args.maxiters)
            step()
        barrier()
        wall_e2e_packed = (time.perf_counter() - t0) / args.steps * 1e3
        wall_e2e = wall_e2e_packed
        if timed_stages:
            for name in ("optimize", "trace_pack"):
                ms_, n_ = eng.kernel_time(name)
                stages[name] = {"ms": round(ms_ / max(n_, 1), 4), "launches": int(n_)}
            eng.profile(0)
        ref_fp = (state["pareto_k"], state["idx"].copy(), np.array(state["draws"], copy=True) if isinstance(state["draws"], np.ndarray) else None,
                  state["best"].copy())
        state["elbo_packed"] = np.array(state["elbo"], copy=True)      # PACKED layout [P]: the streamed loop below leaves state["elbo"] in slot layout
        # the same job as ONE dataflow: fits and scans of the points a path has already produced run while the paths are still being optimised
        if comm is not None or not use_dist:
            try:
                for _ in range(2):
                    step(streamed=True)
                barrier()
                t0 = time.perf_counter()
                per_step = []
                for _ in range(args.steps):
                    t1 = time.perf_counter()
                    step(streamed=True)
                    per_step.append((time.perf_counter() - t1) * 1e3)
                barrier()
                wall_e2e = (time.perf_counter() - t0) / args.steps * 1e3
                if os.environ.get("BENCH_STEP_TIMES"):
                    print("streamed steps (ms):", " ".join(f"{t:.2f}" for t in per_step), file=sys.stderr)
                streamed_equal = bool(state["pareto_k"] == ref_fp[0] and np.array_equal(state["idx"], ref_fp[1]) and np.array_equal(state["best"], ref_fp[3])
                                      and (ref_fp[2] is None or np.array_equal(state["draws"], ref_fp[2])))
                stream_note = "pfmi_stream_enqueue / pfmi_stream_wait"
            except Exception as ex:                                     # (e.g. PFMI_ERR_UNSUPPORTED: no room for the fixed-stride layout)
                stream_note = f"streaming pipeline not used: {ex!r}"
            npts = eng.optimize_batch(x0s, J, args.maxiters)           # back to the packed layout for what follows

    # ---- the same job through the public host API (pfmi.multipathfinder: x0 sampling, device L-BFGS, fit, ELBO, pool, PSIS,
    #      resample, result objects), single GPU only
    api_wall = None
    if G == 1 and not use_dist and not args.host_traces and not args.minimal:
        ts = []
        for rep in range(7):                                            # 2 untimed calls (first-use allocations), median of 5
            t0 = time.perf_counter()
            # (the benchmark's own starting points: the same K paths, the same 11 278 fits at config 3, as the lines above)
            pfmi.multipathfinder(tg, ndraws, init=list(x0s), ndraws_elbo=N_e, history_length=J, rng=pfmi.HostRNG(master), engine=eng,
                                 init_scale=sc, maxiters=args.maxiters)
            ts.append((time.perf_counter() - t0) * 1e3)
        api_wall = sorted(ts[2:])[2]
        npts = eng.optimize_batch(x0s, J, args.maxiters)               # restore the benchmark's own traces for the profile step

    # ---- host-callback targets (the reference's general case: logp is an arbitrary host closure, src/elbo.jl:15): every draw has
    #      to cross PCIe, so this is a separate, much lower line.  Bounded sample: the first 2 paths of this run, a vectorised
    #      NumPy closure (-|x|^2 / 2); the draws of block i + 1 are generated / downloaded while the host evaluates block i.
    callback_line = None
    if G == 1 and not use_dist and not args.host_traces and not args.no_cpu_baseline and not args.minimal:
        try:
            import cpu_lapack_baseline as cl
            from helpers import demo_host_target
            aff, logical, quota = cl.usable_cores()
            cores = max(1, min(aff, int(quota)) if quota else aff)
            Kc = min(2, Kl)
            trs = [eng.get_trace(k, logp=False) for k in range(Kc)]
            e2 = pfmi.Engine(local_rank)

            def run_cb(target, nthreads):
                e2.set_target(target)
                e2.set_callback_threads(nthreads)
                e2.set_traces([t[0] for t in trs], [t[2] for t in trs])
                e2.fit_batch(J)
                sd = seeds[:e2.P]
                e2.elbo_batch(N_e, sd)                                  # warm-up: pinned staging is allocated here
                t0 = time.perf_counter()
                el = e2.elbo_batch(N_e, sd)[0]
                dtc = time.perf_counter() - t0
                st = e2.callback_stats()
                return {"draws_per_s": round((e2.P - Kc) * N_e / dtc, 1), "wall_s": round(dtc, 4), "callback_s": round(st["callback_seconds"], 4),
                        "pcie_GBps_device_to_host": round(st["bytes_to_host"] / dtc / 1e9, 2)}, el

            callback_line = {"sample": f"first {Kc} paths, {sum(len(t[0]) for t in trs) - Kc} fits x {N_e} draws, d={d}",
                             "note": "pinned staging, 64 MB blocks, generation + download of block i+1 overlap the host's evaluation of block i; "
                                     "callback_threads = pfmi_set_callback_threads (the reference's ntasks, src/elbo.jl:3-6): every staged block is cut "
                                     "into that many column ranges evaluated concurrently"}
            if getattr(tg, "kind", 1) == 0:                             # compiled host closure (examples/device_logp: plain C, the same target)
                ctg = demo_host_target(tg)
                c1, el1 = run_cb(ctg, 1)
                cn, eln = run_cb(ctg, cores)
                callback_line["compiled_closure"] = {"threads_1": c1, f"threads_{cores}": cn, "speedup": round(cn["draws_per_s"] / c1["draws_per_s"], 2),
                                                     "identical_results": bool(np.array_equal(el1, eln, equal_nan=True)),
                                                     "closure": "pfx_host_gauss_logp (scalar C, one column at a time like src/elbo.jl:15)"}
                if not np.array_equal(el1, eln, equal_nan=True):
                    self_check_failures.append("host-closure ELBO table depends on the number of callback threads")
            cbt = pfmi.CallbackTarget(d, lambda x: float(-0.5 * (x @ x)), logp_batch=lambda X: -0.5 * np.einsum("ij,ij->j", X, X))
            callback_line["numpy_closure"] = {"threads_1": run_cb(cbt, 1)[0], f"threads_{cores}": run_cb(cbt, cores)[0],
                                              "closure": "NumPy -|x|^2/2 on (d, n) blocks through ctypes (the GIL serialises what NumPy does not release)"}
            best_line = max((v for kk in ("compiled_closure", "numpy_closure") if kk in callback_line for k2, v in callback_line[kk].items()
                             if k2.startswith("threads_")), key=lambda v: v["draws_per_s"])
            callback_line.update(draws_per_s=best_line["draws_per_s"], pcie_GBps_device_to_host=best_line["pcie_GBps_device_to_host"], cores=cores)
            e2.close()
        except Exception as ex:  # pragma: no cover
            callback_line = {"error": repr(ex)}

    # ---- DEVICE-callback targets (round 3): the same arbitrary-closure contract, but the closure is a kernel -- the draws are
    #      materialised in HBM (8 d bytes written per draw) and read there by the user's kernel (8 d bytes read): the path on which
    #      SURVEY 8(d)'s 16 d bytes per draw PHYSICALLY move, so its HBM fraction is a real utilisation.  Bounded sample: 8 paths.
    devcb_line = None
    if G == 1 and not use_dist and not args.host_traces and (not args.minimal or args.with_devcb):
        try:
            Kd = min(8, Kl)
            trs = [eng.get_trace(k, logp=False) for k in range(Kd)]
            e3 = pfmi.Engine(local_rank)
            e3.set_target(_demo_device_target(pfmi, tg))
            e3.set_traces([t[0] for t in trs], [t[2] for t in trs])
            e3.fit_batch(J)
            sd = seeds[:e3.P]
            e3.elbo_batch(N_e, sd)                                      # warm-up: the block buffers are allocated here
            reps = 1 if args.minimal else 3
            e3.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                e3.elbo_batch_enqueue(N_e, sd)
            elbo_d = e3.elbo_batch_wait()[0]
            dtd = (time.perf_counter() - t0) / reps
            ndr = (e3.P - Kd) * N_e
            # the kernels' own times: one more scan on ONE stream (the timed scans above alternate their blocks between two streams, so that the
            # closure of block i runs beside the writer of block i + 1: there a kernel's event pair also spans its neighbour's work)
            pfmi.lib().pfmi_debug_set(b"PFMI_DEVCB_OVERLAP", b"0")
            e3.profile(2)
            e3.elbo_batch(N_e, sd)
            tw, nw = e3.kernel_time("elbo_draws_x")
            tr_, nr = e3.kernel_time("device_callback")
            e3.profile(0)
            e3.sync()
            t0 = time.perf_counter()
            e3.elbo_batch_enqueue(N_e, sd)
            elbo_1s = e3.elbo_batch_wait()[0]
            dt1 = time.perf_counter() - t0
            pfmi.lib().pfmi_debug_set(b"PFMI_DEVCB_OVERLAP", None)
            if not np.array_equal(elbo_1s, elbo_d, equal_nan=True):
                self_check_failures.append("device-closure ELBO table differs between the two-stream and the one-stream scan")
            moved = 16.0 * d * ndr
            # reference table in the PACKED layout (entry p = point p of pfmi_set_traces).  The streamed loop leaves state["elbo"] in the
            # fixed-stride slot layout k (maxiters + 1) + l, which round 5 sliced as if it were packed: the check went NaN unnoticed.
            ref_el = state.get("elbo_packed", state["elbo"])[:e3.P]
            assert len(ref_el) == e3.P == len(elbo_d), (len(ref_el), e3.P, len(elbo_d))
            fin = np.isfinite(ref_el)
            assert fin.sum() == e3.P - Kd and np.array_equal(fin, np.isfinite(elbo_d)), "device-closure ELBO table: finite pattern differs from the built-in target's"
            devcb_diff = float(np.max(np.abs(elbo_d[fin] - ref_el[fin]) / (1 + np.abs(ref_el[fin]))))
            devcb_line = {"draws_per_s": round(ndr / dtd, 1), "wall_s": round(dtd, 5),
                          "hbm_GBps": round(moved / dtd / 1e9, 1), "frac_of_8TBps_spec": round(moved / dtd / 8e12, 4),
                          "frac_of_6.29TBps_measured_copy": round(moved / dtd / 6.29e12, 4),
                          "bytes_per_draw_moved": 16.0 * d,
                          "writer_kernel": {"ms": round(tw, 3), "launches": int(nw), "GBps_written": round(8.0 * d * ndr / max(tw, 1e-9) / 1e6, 1)},
                          "reader_kernel": {"ms": round(tr_, 3), "launches": int(nr), "GBps_read": round(8.0 * d * ndr / max(tr_, 1e-9) / 1e6, 1)},
                          "max_rel_elbo_diff_vs_builtin_target": devcb_diff,
                          "one_stream": {"wall_s": round(dt1, 5), "hbm_GBps": round(moved / dt1 / 1e9, 1), "frac_of_6.29TBps_measured_copy": round(moved / dt1 / 6.29e12, 4),
                                         "identical_results": bool(np.array_equal(elbo_1s, elbo_d, equal_nan=True))},
                          "sample": f"first {Kd} paths, {e3.P - Kd} fits x {N_e} draws, d={d}; closure = examples/device_logp (HIP, same target)",
                          "note": "draws written to HBM by the library's draw kernel, read by the user's kernel; no PCIe.  Round 6: the blocks of fits alternate "
                                  "between two streams, so the closure of block i runs beside the writer of block i + 1 on the same CUs (one_stream: the "
                                  "same scan with PFMI_DEVCB_OVERLAP=0; writer_kernel / reader_kernel are timed there)"}
            # ---- the writer's own issue floor, priced like the scan's: nothing co-issues with the f64 MFMA on gfx950, so
            # the floor is the SUM of the issue streams, with the unit times of the co-issue micro-benchmark (2 waves per SIMD:
            # MFMA 4x4x4 7.44 ns, 16x16x4 = four times that, Philox round 10.5 ns, other VALU 2.0 ns).  Per wave and 16-row block of its TWO
            # 16-draw groups (instruction counts read off the ISA of pf_elbo_xw_kernel<KC>, interior bodies):
            #   pass 1  2 x 4 x KC/4 MFMA 4x4x4 (w += Vh'z), 2 Philox4x32-7, ~95 VALU (2 x 4 look-ups of 9, |u|^2, addresses)
            #   pass 2  2 x KC/4 MFMA 16x16x4 (x~ = z - Vh tv), 2 Philox4x32-7, ~105 VALU (look-ups, x = mu + sqrt(alpha) x~, store addresses)
            kc_w = next(o for o in (4, 8, 12, 16, 20, 32) if 2 * J <= o)
            nt_w, nblk_w = kc_w // 4, -(-d // 16)
            per_pair_us = ((2 * 4 * nt_w) * 7.44 + (2 * nt_w) * 4 * 7.44 + 28 * 10.5 + 200 * 2.0) * 1e-3
            walks = (e3.P - Kd) * (-(-(-(-N_e // 16)) // 16)) * 8          # fits x batches of 16 group slots x 8 waves, each walking nblk blocks twice
            w_floor_ms = per_pair_us * walks * nblk_w / (4.0 * 256) * 1e-3
            tw_launch = tw                                                # kernel_time: ms of ONE scan (all its writer launches)
            devcb_line["writer_issue_floor"] = {
                "floor_ms": round(w_floor_ms, 3), "measured_over_floor": round(tw_launch / w_floor_ms, 3) if w_floor_ms > 0 else None,
                "model": f"per wave and 16-row block of two 16-draw groups, both passes: {2 * 4 * nt_w} MFMA4 x 7.44 ns + {2 * nt_w} MFMA16 x 29.76 ns + "
                         "28 Philox rounds x 10.5 ns + ~200 VALU x 2.0 ns; 2 waves per SIMD, no MFMA / VALU co-issue (profiles/r02_coissue_microbench.txt)",
                "note": "NOT the binding constraint: ablations (profiles/r04_experiments.md 1, r06_experiments.md) -- without Philox -0.17 ms, without the "
                        "table look-ups -0.44 ms, without the MFMAs -1.0 ms of 5.1 -- show a wave issuing ~1/3 of its cycles; what is left is LDS "
                        "queueing of the random 16-byte table reads (the generator runs TWICE) and the dependent look-up -> cubic -> MFMA -> store chain "
                        "with two waves per SIMD"}
            e3.close()
        except Exception as ex:  # pragma: no cover
            devcb_line = {"error": repr(ex)}

    # (VERDICT r2 weak #11 asked for a config-3-shaped target Pathfinder can fit.  tests/probes/khat_probe.py tried six: at d = 1000,
    #  J = 6 the pooled Pareto k stays >= 0.92 even for a unit diagonal + rank 4 -- the diagonal of an L-BFGS inverse Hessian is only
    #  roughly right in 1000 dimensions -- so no such variant exists short of the isotropic target, whose traces are one iteration
    #  long; profiles/r03_khat_variants.txt.  The headline's k = 3.8 is the reference algorithm's own answer on this target.)
    # ---- roofline of the dominant kernel (pf_elbo_draws_kernel), hipEvents on the engine's stream ---------
    roofline = None
    if rank == 0 and not args.minimal:
        ms, n = scan_ms, scan_n                                        # hipEvents around the scan's launches in the K timed steps
        m = 2 * J
        bytes_per_draw = 16.0 * d + 8.0 * d * (m + 2) / N_e           # algorithmic bytes per ELBO draw
        alg_bytes = bytes_per_draw * draws_local                       # one launch = every ELBO draw of this rank
        achieved = alg_bytes / (ms / max(n, 1) * 1e-3) / 1e9 if ms > 0 else 0.0
        # ---- what bounds the kernel: fp64 ISSUE.  On gfx950 the f64 MFMA executes on the SIMD's fp64 lanes: nothing co-issues with it
        # (profiles/r02_coissue_microbench.txt, re-tested in round 3), so the floor of the kernel is the SUM of its issue streams.
        # Per 16 rows x 16 draws a group issues 4 x (2 KC/4 + RPAD/4) MFMA 4x4x4 (512 flop each).
        kc = next(o for o in (4, 8, 12, 16, 20, 32) if m <= o)
        rpad = (8 if getattr(tg, "r", 0) <= 8 else 16) if getattr(tg, "r", 0) > 0 else 0
        ncols = 2 * kc + rpad          # w = Vh'z, A3 = Vh'(a s^2 z), A4 = Wd'(s z)
        nblk = -(-d // 16)
        t_launch = ms / max(n, 1) * 1e-3
        # useful flops on the UNPADDED d (the kernel issues nblk = ceil(d / 16) row blocks, d = 1000 -> 1008 rows: the 8 pad rows are not work
        # anybody asked for and are not counted; the issue floor below prices what is really issued)
        mfma_flops = draws_local / 16.0 * (d / 16.0) * 4 * (ncols // 4) * 512.0
        mfma_tf = mfma_flops / t_launch / 1e12 if ms > 0 else 0.0
        # summed-issue floor from the unit times of the co-issue microbenchmark (2 waves per SIMD; they embed the sustained clock):
        # MFMA 4x4x4 7.44 ns, Philox round 10.5 ns, other VALU ~2.0 ns; per wave and block of 16 rows x 32 draws (two groups per wave)
        issue = None
        if kc <= 12 and N_e >= 768:
            n_mfma, n_phx, n_valu = 2 * 4 * (ncols // 4), 2 * 7, 111
            per_block_us = (n_mfma * 7.44 + n_phx * 10.5 + n_valu * 2.0) * 1e-3
            batches = -(-(-(-N_e // 16) + 1) // 16)                     # 16 group slots per workgroup pass (8 waves x 2), one pseudo group
            wave_blocks_per_simd = nfits_local * batches * nblk * 8 / (4.0 * 256)
            floor_ms = per_block_us * wave_blocks_per_simd * 1e-3
            issue = {"floor_ms": round(floor_ms, 3), "measured_over_floor": round(t_launch * 1e3 / floor_ms, 4) if floor_ms > 0 else None,
                     "frac_of_floor": round(floor_ms / (t_launch * 1e3), 4) if ms > 0 else None,
                     "model": f"per wave and 16x32 block: {n_mfma} MFMA4 x 7.44 ns + {n_phx} Philox rounds x 10.5 ns + {n_valu} VALU x 2.0 ns, "
                              "2 waves per SIMD, no MFMA/VALU co-issue (profiles/r02_coissue_microbench.txt, profiles/r03_*)"}
        traffic, traffic_meta = None, None
        if not args.no_pmc and not args.minimal and G == 1 and not use_dist:
            pmc, why = _pmc_traffic([a for a in sys.argv[1:] if a not in ("--no-cpu-baseline",)],
                                    {"scan": "%pf_elbo_qf_kernel%", "writer": "%pf_elbo_xw_kernel%", "reader": "%pfx_%"})
            if pmc is not None and "scan" in pmc:
                traffic, traffic_meta = pmc["scan"]["traffic_bytes_per_launch"], pmc["scan"]
            else:
                traffic_meta = {"source": f"unavailable in this run ({why})"}
            if pmc is not None and isinstance(devcb_line, dict) and "writer" in pmc and "reader" in pmc and "error" not in devcb_line:
                # the device-closure path per launch (one block of fits): measured bytes against the bytes that must move (8 d per draw
                # written by the writer, 8 d per draw read by the closure)
                devcb_line["measured_hbm_bytes_per_launch"] = {
                    "writer": {k: pmc["writer"][k] for k in ("fetch_bytes_per_launch", "write_bytes_per_launch", "launches_profiled")},
                    "reader": {k: pmc["reader"][k] for k in ("fetch_bytes_per_launch", "write_bytes_per_launch", "launches_profiled")},
                    "note": "per launch = one block of fits (PFMI_DEVCB_CHUNK_MB); the writer must write and the reader must read "
                            "8 d bytes per draw of the block; FETCH_SIZE x 2 (gfx950), WRITE_SIZE as reported"}
        roofline = {"bound": "mfma", "achieved": round(mfma_tf, 2), "peak": 78.6, "unit": "TFLOP/s", "frac": round(mfma_tf / 78.6, 4),
                    "traffic": traffic, "traffic_detail": traffic_meta,
                    "kernel": "pf_elbo_qf_kernel (single-pass ELBO scan; one scan = one launch: a workgroup per fit, and behind them the one-batch "
                              "pieces of the fits beyond the last full round of CUs; timed by a hipEvent pair in the engine's stream in each of the timed steps)",
                    "launches": int(n), "avg_launch_ms": round(ms / max(n, 1), 4),
                    "label": "f64 matrix flops of the scan / launch time against the 78.6 TF f64 MFMA peak.  The kernel is fp64-ISSUE bound: "
                             "its floor is the SUM of the MFMA and VALU issue streams (see issue_floor), not the matrix peak alone",
                    "issue_floor": issue,
                    "hbm_equivalent": {"achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
                                       "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_per_draw": bytes_per_draw,
                                       "label": "SECONDARY: SURVEY 8(d)'s yardstick (16 d + factor bytes per draw) / launch time.  The fused scan "
                                                "never forms x, so these bytes are NOT moved (see traffic: a few % of them) and this is not a "
                                                "utilisation; the path where they do move is device_callback_target"}}

    # ---- CPU baseline: the reference algorithm on this box's host cores, two legs ------------
    #   "port":   oracle/pf_oracle.c (scalar C restatement, reflector-by-reflector Q apply), OpenMP over paths
    #   "lapack": tests/cpu_lapack_baseline.py -- the same per-fit pipeline through dgeqrf / dormqr / dtrmm on d x N blocks + NumPy
    #             reductions, threads over paths, BLAS pinned to 1 thread per path: what Julia's lmul!(Q, .) does (src/woodbury.jl:136-143)
    # `cores` = the cores this process can really use (affinity mask, capped by a cgroup CPU quota), NOT os.cpu_count(); the scaling of
    # the port from 1 thread to all cores is reported so that a reader can see where it stops.
    cpu = None
    if rank == 0 and G == 1 and not args.no_cpu_baseline:          # contract: rank 0 at N = 1 only
        try:
            from helpers import oracle_target
            from oracle import pf_oracle as po
            import cpu_lapack_baseline as cl
            aff, logical, quota = cl.usable_cores()
            cores = max(1, min(aff, int(quota)) if quota else aff)
            otg = oracle_target(tg)
            if traces is None:                                         # device-made traces: download a sample for the CPU leg
                from types import SimpleNamespace
                traces = []
                for kk in range(min(Kl, cores)):
                    th_k, _, gr_k = eng.get_trace(kk, logp=False)
                    traces.append(SimpleNamespace(points=th_k, gradients=gr_k))
            minlen = min(len(t.points) for t in traces) - 1

            def run(nf, nthr, reps=1):
                # bounded sample: `nthr * reps` paths (this rank's traces re-used cyclically), first `nf` fits of each, `nthr` at a time
                sel = [traces[i % len(traces)] for i in range(nthr * reps)]
                th = np.concatenate([t.points[:nf + 1] for t in sel])
                gr = np.concatenate([t.gradients[:nf + 1] for t in sel])
                off = np.arange(nthr * reps + 1, dtype=np.int64) * (nf + 1)
                sd = np.arange(len(th), dtype=np.uint64) + np.uint64(1)
                t1 = time.perf_counter()
                r = po.multipath_fit_elbo(off, th, gr, J, otg, N_e, sd, nthreads=nthr)
                return r["total_draws"], time.perf_counter() - t1

            budget = args.cpu_seconds
            n_probe, t_probe = run(min(8, minlen), cores)
            per_fit = t_probe / min(8, minlen)                        # wall seconds per fit and path with every core busy
            nf = int(max(2, min(0.45 * budget / max(per_fit, 1e-3), minlen)))
            reps = int(max(1, min(8, 0.45 * budget / max(per_fit * nf, 1e-3))))      # short traces: several paths per thread, one after the other
            n_all, t_all = run(nf, cores, reps)
            # the SAME per-thread work (first nf fits of a path: early fits have a shorter history and are cheaper, so only equal windows
            # compare) on 1 thread and on a ladder of thread counts: where the port stops scaling is visible in the line
            n_one, t_one = run(nf, 1, reps)
            one = round(n_one / t_one, 1)
            scaling = {"1": one, str(cores): round(n_all / t_all, 1)}
            nf_s = nf if t_all < 3.0 else int(max(8, nf // 4))
            for nthr in (2, 8, 32, 128):
                if nthr < cores:
                    n_s, t_s = run(nf_s, nthr)
                    scaling[str(nthr)] = round(n_s / t_s, 1)
            # LAPACK leg: warm-up (SciPy import, workspace queries), calibrate with two fits per path, then ~0.3 of the budget
            lap = None
            try:
                cl.timed_run(traces, J, tg, N_e, 1, 1)
                n_p, t_p = cl.timed_run(traces, J, tg, N_e, 2, cores)
                nf_l = int(max(2, min(0.3 * budget / max(t_p / 2, 1e-3), minlen)))
                n_l, t_l = cl.timed_run(traces, J, tg, N_e, nf_l, cores)
                n_1, t_1 = cl.timed_run(traces, J, tg, N_e, nf_l, 1)
                lap = {"value": round(n_l / t_l, 1), "unit": "ELBO draws/s", "cores": cores, "kind": "lapack", "value_1_thread": round(n_1 / t_1, 1),
                       "sample": f"{cores} paths x first {nf_l} fits x {N_e} draws (d={d}, J={J}) = {n_l} draws in {t_l:.1f} s; SciPy dgeqrf / dormqr / dtrmm "
                                 "on d x N blocks + NumPy randn / reductions, one thread per path, BLAS pinned to 1 thread per path"}
            except Exception as ex:  # pragma: no cover
                lap = {"value": None, "kind": "lapack", "sample": f"failed: {ex!r}"}
            cpu = {"value": round(n_all / t_all, 1), "unit": "ELBO draws/s", "cores": cores,
                   "kind": "port", "value_1_thread": one,
                   "sample": f"{cores * reps} paths x first {nf} fits x {N_e} draws (d={d}, J={J}) = "
                             f"{n_all} draws in {t_all:.1f} s, OpenMP over paths (oracle/pf_oracle.c, scalar reflector-by-reflector Q apply)",
                   "usable_cores": {"sched_getaffinity": aff, "os_cpu_count": logical, "cgroup_cpu_quota": quota, "used": cores},
                   "port_scaling_draws_per_s_by_threads": scaling,
                   "port_scaling_note": f"1 and {cores} threads: first {nf} fits per path; the other rungs: first {nf_s} (early fits are cheaper: compare like "
                                        f"with like); speed-up all cores / 1 thread = {(n_all / t_all) / one if one else float('nan'):.1f}x",
                   "lapack": lap}
        except Exception as e:  # pragma: no cover
            cpu = {"value": None, "unit": "ELBO draws/s", "cores": os.cpu_count(), "kind": "port",
                   "sample": f"failed: {e!r}"}

    # ---- where the curve bends: the parts of a step / of the end-to-end call that do NOT shrink with the GPU count, on
    #      rank 0, per step, next to the parts that do (fit, ELBO scan, pool)
    non_sharding = None
    if rank == 0 and stages:
        ps = lambda n: stages.get(n, {}).get("ms_per_step", 0.0)        # noqa: E731
        shard_ms = ps("fit") + ps("elbo_draws") + ps("elbo_draws_x") + ps("elbo_reduce")
        fixed = {"history_walk": ps("history"), "psis_replicated": ps("psis"), "index_selection_and_owner_gather": ps("resample"),
                 "collective_handshake_host": ps("comm_handshake_host"), "all_gather_log_ratios": ps("comm_allgather"),
                 "owned_columns_to_rank0": ps("comm_sendrecv")}
        fixed["host_launch_gaps_and_result_download"] = round(ms_per_step - shard_ms - sum(fixed.values()), 4)
        non_sharding = {"per_step_ms": fixed, "sharding_stages_ms": round(shard_ms, 4), "step_ms": round(ms_per_step, 4),
                        "non_sharding_total_ms": round(ms_per_step - shard_ms, 4),
                        "outside_the_hot_path_ms": {"optimize_device_lbfgs": stages.get("optimize", {}).get("ms"),
                                                    "trace_pack": stages.get("trace_pack", {}).get("ms")},
                        "note": "rank 0; stage figures are hipEvent pairs left in the engine's stream during the timed steps (the handshake of the "
                                "process-per-GPU mode is host wall-clock: it synchronises); with N ranks the sharding stages shrink ~1/N, these do not"}

    if rank == 0:
        line = {
            "metric": "ELBO draws/sec (multipathfinder hot path: fit + ELBO + pool + PSIS + resample)",
            "value": round(value, 1), "unit": "ELBO draws/s", "n_gpus": G, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"multipathfinder npaths={K} d={d} " + {"lowrank": "correlated Gaussian (low-rank r=8 + diag)",
                                   "diag": "diagonal Gaussian", "iso": "isotropic Gaussian", "funnel": "funnel"}[args.target] + ", "
                                   f"history_length={J}, ndraws_elbo={N_e}, ndraws={ndraws}",
                       "npaths": K, "paths_per_gpu": Kl, "fits_total": int(total_draws // N_e),
                       "elbo_draws_per_step": int(total_draws), "parallelism": f"paths sharded x{G}",
                       "ranks_in_collective": ranks_seen,
                       "collective_backend": comm_note if use_dist else None},
            "multipathfinder_hot_path_ms": round(ms_per_step, 3),
            "multipathfinder_wall_ms_incl_device_lbfgs": None if wall_e2e is None else round(wall_e2e, 3),
            "multipathfinder_wall_ms_incl_device_lbfgs_packed_route": None if wall_e2e_packed is None else round(wall_e2e_packed, 3),
            "streamed_equals_packed": streamed_equal, "streaming_note": stream_note,
            "multipathfinder_api_wall_ms": None if api_wall is None else round(api_wall, 3),
            "traces": "host numpy L-BFGS driver" if args.host_traces else "device L-BFGS (pfmi_optimize_batch)",
            "pareto_k": state.get("pareto_k"),
            "sharded_equals_single": verdict, "sharded_equals_single_note": vnote, "rccl_version": rccl_version,
            "stages_ms": stages,
            "non_sharding_ms": non_sharding,
            "callback_target": callback_line,
            "device_callback_target": devcb_line,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        # ---- self-checks of the line are FATAL (a NaN parity figure went out unnoticed): the line is still printed, the
        #      process then exits non-zero
        if isinstance(devcb_line, dict) and "error" not in devcb_line:
            v = devcb_line["max_rel_elbo_diff_vs_builtin_target"]
            if not (np.isfinite(v) and v <= 1e-10):
                self_check_failures.append(f"device-closure ELBO table differs from the built-in target's: max rel diff {v}")
        elif isinstance(devcb_line, dict):
            self_check_failures.append(f"device-closure sample failed: {devcb_line['error']}")
        if streamed_equal is False:
            self_check_failures.append("streamed pipeline result differs from the packed route's")
        if verdict is False:
            self_check_failures.append(f"sharded run differs from the single-GPU recomputation: {vnote}")
        if state.get("pareto_k") is not None and not np.isfinite(state["pareto_k"]):
            self_check_failures.append("pooled Pareto k is not finite")
        line["self_checks"] = {"failed": self_check_failures, "ok": not self_check_failures}
    if use_dist:
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()
    elif comm is not None:
        comm.close()
    eng.close()
    if rank == 0:
        # RCCL writes a version banner through C stdio that would otherwise be flushed at exit, AFTER this line:
        # drain it first so that the JSON result is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)
        if self_check_failures:
            sys.exit("bench.py: SELF-CHECK FAILED: " + "; ".join(self_check_failures))


if __name__ == "__main__":
    main()
