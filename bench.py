#!/usr/bin/env python
"""bench.py -- grid-points/s of one seabreeze_diag call (BASELINE.json metric).

A "step" is one full seabreeze_diag call (generic/sea_breeze_diag.f90:55 semantics:
sigmoid statistics, t0, window contrast, per-column level search, thresholds, state
update) over the whole grid, inputs already resident in HBM.  Default workload:
configs[2] of BASELINE.json, the configuration the metric is quoted on -- N1280
(2560x1920) fp64 with nz=56 levels -- which fits one MI355X (about 16 GB).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N>1 splits the grid into N latitude bands (strong scaling): per step every rank
all-gathers its sigma moments and swaps theta halo rows with its band neighbours over
RCCL (the reference's swap_bounds stub, generic/halo_exchange_mod.f90:12-17).

Prints ONE JSON line on rank 0 with the `roofline`, `cpu_baseline` and `parity` objects.
`value` is nx*ny / (wall time of the K timed steps / K), as the driver's contract asks;
`value_median` is the same from the median of per-step HIP-event times,
taken in a second K-step pass so that no event sits between the timed launches.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
# torch BEFORE the library is loaded: PyTorch bundles a HIP runtime and loads it by path, libseabreeze_hip.so asks for
# the soname -- in this order the library binds to the runtime torch brought and the process has ONE runtime, so torch's
# stream handle is a valid hipStream_t for the C ABI and torch.cuda.synchronize() sees the library's kernels.  In the
# other order the process holds two runtimes, of which only the first to initialise finds the GPU (measured on this pool);
# the run checks `hip.hip_runtimes()`, names the runtime in config.hip_runtime and refuses to time anything otherwise.
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
if os.environ.get("SEABREEZE_BENCH_LIBRARY_FIRST"):
    # test knob (tests/test_bench_contract_gpu.py): the OTHER load order must end in a clear refusal, not in a number
    from seabreeze_param_amd import hip as _hip_first
    _hip_first.load_library()
import torch  # noqa: E402,F401

from seabreeze_param_amd import hip, synth  # noqa: E402
from seabreeze_param_amd.bands import BandRunner, row_cost, split_rows  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
REL_FLOOR = 1e-2        # parity: |a-b| / max(|b|, REL_FLOOR)


def algorithmic_bytes(n, n_band, nz, s=8, wind_final=True):
    """B = s*[5N + (nz+7)*N_c], split per kernel."""
    return {
        "k_scan": s * (3 * n - n_band),                    # read sigma, mask; write sb_con outside the band
        # single domain: k_wind applies the thresholds and the state update; a band step leaves that to k_thc3
        "k_thc": s * (2 * n + (0 if wind_final else 6 * n_band)),      # theta, z in (sigma's second read is not counted)
        "k_wind": s * (nz + (8 if wind_final else 2)) * n_band,        # p column, u, v (+ ws, wd in; ws, wd, thc, sb_con out)
        "k_prep": 0,                                       # lists and statistics: workspace traffic only
        "total": s * (5 * n + (nz + 7) * n_band),
    }


def pmc_traffic(kernel, nx, ny, nz):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/pmc_traffic.json, made by tools/collect_profiles.sh + tools/make_traffic_json.py:
    (2*FETCH_SIZE + WRITE_SIZE)*1024, the gfx950 correction of the microarchitecture guide).
    None when no pass exists for this workload."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    c = d.get("config", {})
    if (c.get("nx"), c.get("ny"), c.get("nz")) != (nx, ny, nz):
        return None, None
    ks = d.get("kernels", {})
    ent = ks.get(kernel) or (ks.get("k_strip") if kernel == "k_thc" else None) or ks.get(kernel + "3") or {}
    return ent.get("hbm_bytes"), d.get("source")


def host_cores() -> int:
    """Cores this process may really use: affinity, capped by the cgroup CPU quota and by
    the 16-core share a one-GPU box grants."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def relerr(a, b):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.maximum(np.abs(b.astype(np.float64)), REL_FLOOR)
    d[np.isnan(a) & np.isnan(b)] = 0.0
    return float(d.max()) if d.size else 0.0


def cpu_baseline_and_parity(st, cdist, p, u, v, thetas, nz, gpu_states, timestep, budget_s=20.0, prec=8):
    """The CPU oracle (our Fortran restatement of the reference, `kind: port`) on this box's host cores,
    in two roles: (1) checker -- the step sequence tn = 1, 2, 15 (first step, ordinary step, a step whose
    target_time branch fires) on the same inputs as the GPU ran it, compared field by field; (2) timed
    baseline -- serial and all-core OpenMP, bounded by budget_s each."""
    from oracle.pyoracle import Oracle        # checker / baseline only
    ny, nx = st.ny, st.nx
    ncores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(ncores)
    orc_omp = Oracle(prec, omp=True)
    # ---- parity -----------------------------------------------------------------------------------
    names = ("windspeed", "winddir", "thc", "sb_con")
    if prec == 8:
        state = [np.zeros((ny, nx), p.dtype) for _ in range(4)]
        worst = {n: 0.0 for n in names}
        pattern_equal = True
        for (tn, inputs), gstate in zip(gpu_states["steps"], gpu_states["states"]):
            pp, uu, vv, th = inputs
            orc_omp.seabreeze_diag(timestep, tn, pp, uu, vv, th, cdist, st.z, st.sigma, *state, halo=0, bnd=1, omp=True)
            for nme, a, b in zip(names, gstate, state):
                worst[nme] = max(worst[nme], relerr(a, b))
            pattern_equal = pattern_equal and bool(np.array_equal(gstate[3] != 0, state[3] != 0))
        parity = {"max_rel_err": worst, "rel_floor": REL_FLOOR, "tolerance": 1e-6,
                  "steps": [tn for tn, _ in gpu_states["steps"]], "trigger_pattern_equal": pattern_equal,
                  "checker": "oracle/sb_oracle.f90 (OpenMP build), same inputs and step sequence",
                  "ok": bool(all(v < 1e-6 for v in worst.values()) and pattern_equal)}
    else:
        # Single precision is checked against the DOUBLE-precision arithmetic of the reference on the same
        # (fp32-representable) inputs: the reference's own fp32 build sums up to (2*16+1)^2 temperatures near
        # 290 K sequentially in fp32 and carries ~6e-4 K of rounding noise in thc (tests/fp32_tolerance_study.py),
        # more than the HIP path's error, so it cannot serve as the yardstick.  Tolerances as in
        # tests/test_parity_gpu.py::test_baseline_config3_fp32_vs_oracle.
        from oracle import fp32_criterion as crit          # the rule tests/test_parity_gpu.py holds configs[3] to as well
        orc8 = Oracle(8, omp=True)
        f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        state = [np.zeros((ny, nx), np.float64) for _ in range(4)]
        cd8, z8, sg8 = f8(cdist), f8(st.z), f8(st.sigma)
        band = np.abs(cd8) <= 180.0
        g_prev = [np.zeros((ny, nx), np.float32) for _ in range(4)]
        per_step = []
        for (tn, inputs), gstate in zip(gpu_states["steps"], gpu_states["states"]):
            pp, uu, vv, th = inputs
            o_prev = [a.copy() for a in state]
            orc8.seabreeze_diag(timestep, tn, f8(pp), f8(uu), f8(vv), f8(th), cd8, z8, sg8, *state, halo=0, bnd=1, omp=True)
            per_step.append(crit.check_step(tn, g_prev, gstate, o_prev, state, band, timestep=timestep))
            g_prev = gstate
        parity = crit.merge(per_step)
        parity["steps"] = [tn for tn, _ in gpu_states["steps"]]
        parity["checker"] = ("oracle/sb_oracle.f90 in DOUBLE precision on the same fp32-representable inputs (the reference's "
                             "own fp32 arithmetic carries ~6e-4 K of window-sum noise: tests/fp32_tolerance_study.py); sb_con by "
                             "per-cell error propagation from the measured thc and wind errors, no cell masked "
                             "(oracle/fp32_criterion.py)")
    # ---- timing -----------------------------------------------------------------------------------
    out = {}
    for name, omp in (("serial", False), ("omp", True)):
        orc = orc_omp if omp else Oracle(prec)
        state = [np.zeros((ny, nx), p.dtype) for _ in range(4)]
        times = []
        t_all = time.perf_counter()
        tn = 1
        while True:
            t0 = time.perf_counter()
            orc.seabreeze_diag(timestep, tn, p, u, v, thetas[0], cdist, st.z, st.sigma, *state, halo=0, bnd=1, omp=omp)
            times.append(time.perf_counter() - t0)
            tn += 1
            if time.perf_counter() - t_all > budget_s or len(times) >= 12:
                break
        # first call pays page faults; use the median of the rest when there is a rest
        med = float(np.median(times[1:])) if len(times) > 1 else times[0]
        out[name] = dict(s_per_call=med, calls=len(times))
    return out, ncores, parity


def band_parity(st, cdist, rows, h, host_sets, gpu_states, timestep, prec):
    """N > 1: every rank checks ITS band against the CPU oracle's raw-index (ghost-cell) flavour -- the band's own rows of
    the 3-D fields, the 2-D fields framed with the ghost rows and columns the exchange must have delivered (built here
    from the global 2-D fields with numpy: latitude clamp, longitude wrap), and the global sigmoid scalars.  A wrong
    ghost row, a wrong all-gathered statistic or a wrong band cut shows up here.  Returns this rank's worst errors."""
    from oracle.pyoracle import Oracle        # checker only
    ny, nx = st.ny, st.nx
    r0, r1 = rows
    orc = Oracle(8)
    f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    rr = np.clip(np.arange(r0 - h, r1 + h), 0, ny - 1)
    cc = np.arange(-h, nx + h) % nx
    frame = lambda full: f8(full[np.ix_(rr, cc)])
    ext = orc.sigmoid_scalars(f8(st.sigma))
    z_f, sg_f, mk_f = frame(st.z), frame(st.sigma), frame(cdist)
    state = [np.zeros((r1 - r0, nx), np.float64) for _ in range(4)]
    band = np.abs(f8(cdist[r0:r1])) <= 180.0
    worst = np.zeros(4)
    flips = 0
    for (tn, k), gstate in zip(gpu_states["steps"], gpu_states["states"]):
        pp, uu, vv, th = host_sets[k]
        orc.seabreeze_diag(timestep, tn, f8(pp), f8(uu), f8(vv), frame(th), mk_f, z_f, sg_f, *state, halo=h, bnd=2, ext_stats=ext)
        g = [a.astype(np.float64) for a in gstate]
        if prec == 8:
            for i in range(4):
                worst[i] = max(worst[i], relerr(g[i], state[i]))
        else:                                  # fp32 against the fp64 arithmetic: see cpu_baseline_and_parity
            worst[0] = max(worst[0], relerr(g[0][band], state[0][band]) if band.any() else 0.0)
            dd = np.abs(g[1][band] - state[1][band])
            worst[1] = max(worst[1], float(np.minimum(dd, 360.0 - dd).max()) if band.any() else 0.0)
            worst[2] = max(worst[2], float(np.abs(g[2][band] - state[2][band]).max()) if band.any() else 0.0)
        flips += int(((g[3] != 0) != (state[3] != 0)).sum())
    return worst, flips


def time_setup_kernels(ctx, torch, st, coast, dt, kwin, reps=5):
    """get_edges / get_dist with device-resident arguments (SURVEY.md §8(d) secondary rows): HIP-event
    median over `reps` launches each, algorithmic bytes 3*N*s each."""
    dev = torch.device("cuda", torch.cuda.current_device())
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    lf, ic, co = up(st.landfrac), up(st.icefrac), up(coast)
    out = torch.empty_like(lf)
    stream = hip.torch_stream_handle(torch)
    ny, nx = st.ny, st.nx
    esz = np.dtype(dt).itemsize
    res = {}

    def timed(fn):
        ts = []
        fn()                                            # (untimed: tables and workspace of the first call)
        for _ in range(reps):
            if stream is not None:                      # one HIP runtime: events on the stream the kernels run on
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            else:                                       # two runtimes: the library's own stream, host clock round a synchronised call
                ctx.synchronize()
                t0 = time.perf_counter()
                fn()
                ctx.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts))

    ms = timed(lambda: ctx.get_edges_dev(dt, nx, ny, lf.data_ptr(), ic.data_ptr(), out.data_ptr(), stream=stream))
    alg = 3 * nx * ny * esz
    res["get_edges"] = {"ms": round(ms, 5), "algorithmic_bytes": alg, "achieved": alg / ms / 1e6,
                        "frac": alg / ms / 1e6 / HBM_PEAK_GBS}
    ms = timed(lambda: ctx.get_dist_dev(dt, nx, ny, co.data_ptr(), lf.data_ptr(), st.lon, st.lat, out.data_ptr(),
                                        kwin=kwin, stream=stream))
    res["get_dist"] = {"ms": round(ms, 5), "algorithmic_bytes": alg, "achieved": alg / ms / 1e6,
                       "frac": alg / ms / 1e6 / HBM_PEAK_GBS, "window_half_width": kwin}
    for k in res.values():
        k["unit"] = "GB/s"
    return res


class Watchdog:
    """N > 1 only: a rank that passes no stage mark for `limit` seconds (a collective some other rank never joined, a
    transport that never connects) says where it stands and ends its process with exit code 5 -- the launcher then
    tears the job down instead of waiting for ever.  The multi-rank RCCL path has not run on hardware yet."""

    def __init__(self, limit, rank):
        import threading
        self.limit, self.rank, self.stage, self.t = limit, rank, "start", time.monotonic()
        if limit > 0:
            threading.Thread(target=self._watch, daemon=True).start()

    def mark(self, stage):
        self.stage, self.t = stage, time.monotonic()

    def _watch(self):
        while True:
            time.sleep(1.0)
            if time.monotonic() - self.t > self.limit:
                print(f"[bench rank {self.rank}] no progress for {self.limit:.0f} s in stage '{self.stage}': giving up",
                      file=sys.stderr, flush=True)
                os._exit(5)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nx", type=int, default=2560)
    ap.add_argument("--ny", type=int, default=1920)
    ap.add_argument("--nz", type=int, default=56)
    ap.add_argument("--dtype", choices=("f64", "f32"), default="f64",
                    help="working precision (BASELINE configs[3], the N2560 grid, is quoted in fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle (baseline AND parity check)")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU baseline per variant")
    ap.add_argument("--comm", choices=("native", "torch", "gloo"), default="native",
                    help="N>1: ghost rows + moments over the library's own RCCL communicator (native; the "
                         "control plane is gloo), over torch.distributed's nccl backend (torch), or over gloo "
                         "(rehearsal of the N>1 code path on a box with fewer GPUs than ranks).  A transport that "
                         "cannot be initialised ends the run with a non-zero exit code: there is no silent fallback.")
    ap.add_argument("--profile-passes", type=int, default=1,
                    help="extra K-step passes with HIP events around every kernel (0 = events inside the timed pass)")
    ap.add_argument("--no-fold", action="store_true", help="measurement: k_prep as a kernel of its own (sb_set_fold(ctx, 0))")
    ap.add_argument("--no-replan", action="store_true",
                    help="skip the extra passes with sb_set_plan_cache(ctx, 0) (profiling runs: rocprofv3's per-kernel averages "
                         "then refer to the stored-plan state alone)")
    ap.add_argument("--watchdog", type=float, default=600.0,
                    help="N>1: seconds a rank may spend in one stage of the run before it gives up (0: never)")
    ap.add_argument("--static-sigma", action="store_true",
                    help="opt-in variant, never the headline: sigma's statistics formed once (sb_set_static_sigma)")
    args = ap.parse_args()

    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if len(hip.hip_runtimes()) > 1:
        raise SystemExit(f"bench.py: two HIP runtimes in this process ({hip.hip_runtimes()}): import torch before "
                         "libseabreeze_hip.so is loaded -- refusing to time kernels torch's synchronisation cannot see")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.comm == "gloo":
        local_rank %= max(1, torch.cuda.device_count())      # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    # every launch of this run -- the library's kernels through the _dev entry points and the timing events --
    # goes to one explicit stream (torch's default stream has handle 0, which the C ABI reads as "the context's own")
    torch.cuda.set_stream(torch.cuda.Stream())
    comm = args.comm
    dog = Watchdog(args.watchdog if world > 1 else 0.0, rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if comm in ("native", "gloo"):
            dist.init_process_group("gloo")
            if comm == "gloo":
                comm = "torch"
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    nx, ny, nz = args.nx, args.ny, args.nz
    dt = np.float64 if args.dtype == "f64" else np.float32
    # which BASELINE.json configuration this run is (None: a grid / precision BASELINE.json does not name)
    cfg_index = {(1024, 768, "f64"): 1, (2560, 1920, "f64"): 2, (5120, 3840, "f32"): 3}.get((nx, ny, args.dtype))
    esz = 8 if args.dtype == "f64" else 4
    K, W = args.steps, args.warmup

    # ---- synthetic inputs (host, deterministic), shared by GPU and CPU baseline --------
    t_gen = time.perf_counter()
    st = synth.static_fields(nx, ny, dt)
    ctx = hip.Context(local_rank)
    runtimes = hip.hip_runtimes()
    shared_runtime = len(runtimes) == 1
    if not shared_runtime:
        raise SystemExit(f"bench.py: torch and libseabreeze_hip.so do not share one HIP runtime ({runtimes}): import torch before "
                         "the library is loaded -- refusing to time kernels torch's synchronisation cannot see")
    if args.no_fold:
        ctx.set_fold(False)
    if world > 1 and comm == "native":
        # rank 0 makes the RCCL id, gloo hands it round, every rank joins; all ranks agree on the outcome
        uid = [hip.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ok = torch.ones(1, dtype=torch.int32)
        try:
            ctx.comm_init(uid[0], rank, world)
        except hip.SeabreezeHipError as e:
            print(f"[bench rank {rank}] native RCCL init failed: {e}", file=sys.stderr, flush=True)
            ok[0] = 0
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok[0]) == 0:
            # no silent change of transport: an N>1 number must be an RCCL number unless another was asked for
            if rank == 0:
                print("[bench] the library's RCCL communicator could not be initialised on every rank; "
                      "rerun with --comm torch or --comm gloo to use another transport", file=sys.stderr, flush=True)
            dist.barrier()
            dist.destroy_process_group()
            raise SystemExit(3)
    dog.mark("setup chain and synthetic inputs")
    coast = ctx.get_edges(st.landfrac, st.icefrac)                    # HIP (product) setup chain
    cdist = ctx.get_dist(coast, st.landfrac, st.lon, st.lat)
    kwin = hip.dist_window(st.lon, st.lat)
    band = np.abs(cdist) <= 180.0
    n_band_total = int(band.sum())
    # bands cut by cost (5 units per cell + nz+7 per coastal-band cell), not by row count: the band is clustered
    # in latitude; every rank derives the same cuts from the same distance field
    r0, r1 = split_rows(ny, world, cost=row_cost(band, nz) if world > 1 else None, min_rows=kwin + 1)[rank]
    rows = (r0, r1)                      # every rank generates only its own band of the 3-D fields
    p_full = synth.pressure_3d(st, nz, dt, rows=rows)
    theta_a = synth.theta_step(st, 1, dt)
    theta_b = synth.theta_step(st, 2, dt)
    theta_c = synth.theta_step(st, 3, dt)
    u_full, v_full = synth.wind_step(st, nz, 1, dt, rows=rows)
    gen_s = time.perf_counter() - t_gen

    dog.mark("uploads")
    runner = BandRunner(ctx, torch, dist if world > 1 else None, rank, world, nx, ny, nz, halo=kwin + 1, dtype=dt,
                        comm=comm if world > 1 else "torch", rows=rows, static_sigma=args.static_sigma)
    runner.upload_static(st.z, st.sigma, cdist)
    # three input sets at different addresses, taken in turn: B swaps u and v (distinct synthetic
    # winds), C negates them (speed as A, the opposite direction), each with its own step's theta -- no step re-reads
    # lines that the step before it, or the one before that, fetched (a set's 3-D fields are 6.6 GB in fp64)
    neg_u, neg_v = -u_full, -v_full
    set_a = runner.upload_step_inputs(p_full, u_full, v_full, theta_a, local3d=True)
    set_b = runner.upload_step_inputs(p_full, v_full, u_full, theta_b, local3d=True)
    set_c = runner.upload_step_inputs(p_full, neg_u, neg_v, theta_c, local3d=True)
    sets = (set_a, set_b, set_c)
    NSET = len(sets)
    host_sets = ((p_full, u_full, v_full, theta_a), (p_full, v_full, u_full, theta_b), (p_full, neg_u, neg_v, theta_c))
    timestep = 1440.0    # s; target_time branch fires every 15th step

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- parity sequence (single GPU): tn = 1, 2, 15 from a zero state, states kept for the checker ----
    gpu_states = None
    dog.mark("parity steps (first band steps: ghost rows and moments travel)")
    if not args.no_cpu_baseline:
        gpu_states = {"steps": [], "states": []}
        for tn in (1, 2, 15):
            runner.step(timestep, tn, sets[tn % NSET])
            runner.synchronize()
            gpu_states["steps"].append((tn, host_sets[tn % NSET]) if world == 1 else (tn, tn % NSET))
            gpu_states["states"].append([t.cpu().numpy().copy() for t in (runner.ws, runner.wd, runner.thc, runner.sb_con)])
    band_check = None
    dog.mark("band parity against the oracle")
    if world > 1 and gpu_states is not None:
        # N > 1: every rank checks its own band (a serial oracle call on 1/N of the grid), the worst error travels to rank 0
        worst, flips = band_parity(st, cdist, rows, runner.h, host_sets, gpu_states, timestep, esz)
        red = torch.tensor(list(worst) + [float(flips)], dtype=torch.float64,
                           device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(red[:4], op=dist.ReduceOp.MAX)
        dist.all_reduce(red[4:], op=dist.ReduceOp.SUM)
        band_check = [float(x) for x in red.cpu()]

    def timed_pass(tn, warm):
        """W untimed steps, then exactly K steps between barrier + synchronisation (of the library's context and of
        torch's device, whichever runtime each lives in) on both sides; the maximum over ranks."""
        for _ in range(warm):
            runner.step(timestep, tn, sets[tn % NSET]); tn += 1
        runner.synchronize(); barrier()
        t0 = time.perf_counter()
        for _ in range(K):
            runner.step(timestep, tn, sets[tn % NSET]); tn += 1
        runner.synchronize(); barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([el], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el, tn

    events_inside = args.profile_passes == 0
    tn = 1
    dog.mark("warm-up and timed steps")
    if events_inside:                                   # (the PMC passes: events of the K timed steps only)
        for _ in range(W):
            runner.step(timestep, tn, sets[tn % NSET]); tn += 1
        runner.synchronize()
        ctx.profile_begin(K)
        elapsed, tn = timed_pass(tn, 0)
    else:
        elapsed, tn = timed_pass(tn, W)

    # ---- per-step HIP-event times (median): a K-step pass of its own; only where torch's events sit
    # on the stream the kernels run on (one HIP runtime) ---------------------------------------------------------
    median_ms = None
    dog.mark("event passes")
    if shared_runtime:
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
        evs[0].record()
        for i in range(K):
            runner.step(timestep, tn, sets[tn % NSET]); tn += 1
            evs[i + 1].record()
        runner.synchronize()
        step_ms = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(K)])
        median_ms = float(np.median(step_ms))

    # ---- per-kernel HIP-event timing (the library's own events on the stream its kernels run on) -------
    if not events_inside:
        ctx.profile_begin(K * args.profile_passes)
        for _ in range(K * args.profile_passes):
            runner.step(timestep, tn, sets[tn % NSET]); tn += 1
    kern_ms, ncalls = ctx.profile_end()
    counters = ctx.last_counters()

    # ---- the same K steps with the contrast kernel's plan remade in every call (sb_set_plan_cache(0)): what a first
    # call, or a call after the ice edge moved, costs.  The headline above is the stored-plan state. -------------
    replan = None
    if world == 1 and not args.no_replan:
        ctx.set_plan_cache(False)
        el_r, tn = timed_pass(tn, W)
        ctx.profile_begin(K)
        for _ in range(K):
            runner.step(timestep, tn, sets[tn % NSET]); tn += 1
        kr, _ = ctx.profile_end()
        ctx.set_plan_cache(True)
        replan = {"ms_per_step": el_r / K * 1e3, "k_thc": round(kr["k_thc"], 5), "steps": K,
                  "what": "sb_set_plan_cache(ctx, 0): shares, schedule, cell lists, radius search in every call"}

    dog.mark("report")
    ms_per_step = elapsed / K * 1e3
    value = nx * ny / (elapsed / K)

    n_local = nx * (r1 - r0)
    n_band_local = int(band[r0:r1].sum())
    ab = algorithmic_bytes(n_local, n_band_local, nz, s=esz, wind_final=(world == 1))
    knames = ("k_scan", "k_wind", "k_thc")
    dom = max(knames, key=lambda k: kern_ms[k])
    fracs = {k: (ab[k] / (kern_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS if kern_ms[k] > 0 else None) for k in knames}
    lim = min((k for k in knames if fracs[k] is not None), key=lambda k: fracs[k], default=None)
    dom_gbs = ab[dom] / (kern_ms[dom] * 1e-3) / 1e9 if kern_ms[dom] > 0 else 0.0
    call_gbs = ab["total"] / (elapsed / K) / 1e9
    traffic, traffic_src = pmc_traffic(dom, nx, ny, nz) if world == 1 else (None, None)

    result = {
        "metric": "grid-points/sec for sea_breeze_diag on N2560x1920 global grid; achieved HBM GB/s",
        "value": value,
        "unit": "grid-points/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": ms_per_step,
        "median_ms_per_step": median_ms,
        "value_median": nx * ny / (median_ms * 1e-3) if median_ms else None,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": f"seabreeze_diag generic flavour, N{nx // 2} ({nx}x{ny}) global grid, nz={nz}, {'fp64' if esz == 8 else 'fp32'} "
                        + (f"(BASELINE.json configs[{cfg_index}])" if cfg_index is not None else "(not a BASELINE.json configuration)"),
            "baseline_config_index": cfg_index,
            "hip_runtime": runtimes,
            "stream": "torch's current stream (torch and the library share one HIP runtime)",
            "plan_cache": "stored",
            "nx": nx, "ny": ny, "nz": nz,
            "band_fraction": n_band_total / (nx * ny),
            "search_halo": kwin + 1,
            "parallelism": f"latband{world}",
            "variant": ("static-sigma (opt-in; not the reference's per-call statistics)" if args.static_sigma else "default")
                       + (", k_prep as its own kernel" if args.no_fold else ""),
            "comm": (comm + ("-rccl" if comm == "native" else "-" + dist.get_backend())) if world > 1 else "none",
            "multi_rank_rccl": "unmeasured on hardware so far (one GPU per box in the build pool)" if world == 1 else "this run",
            "input_gen_s": round(gen_s, 1),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": dom,
            "achieved": dom_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": dom_gbs / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": ("committed rocprofv3 PMC passes of this command: " + ", ".join(traffic_src)) if traffic_src else None,
            "algorithmic_bytes_per_launch": ab[dom],
            "kernel_ms": {k: round(vv, 5) for k, vv in kern_ms.items()},
            "kernel_algorithmic_bytes": {k: ab[k] for k in knames},
            "kernel_frac": fracs,
            "limiting": {"kernel": lim, "frac": fracs[lim]} if lim else None,      # the kernel furthest below its roofline
            "event_calls": ncalls,
            "whole_call": {"algorithmic_bytes": ab["total"], "achieved": call_gbs, "frac": call_gbs / HBM_PEAK_GBS},
            "plan_cache": "stored",
            "launches_per_call": ctx.last_step_report()["kernel_launches"],
            "replan": replan,
            "rank0_counters": counters,
        },
    }

    if rank == 0 and world == 1:
        result["secondary"] = time_setup_kernels(ctx, torch, st, coast, dt, kwin)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb, ncores, parity = cpu_baseline_and_parity(st, cdist, p_full, u_full, v_full, (theta_a, theta_b), nz,
                                                     gpu_states, timestep, args.cpu_budget, prec=esz)
        result["parity"] = parity
        result["cpu_baseline"] = {
            "value": nx * ny / cb["omp"]["s_per_call"],
            "unit": "grid-points/s",
            "cores": ncores,
            "kind": "port",
            "sample": f"{cb['omp']['calls']} full calls of the same {nx}x{ny}x{nz} workload "
                      f"(median of calls 2..n), oracle/sb_oracle.f90 amdflang -O2 -fopenmp: the restatement that "
                      f"tests/test_oracle_pin.py shows bit-identical to the reference's own files compiled as oracle/_ref",
            "serial": {"value": nx * ny / cb["serial"]["s_per_call"], "cores": 1, "calls": cb["serial"]["calls"]},
        }
    if rank == 0 and band_check is not None:
        names = ("windspeed", "winddir", "thc", "sb_con")
        if esz == 8:
            result["parity"] = {"max_rel_err": dict(zip(names, band_check[:4])), "rel_floor": REL_FLOOR, "tolerance": 1e-6,
                                "steps": [1, 2, 15], "trigger_pattern_equal": band_check[4] == 0,
                                "checker": "oracle/sb_oracle.f90, ghost-cell flavour, every rank on its own band "
                                           "(ghost frame from the global 2-D fields, global sigmoid scalars); worst over ranks",
                                "ok": bool(all(v < 1e-6 for v in band_check[:4]) and band_check[4] == 0)}
        else:
            tol = (5e-6, 1e-3, 2e-4)
            result["parity"] = {"max_err": {"windspeed_rel": band_check[0], "winddir_abs_deg": band_check[1], "thc_abs_K": band_check[2]},
                                "tolerance": dict(zip(("windspeed_rel", "winddir_abs_deg", "thc_abs_K"), tol)),
                                "steps": [1, 2, 15], "trigger_flips": int(band_check[4]),
                                "checker": "oracle/sb_oracle.f90 in double precision, ghost-cell flavour, every rank on its own band; worst over ranks",
                                "ok": bool(all(band_check[i] <= tol[i] for i in range(3)) and band_check[4] <= max(2, n_band_total // 2000))}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0 and "parity" in result and not result["parity"]["ok"]:
        print("[bench] PARITY FAILURE: the GPU outputs differ from the CPU oracle beyond 1e-6", file=sys.stderr)
        raise SystemExit(4)


if __name__ == "__main__":
    main()
