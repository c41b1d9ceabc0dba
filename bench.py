#!/usr/bin/env python3
"""Headline benchmark: env-steps/s and achieved HBM GB/s of the persistent vmap_sim_ahead kernel.

Workload (BASELINE.json configs[2]): PMSM dq-frame, explicit Euler, fp32, batch 2^22 per GPU,
tau = 1e-4. One bench "step" = one vmap_sim_ahead launch of `--chunk` (default 100) solver steps over the whole
batch with full reference outputs (observations + all physical-state trajectories); chunks are chained through
last_state, so the default 100 timed steps are the 10 000-step run. Inputs (initial state, one action chunk that
every step reuses) are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU, batch sharded (weak scaling: 2^22 envs per rank), no collective on the stepping path. Either
launched by torch.distributed.run (WORLD_SIZE / RANK / LOCAL_RANK in the environment) or invoked directly — then this
process, before touching any GPU, starts the N ranks itself as child processes (torch.distributed.run) and forwards
their output and exit code; the observations are reassembled with ONE RCCL all-gather of the final
observation row at the end of the timed region (`--gather chunk`: after every chunk on a side stream, overlapped;
`--gather none`: never).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "exciting-environments_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s measured copy ceiling

WORKLOADS = {
    # name: (registry, solver, dtype, tau, log2 batch, default chunk)
    "pmsm_euler_f32": ("PMSM", "euler", torch.float32, 1e-4, 22, 100),
    "pendulum_euler_f32": ("PENDULUM", "euler", torch.float32, 2e-2, 20, 1000),
    "msd_tsit5_f64": ("MASS_SPRING_DAMPER", "tsit5", torch.float64, 1e-4, 20, 500),
    # the other environments / solvers at the headline batch (not BASELINE configs; for the per-env table in DESIGN.md)
    "msd_euler_f32": ("MASS_SPRING_DAMPER", "euler", torch.float32, 1e-4, 22, 100),
    "cartpole_euler_f32": ("CART_POLE", "euler", torch.float32, 2e-2, 22, 100),
    "acrobot_euler_f32": ("ACROBOT", "euler", torch.float32, 1e-3, 22, 100),
    "tank_euler_f32": ("FLUID_TANK", "euler", torch.float32, 1e-3, 22, 100),
    "pmsm_tsit5_f32": ("PMSM", "tsit5", torch.float32, 1e-4, 22, 100),
    "pmsm_rk4_f32": ("PMSM", "rk4", torch.float32, 1e-4, 22, 100),
    "acrobot_tsit5_f32": ("ACROBOT", "tsit5", torch.float32, 1e-3, 22, 100),
    "pmsm_euler_f64": ("PMSM", "euler", torch.float64, 1e-4, 21, 100),
    "pmsm_sat_euler_f32": ("PMSM_SAT", "euler", torch.float32, 1e-4, 22, 100),  # saturated model, synthetic LUT
    "pmsm_sat_tsit5_f32": ("PMSM_SAT", "tsit5", torch.float32, 1e-4, 22, 100),
    "pmsm_sat_euler_f64": ("PMSM_SAT", "euler", torch.float64, 1e-4, 21, 100),
    # more solver / dtype combinations of the small models (workgroup-shape sweep, tools/r4_block_sweep.sh)
    "pendulum_euler_f64": ("PENDULUM", "euler", torch.float64, 2e-2, 21, 100),
    "pendulum_tsit5_f32": ("PENDULUM", "tsit5", torch.float32, 2e-2, 22, 100),
    "pendulum_rk4_f32": ("PENDULUM", "rk4", torch.float32, 2e-2, 22, 100),
    "msd_tsit5_f32": ("MASS_SPRING_DAMPER", "tsit5", torch.float32, 1e-4, 22, 100),
    "msd_euler_f64": ("MASS_SPRING_DAMPER", "euler", torch.float64, 1e-4, 21, 100),
    "tank_tsit5_f32": ("FLUID_TANK", "tsit5", torch.float32, 1e-3, 22, 100),
    "tank_euler_f64": ("FLUID_TANK", "euler", torch.float64, 1e-3, 21, 100),
    "cartpole_tsit5_f32": ("CART_POLE", "tsit5", torch.float32, 2e-2, 22, 100),
    "cartpole_euler_f64": ("CART_POLE", "euler", torch.float64, 2e-2, 21, 100),
    "acrobot_euler_f64": ("ACROBOT", "euler", torch.float64, 1e-3, 21, 100),
    "cartpole_rk4_f32": ("CART_POLE", "rk4", torch.float32, 2e-2, 22, 100),
    "cartpole_tsit5_f64": ("CART_POLE", "tsit5", torch.float64, 2e-2, 21, 100),
    "acrobot_rk4_f32": ("ACROBOT", "rk4", torch.float32, 1e-3, 22, 100),
    "pendulum_tsit5_f64": ("PENDULUM", "tsit5", torch.float64, 2e-2, 21, 100),
    "msd_rk4_f32": ("MASS_SPRING_DAMPER", "rk4", torch.float32, 1e-4, 22, 100),
}
ORACLE_NAME = {"PMSM": "pmsm", "PENDULUM": "pendulum", "MASS_SPRING_DAMPER": "mass_spring_damper", "CART_POLE": "cartpole",
               "ACROBOT": "acrobot", "FLUID_TANK": "fluid_tank"}


def synthetic_pmsm_lut(n_d=26, n_q=51):
    """A smooth saturating machine on the BRUSA current range (synthetic: the reference's motor data is not shipped)."""
    i_d, i_q = np.linspace(-250.0, 0.0, n_d)[None], np.linspace(-250.0, 250.0, n_q)[None]
    ID, IQ = np.meshgrid(i_d[0], i_q[0])
    sat = 1.0 / (1.0 + (ID / 300.0) ** 2 + (IQ / 280.0) ** 2)
    cross = 2e-5 * np.tanh(ID / 100.0) * np.tanh(IQ / 100.0)
    return dict(i_d_vec=i_d, i_q_vec=i_q, L_dd=0.37e-3 * (0.6 + 0.4 * sat), L_qq=1.2e-3 * (0.5 + 0.5 * sat), L_dq=cross,
                L_qd=cross.copy(), Psi_d=65.6e-3 + 0.37e-3 * 300.0 * np.arctan(ID / 300.0),
                Psi_q=1.2e-3 * 280.0 * np.arctan(IQ / 280.0))


def build_env(args, device, rank):
    import exciting_environments_amd as ex
    from exciting_environments_amd import EnvironmentRegistry

    reg, solver, dtype, tau, log2b, chunk = WORKLOADS[args.workload]
    B = args.batch or (1 << log2b)
    Kc = args.chunk or chunk
    solv = {"euler": ex.Euler(), "rk4": ex.RK4(), "tsit5": ex.Tsit5()}[solver]
    if reg == "PMSM_SAT":
        env = EnvironmentRegistry.PMSM.make(batch_size=B, tau=tau, solver=solv, dtype=dtype, device=device, saturated=True,
                                            motor_variant=ex.MotorVariant.BRUSA, pmsm_lut=synthetic_pmsm_lut())
        reg = "PMSM"
    else:
        env = getattr(EnvironmentRegistry, reg).make(batch_size=B, tau=tau, solver=solv, dtype=dtype, device=device)
    env.sim_ahead_semantics = args.semantics
    _, state = env.vmap_reset()
    g = torch.Generator(device=device)
    g.manual_seed(1236 + rank)
    u = lambda lo, hi: torch.rand(B, generator=g, dtype=dtype, device=device) * (hi - lo) + lo
    ps = state.physical_state
    if reg == "PMSM":  # stable region omega_el <= 600 rad/s
        ps.i_d = torch.full((B,), -125.0, dtype=dtype, device=device)
        ps.epsilon = u(-np.pi, np.pi)
        ps.omega_el = u(0.0, 600.0) if not getattr(env.env_properties, "saturated", False) else u(0.0, 150.0)
    elif reg == "PENDULUM":
        ps.theta, ps.omega = u(-np.pi, np.pi), u(-1.0, 1.0)
    elif reg == "CART_POLE":
        ps.theta, ps.omega = u(-0.2, 0.2), u(-0.1, 0.1)
    elif reg == "ACROBOT":
        ps.theta_1, ps.theta_2 = u(-np.pi, np.pi), u(-1.0, 1.0)
    elif reg == "FLUID_TANK":
        ps.height = u(0.5, 2.5)
    g.manual_seed(1237 + rank)
    if args.action_layout == "tiled":
        actions = env.new_actions_buffer(Kc, layout="tiled")
        actions.uniform_(-1, 1, generator=g)
    else:
        actions = env.new_actions_buffer(Kc) if args.action_layout == "lane_major" else torch.empty(
            (B, Kc, env.action_dim), dtype=dtype, device=device)
        flat = torch.rand(actions.numel(), generator=g, dtype=dtype, device=device) * 2 - 1
        actions.copy_(flat.view(Kc, env.action_dim, B).permute(2, 0, 1))
        del flat
    env.traj_layout = args.traj_layout
    env.env_major_workspace = not getattr(args, "no_workspace", False)
    env.env_major_fused = not getattr(args, "no_fused", False)
    env.store_state_trajectory = not getattr(args, "obs_only", False)
    if getattr(args, "no_pool", False):
        env.trajectory_pool = False
        env.trajectory_placement = "off"
    return env, state, actions, B, Kc, reg, solver, dtype


def host_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota if one is set."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return n


def cpu_baseline(args, reg, solver, dtype, tau, Kc):
    """The CPU oracle (C/OpenMP restatement pinned to the reference fixtures) timed on this host's cores on a
    bounded sample of the same workload. Reported beside the GPU number; it is a baseline, not the target."""
    import oracle

    name = ORACLE_NAME[reg]
    npdt = np.float32 if dtype == torch.float32 else np.float64
    Bc = 1 << 20  # B = 2^20 (chunks of Kc steps chained until ~cpu_seconds; 1000 steps = 10 chunks of 100)
    from exciting_environments_amd import EnvironmentRegistry

    env = getattr(EnvironmentRegistry, reg).make(batch_size=Bc, tau=tau, dtype=dtype, device="cpu")
    ep = env.env_properties
    params = {n: getattr(ep.static_params, n) for n in env.PARAM_FIELDS}
    pn = {n: (getattr(ep.physical_normalizations, n).min, getattr(ep.physical_normalizations, n).max) for n in env.STATE_FIELDS}
    an = {n: (getattr(ep.action_normalizations, n).min, getattr(ep.action_normalizations, n).max) for n in env.ACTION_FIELDS}
    props, keep = oracle.make_props(name, params, pn, an, npdt, Bc)
    rng = np.random.default_rng(1236)
    _, st = env.vmap_reset()
    st_np = [getattr(st.physical_state, n).numpy().astype(npdt) for n in env.STATE_FIELDS]
    if name == "pmsm":
        st_np[2] = rng.uniform(-np.pi, np.pi, Bc).astype(npdt)
        st_np[6] = rng.uniform(0, 600, Bc).astype(npdt)
    acts = rng.uniform(-1, 1, (Bc, Kc, env.action_dim)).astype(npdt)
    sem = oracle.SEM_AHEAD if args.semantics == "ahead" else oracle.SEM_STEP
    threads = host_cores()
    oracle.set_num_threads(threads)
    S, O = len(env.STATE_FIELDS), len(env.obs_description)
    bufs = (np.zeros((Bc, Kc + 1, O), dtype=npdt), [np.zeros((Bc, Kc + 1), dtype=npdt) for _ in range(S)],
            [np.zeros(Bc, dtype=npdt) for _ in range(S)])
    oracle.sim_ahead(name, solver, st_np, acts, props, tau, semantics=sem, out=bufs)  # warm-up (thread pool, pages)
    reps, t0 = 0, time.perf_counter()
    while True:
        _, _, last = oracle.sim_ahead(name, solver, st_np, acts, props, tau, semantics=sem, out=bufs)
        st_np = [a.copy() for a in last]
        reps += 1
        el = time.perf_counter() - t0
        if el > args.cpu_seconds or reps >= 10000:
            break
    return {
        "value": Bc * Kc * reps / el, "unit": "env-steps/s", "cores": threads, "kind": "port",
        "sample": f"oracle/liboracle.so (C+OpenMP restatement, env-major layout) {name} {solver} "
                  f"{np.dtype(npdt).name} B=2^20 x {Kc} steps x {reps} chained chunks, {el:.1f} s",
    }


def calibrate_placement(env, state, actions, args, Kc, B, dtype):
    """Same-run calibration, outside the timed region: how fast does HBM take the trajectory launch's traffic WITHOUT arithmetic
    over the very buffers a step writes (excenv_stream_pattern: same addresses, same order), how fast the same traffic with every
    stream inside one buffer (one physical region at a time: the slow level of the platform), and a plain fill of
    the same output buffers. Tells a slow box / placement from a slow kernel."""
    from exciting_environments_amd import _native

    if args.path != "sim_ahead" or args.traj_layout != "lane_major" or args.action_layout != "lane_major" or args.obs_only:
        return None
    isz = 4 if dtype == torch.float32 else 8
    if (B * isz) % 16 or Kc < 8:
        return None
    obs, states, _last = env.vmap_sim_ahead(state, actions, env.tau, env.tau)  # the set the next step would write
    leaves = [getattr(states.physical_state, n) for n in env.STATE_FIELDS]
    A, O, S = env.action_dim, obs.shape[2], len(leaves)
    rb = B * isz
    stream = _native.raw_stream(torch.cuda.current_device())
    rd, rd_rs = [actions.data_ptr() + c * rb for c in range(A)], [A * rb] * A
    ob = obs.data_ptr()
    obs_w, obs_rs = [ob + c * rb for c in range(O)], [O * rb] * O
    rows = Kc - 1

    def timed(fn, n=5):
        fn()
        ts = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts))

    pat_bytes = (A + O + S) * rb * rows
    t_pat = timed(lambda: _native.stream_pattern(rd, rd_rs, obs_w + [l.data_ptr() for l in leaves], obs_rs + [rb] * S, rb, rows, stream))
    W = O + S  # every stream inside the observation buffer read as [rows'][O + S][B]: one region at any time, nothing written twice
    rows1 = min(rows, (Kc + 1) * O // W)
    t_one = timed(lambda: _native.stream_pattern(rd, rd_rs, [ob + q * rb for q in range(W)], [W * rb] * W, rb, rows1, stream)) * rows / rows1
    flat = [obs.permute(1, 2, 0)] + [l.t() for l in leaves]  # the contiguous [rows, ., B] memory behind the views
    assert all(t.is_contiguous() for t in flat)
    fill_bytes = sum(t.numel() for t in flat) * isz

    def fill():
        for t in flat:
            t.fill_(0.0)

    t_fill = timed(fill)
    return {"same_run_pattern_gbs": pat_bytes / t_pat / 1e6, "same_run_pattern_one_buffer_gbs": pat_bytes / t_one / 1e6,
            "same_run_fill_gbs": fill_bytes / t_fill / 1e6, "same_run_pattern_ms": t_pat, "same_run_pattern_rows": rows}


def measure_traffic_live(args):
    """HBM bytes per launch of the dominant kernel, measured IN THIS RUN: before this process touches the GPU it runs the
    workload (tools/traffic_probe.py: a calibration copy of known size + three launches) under `rocprofv3 --pmc` in child
    processes — FETCH_SIZE and WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md §HBM prescribes (FETCH_SIZE counts
    half of a wide coalesced read stream on gfx950: doubled; WRITE_SIZE exact; both checked on the calibration copy of the same
    pass). Returns (bytes, description) or (None, reason). Bytes do not depend on where buffers sit, so a sibling process is a
    faithful measurement of the timed one's traffic. Any failure falls back to the committed passes (profiles/traffic.json)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "already running under a profiler"
    probe = os.path.join(ROOT, "tools", "traffic_probe.py")
    hot = "step_kernel" if args.path == "step" else ("sim_ahead_em" if args.traj_layout == "env_major" and args.action_layout == "env_major" else "sim_ahead_kernel")
    vals, calib = {}, {}
    tmp = tempfile.mkdtemp(prefix="excenv_traffic_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, probe,
                   "--workload", args.workload, "--traj-layout", args.traj_layout, "--action-layout", args.action_layout,
                   "--path", args.path]
            if args.batch:
                cmd += ["--batch", str(args.batch)]
            if args.chunk:
                cmd += ["--chunk", str(args.chunk)]
            env = dict(os.environ, TMPDIR="/tmp")
            # own process group, killed as a whole on a timeout (rocprofv3 starts the probe as its child)
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                    start_new_session=True)
            try:
                proc.communicate(timeout=120)
            except subprocess.TimeoutExpired:
                import signal

                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                proc.communicate()
                return None, f"rocprofv3 --pmc {counter} pass timed out"
            r = proc
            per_kernel = {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") != counter:
                        continue
                    per_kernel.setdefault(row.get("Kernel_Name", ""), []).append(float(row.get("Counter_Value", 0) or 0))
            # the matching kernel with the most dispatches (several instantiations can share the name's stem)
            hk = sorted((v for k, v in per_kernel.items() if hot in k), key=len, reverse=True)
            ck = sorted((v for k, v in per_kernel.items() if "trunc" in k.lower()), key=len, reverse=True)
            if r.returncode != 0 or not hk:
                return None, f"rocprofv3 --pmc {counter} pass gave no {hot} dispatch (rc {r.returncode})"
            vals[counter] = float(np.mean(hk[0]))                      # KiB per dispatch
            calib[counter] = float(np.mean(ck[0])) / float(1 << 20) if ck else None  # x the 1 GiB the copy moves
    except Exception as e:  # timeouts, missing files, parse errors: never fatal
        return None, f"live PMC passes failed: {type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # corrections from the calibration copy of the SAME passes (a 1 GiB coalesced read + write): the guide's gfx950 behaviour is
    # FETCH_SIZE = 0.5 x, WRITE_SIZE = 1.0 x; anything else (another ROCm, another counter definition) is not silently trusted
    cf, cw = calib.get("FETCH_SIZE"), calib.get("WRITE_SIZE")
    if cf is None or cw is None or abs(cf - 0.5) > 0.025 or abs(cw - 1.0) > 0.05:
        return None, f"calibration copy reads FETCH_SIZE {cf} x / WRITE_SIZE {cw} x its bytes (expected 0.5 / 1.0): counters not trusted"
    hbm = (vals["FETCH_SIZE"] / cf + vals["WRITE_SIZE"] / cw) * 1024.0
    desc = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes (separate child processes, before the timed "
            f"process touched the GPU) over tools/traffic_probe.py; per launch of {hot}: FETCH_SIZE {vals['FETCH_SIZE']:.0f} KiB "
            f"(/ {cf:.4f}, the factor the calibration copy of the same pass reads), WRITE_SIZE {vals['WRITE_SIZE']:.0f} KiB (/ {cw:.4f})")
    if args.action_layout == "env_major":
        desc += ("; row-major actions are read as 64-byte windows (DESIGN.md §4.1b): for those requests the factor is an upper bound "
                 f"(they are tallied at 64 bytes each; lower bound of the traffic {(vals['FETCH_SIZE'] + vals['WRITE_SIZE'] / cw) * 1024.0:.4e} B)")
    return hbm, desc


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5,
                    help="untimed steps; the first ones create and place the pooled output sets (core_env.py), so keep >= 4")
    ap.add_argument("--workload", default="pmsm_euler_f32", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="envs per GPU (default: the workload's 2^n)")
    ap.add_argument("--chunk", type=int, default=0, help="solver steps per launch")
    ap.add_argument("--semantics", default="ahead", choices=["ahead", "step"])
    ap.add_argument("--traj-layout", default="lane_major", choices=["lane_major", "env_major", "tiled"])
    ap.add_argument("--action-layout", default="lane_major", choices=["lane_major", "env_major", "tiled"])
    ap.add_argument("--gather", default="end", choices=["end", "chunk", "none"],
                    help="N>1: all-gather (RCCL) of observations — 'end': once, the final row of the last chunk, inside the "
                         "timed region; 'chunk': after every chunk on a side stream, overlapped; 'none'")
    ap.add_argument("--vec", type=int, default=0, help="envs per lane (0 auto)")
    ap.add_argument("--lds-pad", type=int, default=0, help="dynamic LDS bytes per workgroup (occupancy cap experiment)")
    ap.add_argument("--path", default="sim_ahead", choices=["sim_ahead", "step"],
                    help="sim_ahead: one persistent launch per bench step (headline); step: one vmap_step launch per bench step")
    ap.add_argument("--obs-only", action="store_true", help="skip the state trajectories (not the reference's full outputs)")
    ap.add_argument("--no-workspace", action="store_true", help="env-major buffers: no transposition workspace")
    ap.add_argument("--no-fused", action="store_true", help="env-major buffers: do not use the fused LDS time-tile kernel")
    ap.add_argument("--placement-candidates", type=int, default=0,
                    help="0 (default): the plain API, every step allocates its outputs like the reference's functional calls. "
                         "N > 0 (sim_ahead path, lane-major trajectories): allocate N output-buffer sets during set-up, keep the "
                         "one a probe launch runs fastest into (physical placement moves the kernel by up to 25 %%, DESIGN.md "
                         "§6) and write it again every step (vmap_sim_ahead(out=...))")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc child passes that measure roofline.traffic in this run (N = 1 only; "
                         "~20 s); the committed passes of profiles/traffic.json are replayed instead")
    ap.add_argument("--no-calibration", action="store_true", help="skip the same-run placement calibration after the timed region")
    ap.add_argument("--no-pool", action="store_true", help="every step allocates its outputs (no pooled sets, no placement check)")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) invoked directly: THIS process has not touched the GPU and never will — it starts
    one child process per GPU through torch.distributed.run (fresh interpreters, no exec of a GPU-initialised process),
    forwards their output (rank 0 prints the JSON line) and exits with the launcher's code."""
    import socket
    import subprocess

    one_gpu = os.environ.get("EXCENV_BENCH_ONE_GPU") == "1"
    have = torch.cuda.device_count()  # does not initialise the HIP runtime
    if have < args.gpus and not one_gpu:
        sys.exit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible "
                 "(rehearsal on one GPU: EXCENV_BENCH_ONE_GPU=1 EXCENV_BENCH_BACKEND=gloo)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=env)
    sys.exit(proc.returncode)


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        launch_ranks(args)  # never returns
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: WORLD_SIZE={world} does not match --gpus {args.gpus}")
    live_traffic = (None, "not requested")
    # the full default line only (quick A/B runs pass --no-cpu-baseline and skip this too); BEFORE anything here touches the GPU
    if world == 1 and not args.no_live_traffic and not args.no_cpu_baseline:
        live_traffic = measure_traffic_live(args)
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    # rehearsal knobs (single-GPU box): EXCENV_BENCH_ONE_GPU=1 puts every rank on GPU 0, EXCENV_BENCH_BACKEND=gloo
    # replaces RCCL (which refuses two ranks on one device). The driver's multi-GPU runs use neither.
    if os.environ.get("EXCENV_BENCH_ONE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("EXCENV_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from exciting_environments_amd import _native
    from exciting_environments_amd.distributed import ObservationGatherer

    env, state, actions, B, Kc, reg, solver, dtype = build_env(args, device, rank)
    if args.vec or args.lds_pad:
        env.launch_opts = _native.launch_opts(envs_per_lane=args.vec, lds_pad_bytes=args.lds_pad)
    use_gather = args.gather != "none" and (world > 1 or os.environ.get("EXCENV_BENCH_FORCE_GATHER") == "1")
    if use_gather and not dist.is_initialized():  # 1-rank rehearsal of the collective path
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": device} if backend == "nccl" else {}))
    ranks_seen = dist.get_world_size() if dist.is_initialized() else 1
    if ranks_seen != world:
        sys.exit(f"bench.py: the process group reports {ranks_seen} ranks, expected {world}")
    gatherer = ObservationGatherer(B * world) if use_gather else None
    gathered = None

    # output buffers of the trajectory path: one set, chosen among a few placements, written again by every step
    bufs, probe_ms = None, []
    if args.path != "step" and args.placement_candidates > 0 and args.traj_layout == "lane_major":
        bufs, probe_ms = env.new_trajectory_buffers(state, actions, env.tau, env.tau, candidates=args.placement_candidates)

    step_actions = [actions[:, k, :].contiguous() for k in range(min(Kc, 8))] if args.path == "step" else None
    step_count = [0]
    last_obs = [None]

    def final_row(obs):
        if obs.ndim == 2:
            return obs
        return obs[:, -1, :] if obs.ndim == 3 else obs[:, :, -1, :].reshape(B, -1)

    def one_step(st):
        nonlocal gathered
        if args.path == "step":
            obs, last = env.vmap_step(st, step_actions[step_count[0] % len(step_actions)])
            step_count[0] += 1
            last_obs[0] = obs
            return last
        nonlocal bufs
        if bufs is not None:
            bufs = env.vmap_sim_ahead(st, actions, env.tau, env.tau, out=bufs)
            obs, states, last = bufs
        else:
            obs, states, last = env.vmap_sim_ahead(st, actions, env.tau, env.tau)
        last_obs[0] = obs
        if gatherer is not None and args.gather == "chunk":
            gatherer.wait()  # previous chunk's gather must have drained before its buffer is reused
            gathered = gatherer.start(final_row(obs), gathered)
        return last

    # set-up: the first calls create (and place) the pooled output sets of the trajectory path; with fewer than six warm-up
    # steps they are made here, outside warm-up and timed region alike
    setup_steps = max(0, 6 - args.warmup) if args.path != "step" else 0  # with fewer than six warm-up steps: this many extra, untimed
    for _ in range(setup_steps):
        state = one_step(state)
    # the pooled output sets are compared by the times of their real launches and a clearly slower one is replaced (core_env.py,
    # "large trajectory outputs"): step until that is over, so that no placement search falls into the timed region
    settle_steps = 0
    while args.path != "step" and not getattr(env, "trajectory_placement_settled", True) and settle_steps < 24:
        state = one_step(state)
        torch.cuda.synchronize()
        settle_steps += 1
    for _ in range(args.warmup):
        state = one_step(state)
    if gatherer is not None:
        # warm the collective up (RCCL sets up its channels / buffers on first use) outside the timed region
        if last_obs[0] is None:
            state = one_step(state)
        gathered = gatherer.start(final_row(last_obs[0]), gathered)
        gatherer.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()

    # one HIP-event pair per step — except on the one-kernel-per-step path at large batches, where a pair of event records between
    # two 70-microsecond kernels is itself a fifth of the step (measured: 92 us per step with them, 73.4 us for the same chained
    # loop without, tools/step_enqueue_probe.py): there one pair brackets each group of EV_GROUP steps
    EV_GROUP = 16 if args.path == "step" else 1
    n_groups = (args.steps + EV_GROUP - 1) // EV_GROUP
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_groups)]
    ev_steps = [min(EV_GROUP, args.steps - g * EV_GROUP) for g in range(n_groups)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        if k % EV_GROUP == 0:
            ev[k // EV_GROUP][0].record()
        state = one_step(state)
        if k % EV_GROUP == EV_GROUP - 1 or k == args.steps - 1:
            ev[k // EV_GROUP][1].record()
    gather_ms = None
    if gatherer is not None:
        if args.gather == "end" and last_obs[0] is not None:  # reassemble the global observation batch once
            torch.cuda.synchronize()  # so that the collective's own time can be told from the stepping time
            tg = time.perf_counter()
            gathered = gatherer.start(final_row(last_obs[0]), gathered)
            gatherer.wait()
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - tg) * 1e3
        else:
            gatherer.wait()
    torch.cuda.synchronize()
    steps_elapsed = time.perf_counter() - t0  # this rank, before waiting for the others
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    finite = bool(torch.isfinite(state.physical_state.__dict__[env.STATE_FIELDS[-1]]).all())
    gather_ok = None
    if gathered is not None and last_obs[0] is not None:  # this rank's slice of the gathered batch is its own final row
        gather_ok = bool(torch.equal(gathered[rank * B:(rank + 1) * B], final_row(last_obs[0])))
    # HIP events on the launch stream (torch's current stream is the stream the C ABI is handed); with the gather enabled
    # the bracket also holds the (asynchronous) enqueue of the collective but no wait for it
    per_step_ms = [a.elapsed_time(b) / n for (a, b), n in zip(ev, ev_steps)]  # per step (a group's average where steps share a pair)
    kernel_ms = float(np.average(per_step_ms, weights=ev_steps)) if args.steps else float("nan")
    # every rank's own numbers (a slow rank decides the MAX-over-ranks time): kernel ms min / median / max, the rank's wall
    # time for its steps (+ its share of the end gather), gathered to rank 0
    mine = [float(np.min(per_step_ms)), float(np.median(per_step_ms)), float(np.max(per_step_ms)), steps_elapsed * 1e3,
            -1.0 if gather_ms is None else gather_ms] if args.steps else [float("nan")] * 5
    per_rank = [mine]
    if world > 1:
        t = torch.tensor(mine, dtype=torch.float64, device=device)
        allr = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allr, t)
        per_rank = [[float(x) for x in r.cpu()] for r in allr]
    if args.path == "step":
        Kc = 1
        bytes_per_step = _native.step_bytes(env.ENV_ID, dtype)
    else:
        bytes_per_step = _native.sim_ahead_bytes(env.ENV_ID, dtype, not args.obs_only)
    algo_bytes = bytes_per_step * B * Kc
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if args.steps else float("nan")

    # HBM traffic per launch from the PMC counters cannot be collected inside this process: it is REPLAYED from the
    # committed rocprofv3 --pmc passes (tools/traffic_probe.py -> profiles/traffic.json) for the same workload key
    traffic, traffic_source = live_traffic if live_traffic[0] is not None else (None, None)
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if traffic is None and os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = (f"{args.workload}|B={B}|chunk={Kc}|{args.traj_layout}" + ("|step" if args.path == "step" else "")
                   + ("|obs_only" if args.obs_only else ""))  # (no committed PMC pass for observations-only launches: traffic stays null)
            if key in tj:
                traffic = tj[key]["hbm_bytes_per_launch"]
                traffic_source = ("replayed from profiles/traffic.json (" + tj[key].get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes")
                                  + f"), not measured in this run ({live_traffic[1]})")
        except Exception:
            traffic = None

    calib = None
    if rank == 0 and not args.no_calibration:
        calib = calibrate_placement(env, state, actions, args, Kc, B, dtype)
    if rank == 0:
        total_steps = B * world * Kc * args.steps
        out = {
            "metric": "env-steps/sec + achieved HBM GB/s, PMSM Euler fp32 batch=2^22, 1/2/4/8 GPU"
            if args.workload == "pmsm_euler_f32" else f"env-steps/sec, {args.workload}",
            "value": total_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if dtype == torch.float32 else "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{reg} {solver} {'fp32' if dtype == torch.float32 else 'fp64'} vmap_{args.path}, "
                            f"batch {B} per GPU, {Kc} solver steps per launch, "
                            + ("observations only" if args.obs_only else "full outputs (obs + state trajectories)"),
                "batch_per_gpu": B, "global_batch": B * world, "chunk_steps": Kc, "semantics": args.semantics,
                "traj_layout": args.traj_layout, "action_layout": args.action_layout,
                "parallelism": f"batch-sharded x{world}" + (
                    "" if gatherer is None else (", one all-gather of the final observations at the end" if args.gather == "end"
                                                 else ", all-gather(final obs) after every chunk, overlapped")),
                "ranks_seen": ranks_seen,
                "backend": (dist.get_backend() if dist.is_initialized() else None),
                "collective": (gatherer.collective if gatherer is not None else None),
                "gathered_slice_matches_local": gather_ok,
                "outputs_finite": finite, "setup_steps": setup_steps, "placement_settle_steps": settle_steps,
                "pooled_set_steady_ms": [None if t.steady_ms is None else round(t.steady_ms, 4) for t in env._placement.sets],
                "pooled_set_first_launch_ms": [None if t.first_ms is None else round(t.first_ms, 4) for t in env._placement.sets],
                "output_buffers": (("library-pooled output sets: a set is written again once nothing refers to it (the plain "
                                    "functional API, core_env.py trajectory sets); sets made this run: "
                                    f"{len(env._placement.sets)}" if getattr(env, "trajectory_pool", False)
                                    else "fresh allocation per step") if bufs is None else
                                   f"one set written again every step (vmap_sim_ahead(out=...)), chosen during set-up as the fastest "
                                   f"of {len(probe_ms) or 1} placements; probe launch ms per placement (rank 0): "
                                   + ", ".join(f"{t:.3f}" for t in probe_ms)),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "kernel": ("step_kernel" if args.path == "step" else ("sim_ahead_emr_kernel / sim_ahead_em_kernel (env-major fused forms)" if args.traj_layout == "env_major" and args.action_layout == "env_major" else "sim_ahead_kernel")), "kernel_ms": kernel_ms,
                "kernel_ms_min_median_max": ([float(np.min(per_step_ms)), float(np.median(per_step_ms)), float(np.max(per_step_ms))]
                                             if args.steps else None),  # spread over the timed steps (buffer placement)
                "kernel_ms_spread_pct": (100.0 * (float(np.max(per_step_ms)) - float(np.min(per_step_ms))) / float(np.median(per_step_ms))
                                         if args.steps else None),
                "kernel_ms_per_step": [round(float(x), 4) for x in per_step_ms],
                "algorithmic_bytes_per_env_step": bytes_per_step,
                "algorithmic_bytes_per_launch": algo_bytes, "frac_of_measured_copy_peak_6290": achieved / 6290.0,
            },
        }
        if calib is not None:  # same-run calibration of the placement (outside the timed region)
            out["roofline"].update(calib)
            out["roofline"]["kernel_vs_same_run_pattern"] = achieved / calib["same_run_pattern_gbs"]
        out["roofline"]["placement_check"] = getattr(env, "last_placement", None)
        # per-rank visibility: [kernel ms min, median, max, wall ms of this rank's timed steps, ms of the end all-gather or -1]
        out["per_rank_kernel_ms"] = [[round(x, 4) for x in r[:3]] for r in per_rank]
        out["per_rank_steps_wall_ms"] = [round(r[3], 3) for r in per_rank]
        out["gather_ms"] = (None if all(r[4] < 0 for r in per_rank) else max(r[4] for r in per_rank))
        # `value` keeps the contract (everything inside the timed region, the end all-gather included). The collective is ONE
        # per run whatever --steps is (1.07 GB received per rank at 8 x 2^22 environments: 3 - 10 ms over xGMI against
        # steps x 5 ms of stepping), so its share shrinks with the number of steps: the rate without it is reported beside it.
        if out["gather_ms"] is not None and args.steps:
            out["value_without_end_gather"] = total_steps / max(elapsed - out["gather_ms"] * 1e-3, 1e-9)
            out["end_gather_share_of_timed_region"] = out["gather_ms"] * 1e-3 / elapsed
        med_rank = float(np.median([r[1] for r in per_rank]))
        slow_rank = float(np.max([r[1] for r in per_rank]))
        # whole-job rate if every rank ran its launches at the median rank's / the slowest rank's median kernel time (no gather):
        # the ingredients of a weak-scaling comparison; the efficiency itself is the driver's to compute from per-N values
        out["value_at_median_rank_kernel"] = B * world * Kc / (med_rank * 1e-3) if args.steps else None
        out["value_at_slowest_rank_kernel"] = B * world * Kc / (slow_rank * 1e-3) if args.steps else None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, reg, solver, dtype, env.tau, Kc)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
