"""ctypes binding of libexcenv_hip.so (C ABI: include/excenv.h).

This is the only compute backend of the package. There is no CPU or PyTorch fallback: if the
shared library is missing, or no HIP device is present when a kernel entry point is called, the
call raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

import torch

MAX_STATE, MAX_ACTION, MAX_STATIC, MAX_CONTROL = 8, 2, 9, 8
LAYOUT_ENV_MAJOR, LAYOUT_LANE_MAJOR, LAYOUT_TILED = 0, 1, 2
TILE = 1024  # EXCENV_TILE
SEM_STEP, SEM_AHEAD = 0, 1
F32, F64 = 0, 1
ABI_VERSION = 7

_LIB_PATH = os.environ.get(  # EXCENV_HIP_LIB: A/B-test another build of the same library (tuning experiments)
    "EXCENV_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libexcenv_hip.so"))


class Param(ctypes.Structure):
    _fields_ = [("value", ctypes.c_double), ("per_env", ctypes.c_void_p)]


class PmsmLut(ctypes.Structure):
    _fields_ = [("n_d", ctypes.c_int32), ("n_q", ctypes.c_int32), ("grid_d", ctypes.c_void_p),
                ("grid_q", ctypes.c_void_p), ("tables", ctypes.c_void_p)]


class Props(ctypes.Structure):
    _fields_ = [
        ("static_params", Param * MAX_STATIC),
        ("state_min", Param * MAX_STATE),
        ("state_max", Param * MAX_STATE),
        ("action_min", Param * MAX_ACTION),
        ("action_max", Param * MAX_ACTION),
        ("pmsm_lut", ctypes.POINTER(PmsmLut)),
    ]


class LaunchOpts(ctypes.Structure):
    """excenv_launch_opts_t: per-call launch shaping (None / NULL = defaults)."""

    _fields_ = [("envs_per_lane", ctypes.c_int32), ("env_major_mode", ctypes.c_int32), ("lds_pad_bytes", ctypes.c_int32),
                ("flags", ctypes.c_int32)]


class TrajGym(ctypes.Structure):
    """excenv_traj_gym_t: optional reward / terminated / truncated trajectories of excenv_sim_ahead."""

    _fields_ = [("reward", ctypes.c_void_p), ("terminated", ctypes.c_void_p), ("truncated", ctypes.c_void_p)]


class Control(ctypes.Structure):
    _fields_ = [
        ("n_control", ctypes.c_int32),
        ("control_idx", ctypes.c_int32 * MAX_CONTROL),
        ("reference", ctypes.c_void_p * MAX_CONTROL),
        ("obs_reference", ctypes.c_void_p * MAX_CONTROL),  # gym_step only: NULL = the observation shows `reference`
    ]


_lib = None


def library_path() -> str:
    return _LIB_PATH


def lib():
    """Load libexcenv_hip.so (built in-tree by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError(
                f"{_LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C exciting-environments_amd/csrc). "
                "There is no CPU fallback."
            )
        l = ctypes.CDLL(_LIB_PATH)
        l.excenv_last_error.restype = ctypes.c_char_p
        l.excenv_last_launch.restype = ctypes.c_char_p
        l.excenv_abi_version.restype = ctypes.c_int
        l.excenv_step_bytes.restype = ctypes.c_int64
        l.excenv_sim_ahead_bytes.restype = ctypes.c_int64
        l.excenv_sim_ahead_workspace_bytes.restype = ctypes.c_int64
        l.excenv_truncated_width.restype = ctypes.c_int32
        for fn in ("excenv_step", "excenv_gym_step", "excenv_sim_ahead", "excenv_sim_ahead_ws", "excenv_transpose", "excenv_env_dims",
                   "excenv_probe_math", "excenv_probe_div", "excenv_rew_trunc_term", "excenv_state_from_observation",
                   "excenv_update_ref", "excenv_update_ref_to", "excenv_random_state", "excenv_observe", "excenv_stream_pattern",
                   "excenv_allgather"):
            getattr(l, fn).restype = ctypes.c_int
        vp, ci, cl, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
        # typed prototypes: plain Python ints / floats / None / byref() pass without per-call ctypes wrapping
        l.excenv_step.argtypes = [ci, ci, ci, cl, vp, vp, cd, vp, vp, vp, vp, vp, vp]
        l.excenv_gym_step.argtypes = [ci, ci, ci, cl, vp, vp, cd, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        l.excenv_sim_ahead_ws.argtypes = [ci, ci, ci, cl, cl, ctypes.c_int32, vp, vp, cd, cd, vp, vp, ci, vp, vp, ci, vp, ci, vp,
                                          vp, cl, vp, vp]
        l.excenv_stream_pattern.argtypes = [ctypes.c_int32, vp, vp, ctypes.c_int32, vp, vp, cl, cl, ctypes.c_int32, vp]
        l.excenv_sim_ahead_fuses_actions.restype = ctypes.c_int
        l.excenv_sim_ahead_fuses_actions.argtypes = [ci, ci, ci, cl, cl, vp, ctypes.c_int32, ci, ci, ci, vp, vp]
        if l.excenv_abi_version() != ABI_VERSION:
            raise ImportError("libexcenv_hip.so: ABI version mismatch")
        _lib = l
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = lib().excenv_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")


def allgather(nccl_comm: int, send: torch.Tensor, recv: torch.Tensor):
    """excenv_allgather: ncclAllGather of `send` (this rank's contiguous slice) into `recv` ([world * send.numel()]) on torch's
    current stream, through the caller's ncclComm_t handle (an integer address). The Python mirror's ObservationGatherer uses
    torch.distributed instead; this is the entry point a non-torch binder would call."""
    _require_device(send, "excenv_allgather")
    assert send.is_contiguous() and recv.is_contiguous() and send.dtype == recv.dtype
    with _on_device(send.device):
        rc = lib().excenv_allgather(ctypes.c_void_p(nccl_comm), ctypes.c_int(dtype_id(send.dtype)), ctypes.c_void_p(send.data_ptr()),
                                    ctypes.c_void_p(recv.data_ptr()), ctypes.c_int64(send.numel()),
                                    ctypes.c_void_p(_raw_stream(send.device)))
    _check(rc, "excenv_allgather")


def last_launch() -> str:
    """excenv_last_launch(): which trajectory-kernel form the last sim_ahead call of this thread enqueued."""
    return lib().excenv_last_launch().decode("utf-8", "replace")


def dtype_id(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.float64:
        return F64
    raise TypeError(f"unsupported dtype {dtype}: the kernels compute in float32 or float64")


def _require_device(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: tensors must live on a HIP device (got {t.device}). The batched ODE kernels have no CPU fallback."
        )


def _raw_stream(device: torch.device) -> int:
    """hipStream_t of torch's current stream on `device` (the fast private accessor when torch has it)."""
    try:
        return torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())
    except AttributeError:  # pragma: no cover
        return torch.cuda.current_stream(device).cuda_stream


class _on_device:
    """`with torch.cuda.device(d)` only when d is not already current (saves a few us per launch)."""

    def __init__(self, device):
        idx = device.index
        self.ctx = None if (idx is None or idx == torch.cuda.current_device()) else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _ptrs(tensors: Sequence[torch.Tensor]):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def ptr_array(addresses: Sequence[int]):
    """void*[n] from raw device addresses (pre-built once per output slot on the vmap_step fast path)."""
    return (ctypes.c_void_p * len(addresses))(*addresses)


_get_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def raw_stream(device_index: int) -> int:
    """The current HIP stream of the (current) device as an integer handle."""
    return _get_raw_stream(device_index) if _get_raw_stream is not None else torch.cuda.current_stream().cuda_stream


def step_raw(env_id, solver_id, dtype_code, B, props_ref, control_ref, tau, in_ptrs, action_ptr, out_ptrs, obs_ptr, opts_ref,
             stream, gym=None):
    """excenv_step / excenv_gym_step with every argument already in its C form (pointer arrays and byref()s cached by the
    caller). `gym` = (reward_ptr, terminated_ptr, truncated_ptr) selects excenv_gym_step. `stream` = raw_stream() of the
    device the buffers live on, which the caller has made current."""
    if gym is None:
        rc = _lib.excenv_step(env_id, solver_id, dtype_code, B, props_ref, control_ref, tau, in_ptrs, action_ptr, out_ptrs,
                              obs_ptr, opts_ref, stream)
    else:
        rc = _lib.excenv_gym_step(env_id, solver_id, dtype_code, B, props_ref, control_ref, tau, in_ptrs, action_ptr, out_ptrs,
                                  obs_ptr, gym[0], gym[1], gym[2], opts_ref, stream)
    if rc != 0:
        _check(rc, "excenv_step" if gym is None else "excenv_gym_step")


def stream_pattern(read_ptrs, read_row_strides, write_ptrs, write_row_strides, row_bytes: int, rows: int, stream: int,
                   nontemporal: bool = True):
    """excenv_stream_pattern: the trajectory kernels' access shape without arithmetic over raw device addresses (calibration;
    whatever the write streams point at is overwritten with meaningless values)."""
    nr, nw = len(read_ptrs), len(write_ptrs)
    rc = lib().excenv_stream_pattern(
        nr, (ctypes.c_void_p * max(nr, 1))(*read_ptrs), (ctypes.c_int64 * max(nr, 1))(*read_row_strides),
        nw, (ctypes.c_void_p * max(nw, 1))(*write_ptrs), (ctypes.c_int64 * max(nw, 1))(*write_row_strides),
        int(row_bytes), int(rows), int(bool(nontemporal)), stream)
    if rc != 0:
        _check(rc, "excenv_stream_pattern")


_hip = None


def _hip_runtime():
    """The HIP runtime torch already loaded (only hipMalloc / hipFree are used: spacer allocations that must not go through
    torch's caching allocator, core_env.py trajectory placement)."""
    global _hip
    if _hip is None:
        # the very file this process has mapped (torch ships its own copy): opening another copy would start a second runtime
        loaded = None
        try:
            with open("/proc/self/maps") as f:
                for line in f:
                    if "libamdhip64" in line:
                        loaded = line.split()[-1]
                        break
        except OSError:
            loaded = None
        for name in ([loaded] if loaded else []):
            try:
                h = ctypes.CDLL(name)
                h.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
                h.hipMalloc.restype = ctypes.c_int
                h.hipFree.argtypes = [ctypes.c_void_p]
                h.hipFree.restype = ctypes.c_int
                _hip = h
                break
            except OSError:
                continue
        if _hip is None:
            _hip = False
    return _hip or None


def raw_malloc(nbytes: int):
    """hipMalloc outside torch's allocator; None when it fails (out of memory) or the runtime cannot be reached."""
    h = _hip_runtime()
    if h is None:
        return None
    p = ctypes.c_void_p()
    if h.hipMalloc(ctypes.byref(p), int(nbytes)) != 0 or not p.value:
        return None
    return p.value


def raw_free(ptr):
    h = _hip_runtime()
    if h is not None and ptr:
        h.hipFree(ctypes.c_void_p(ptr))


def env_dims(env_id: int):
    S, A, O, P = (ctypes.c_int32() for _ in range(4))
    _check(lib().excenv_env_dims(env_id, ctypes.byref(S), ctypes.byref(A), ctypes.byref(O), ctypes.byref(P)), "excenv_env_dims")
    return S.value, A.value, O.value, P.value


def step_bytes(env_id: int, dtype: torch.dtype) -> int:
    return int(lib().excenv_step_bytes(env_id, dtype_id(dtype)))


def sim_ahead_bytes(env_id: int, dtype: torch.dtype, with_state_traj: bool = True) -> int:
    return int(lib().excenv_sim_ahead_bytes(env_id, dtype_id(dtype), int(with_state_traj)))


OPT_NO_FUSED_ACTIONS = 1  # EXCENV_OPT_NO_FUSED_ACTIONS


def launch_opts(envs_per_lane: int = 0, env_major_mode: int = 0, lds_pad_bytes: int = 0, flags: int = 0) -> LaunchOpts:
    return LaunchOpts(int(envs_per_lane), int(env_major_mode), int(lds_pad_bytes), int(flags))


def sim_ahead_fuses_actions(env: int, solver: int, dtype: torch.dtype, B: int, K: int, props, n_control: int, with_gym: bool,
                            action_layout: int, traj_layout: int, actions_ptr: int, opts: Optional[LaunchOpts]) -> bool:
    """excenv_sim_ahead_fuses_actions: the trajectory kernel reads these row-major actions itself (no workspace needed)."""
    return bool(lib().excenv_sim_ahead_fuses_actions(env, solver, dtype_id(dtype), B, K, ctypes.byref(props), n_control, int(with_gym),
                                                     action_layout, traj_layout, actions_ptr,
                                                     ctypes.byref(opts) if opts is not None else None))


def _opts_ref(opts: Optional[LaunchOpts]):
    return ctypes.byref(opts) if opts is not None else None


def make_control(control_idx: Sequence[int], refs: Sequence[torch.Tensor],
                 obs_refs: Optional[Sequence[torch.Tensor]] = None) -> Optional[Control]:
    if not control_idx:
        return None
    c = Control()
    c.n_control = len(control_idx)
    for j, (f, r) in enumerate(zip(control_idx, refs)):
        c.control_idx[j] = f
        c.reference[j] = r.data_ptr()
        if obs_refs is not None:
            c.obs_reference[j] = obs_refs[j].data_ptr()
    return c


def step(env_id, solver_id, dtype, B, props: Props, control: Optional[Control], tau: float,
         state_in: Sequence[torch.Tensor], action: torch.Tensor, state_out: Sequence[torch.Tensor],
         obs: torch.Tensor, opts: Optional[LaunchOpts] = None):
    _require_device(action, "vmap_step")
    with _on_device(action.device):
        stream = _raw_stream(action.device)
        rc = lib().excenv_step(
            ctypes.c_int(env_id), ctypes.c_int(solver_id), ctypes.c_int(dtype_id(dtype)), ctypes.c_int64(B),
            ctypes.byref(props), ctypes.byref(control) if control is not None else None, ctypes.c_double(tau),
            _ptrs(state_in), ctypes.c_void_p(action.data_ptr()), _ptrs(state_out), ctypes.c_void_p(obs.data_ptr()),
            _opts_ref(opts), ctypes.c_void_p(stream),
        )
    _check(rc, "excenv_step")


def truncated_width(env_id: int, n_control: int) -> int:
    return int(lib().excenv_truncated_width(ctypes.c_int(env_id), ctypes.c_int32(n_control)))


def gym_step(env_id, solver_id, dtype, B, props: Props, control: Optional[Control], tau: float,
             state_in: Sequence[torch.Tensor], action: torch.Tensor, state_out: Sequence[torch.Tensor],
             obs: torch.Tensor, reward: torch.Tensor, terminated: torch.Tensor, truncated: torch.Tensor,
             opts: Optional[LaunchOpts] = None):
    _require_device(action, "gym_step")
    with _on_device(action.device):
        stream = _raw_stream(action.device)
        rc = lib().excenv_gym_step(
            ctypes.c_int(env_id), ctypes.c_int(solver_id), ctypes.c_int(dtype_id(dtype)), ctypes.c_int64(B),
            ctypes.byref(props), ctypes.byref(control) if control is not None else None, ctypes.c_double(tau),
            _ptrs(state_in), ctypes.c_void_p(action.data_ptr()), _ptrs(state_out), ctypes.c_void_p(obs.data_ptr()),
            ctypes.c_void_p(reward.data_ptr()), ctypes.c_void_p(terminated.data_ptr()),
            ctypes.c_void_p(truncated.data_ptr()), _opts_ref(opts), ctypes.c_void_p(stream),
        )
    _check(rc, "excenv_gym_step")


def sim_ahead(env_id, solver_id, dtype, B, K, substeps, props: Props, control: Optional[Control],
              obs_stepsize: float, env_tau: float, state_in: Sequence[torch.Tensor], actions: torch.Tensor,
              action_layout: int, obs_traj: torch.Tensor, state_traj: Optional[Sequence[torch.Tensor]],
              traj_layout: int, last_state: Sequence[torch.Tensor], semantics: int,
              workspace: Optional[torch.Tensor] = None, opts: Optional[LaunchOpts] = None, gym=None):
    """gym: None or (reward, terminated, truncated) device tensors in the trajectory layout (excenv_traj_gym_t)."""
    _require_device(obs_traj, "vmap_sim_ahead")
    g = None
    if gym is not None:
        g = TrajGym(gym[0].data_ptr(), gym[1].data_ptr(), gym[2].data_ptr())
    with _on_device(obs_traj.device):
        stream = _raw_stream(obs_traj.device)
        rc = lib().excenv_sim_ahead_ws(
            ctypes.c_int(env_id), ctypes.c_int(solver_id), ctypes.c_int(dtype_id(dtype)), ctypes.c_int64(B),
            ctypes.c_int64(K), ctypes.c_int32(substeps), ctypes.byref(props),
            ctypes.byref(control) if control is not None else None, ctypes.c_double(obs_stepsize),
            ctypes.c_double(env_tau), _ptrs(state_in), ctypes.c_void_p(actions.data_ptr() if K > 0 else None),
            ctypes.c_int(action_layout), ctypes.c_void_p(obs_traj.data_ptr()),
            _ptrs(state_traj) if state_traj is not None else None, ctypes.c_int(traj_layout), _ptrs(last_state),
            ctypes.c_int(semantics), ctypes.byref(g) if g is not None else None,
            ctypes.c_void_p(workspace.data_ptr() if workspace is not None else None),
            ctypes.c_int64(workspace.numel() * workspace.element_size() if workspace is not None else 0),
            _opts_ref(opts), ctypes.c_void_p(stream),
        )
    _check(rc, "excenv_sim_ahead")


def sim_ahead_raw(env_id, solver_id, dtype_code, B, K, substeps, props_ref, control_ref, obs_stepsize, env_tau, in_ptrs,
                  actions_ptr, action_layout, obs_ptr, traj_ptrs, traj_layout, last_ptrs, semantics, ws_ptr, ws_bytes, opts_ref,
                  stream, gym_ref=None):
    """excenv_sim_ahead_ws with every argument already in its C form (the fast path of vmap_sim_ahead: lane-major
    trajectories; gym_ref: None or byref(TrajGym)). The caller has made the buffers' device current."""
    rc = _lib.excenv_sim_ahead_ws(env_id, solver_id, dtype_code, B, K, substeps, props_ref, control_ref, obs_stepsize, env_tau,
                                  in_ptrs, actions_ptr, action_layout, obs_ptr, traj_ptrs, traj_layout, last_ptrs, semantics,
                                  gym_ref, ws_ptr, ws_bytes, opts_ref, stream)
    if rc != 0:
        _check(rc, "excenv_sim_ahead")


def rew_trunc_term(env_id, dtype, B, rows, props: Props, control: Optional[Control], ref_strides: Optional[Sequence[int]],
                   state_traj: Sequence[torch.Tensor], s_sb: int, s_sk: int, reward: torch.Tensor, terminated: torch.Tensor,
                   truncated: torch.Tensor, out_layout: int):
    """excenv_rew_trunc_term: reward / terminated / truncated of a stored trajectory (one thread per (env, row))."""
    _require_device(truncated, "vmap_generate_rew_trunc_term_ahead")
    rs = (ctypes.c_int64 * len(ref_strides))(*ref_strides) if ref_strides else None
    with _on_device(truncated.device):
        rc = lib().excenv_rew_trunc_term(
            ctypes.c_int(env_id), ctypes.c_int(dtype_id(dtype)), ctypes.c_int64(B), ctypes.c_int64(rows), ctypes.byref(props),
            ctypes.byref(control) if control is not None else None, rs, _ptrs(state_traj), ctypes.c_int64(s_sb),
            ctypes.c_int64(s_sk), ctypes.c_void_p(reward.data_ptr() if rows > 1 else None),
            ctypes.c_void_p(terminated.data_ptr() if rows > 1 else None), ctypes.c_void_p(truncated.data_ptr()),
            ctypes.c_int(out_layout), ctypes.c_void_p(_raw_stream(truncated.device)))
    _check(rc, "excenv_rew_trunc_term")


def state_from_observation(env_id, dtype, B, props: Props, control_idx: Sequence[int], obs: torch.Tensor,
                           state_out: Sequence[torch.Tensor], reference_out: Sequence[torch.Tensor]):
    """excenv_state_from_observation: obs [B, O + n_control] -> denormalised state leaves (+ controlled reference leaves)."""
    _require_device(obs, "vmap_generate_state_from_observation")
    nc = len(control_idx)
    with _on_device(obs.device):
        rc = lib().excenv_state_from_observation(
            ctypes.c_int(env_id), ctypes.c_int(dtype_id(dtype)), ctypes.c_int64(B), ctypes.byref(props), ctypes.c_int32(nc),
            (ctypes.c_int32 * nc)(*control_idx) if nc else None, ctypes.c_void_p(obs.data_ptr()), _ptrs(state_out),
            _ptrs(reference_out) if nc else None, ctypes.c_void_p(_raw_stream(obs.device)))
    _check(rc, "excenv_state_from_observation")


def update_ref(env_id, dtype, B, props: Props, control_idx: Sequence[int], reference: Sequence[torch.Tensor],
               keys: torch.Tensor, hold: torch.Tensor, hold_min: int, hold_max: int):
    """excenv_update_ref: in-place reference redraw + hold countdown (GymWrapper.update_ref). keys: int64 [B, 2], hold: int64 [B]."""
    _require_device(keys, "GymWrapper.update_ref")
    assert keys.dtype == torch.int64 and keys.is_contiguous() and hold.dtype == torch.int64 and hold.is_contiguous()
    nc = len(control_idx)
    with _on_device(keys.device):
        rc = lib().excenv_update_ref(
            ctypes.c_int(env_id), ctypes.c_int(dtype_id(dtype)), ctypes.c_int64(B), ctypes.byref(props), ctypes.c_int32(nc),
            (ctypes.c_int32 * nc)(*control_idx) if nc else None, _ptrs(reference) if nc else None,
            ctypes.c_void_p(keys.data_ptr()), ctypes.c_void_p(hold.data_ptr()), ctypes.c_int32(hold_min), ctypes.c_int32(hold_max),
            ctypes.c_void_p(_raw_stream(keys.device)))
    _check(rc, "excenv_update_ref")


def observe(env_id, dtype, B, props: Props, control: Optional[Control], state: Sequence[torch.Tensor], obs: torch.Tensor):
    """excenv_observe: generate_observation for a batch of states in one launch (obs: [B, O + n_control] row-major)."""
    _require_device(obs, "generate_observation")
    with _on_device(obs.device):
        rc = lib().excenv_observe(ctypes.c_int(env_id), ctypes.c_int(dtype_id(dtype)), ctypes.c_int64(B), ctypes.byref(props),
                                  ctypes.byref(control) if control is not None else None, _ptrs(state),
                                  ctypes.c_void_p(obs.data_ptr()), ctypes.c_void_p(_raw_stream(obs.device)))
    _check(rc, "excenv_observe")


def update_ref_to(env_id, dtype, B, props: Props, control_idx: Sequence[int], reference_in: Sequence[torch.Tensor],
                  keys_in: torch.Tensor, hold_in: torch.Tensor, reference_out: Sequence[torch.Tensor], keys_out: torch.Tensor,
                  hold_out: torch.Tensor, hold_min: int, hold_max: int):
    """excenv_update_ref_to: out-of-place reference redraw + hold countdown (inputs untouched, one launch, no copies)."""
    _require_device(keys_in, "GymWrapper.update_ref")
    for t in (keys_in, hold_in, keys_out, hold_out):
        assert t.dtype == torch.int64 and t.is_contiguous()
    nc = len(control_idx)
    with _on_device(keys_in.device):
        rc = lib().excenv_update_ref_to(
            ctypes.c_int(env_id), ctypes.c_int(dtype_id(dtype)), ctypes.c_int64(B), ctypes.byref(props), ctypes.c_int32(nc),
            (ctypes.c_int32 * nc)(*control_idx) if nc else None, _ptrs(reference_in) if nc else None,
            ctypes.c_void_p(keys_in.data_ptr()), ctypes.c_void_p(hold_in.data_ptr()), _ptrs(reference_out) if nc else None,
            ctypes.c_void_p(keys_out.data_ptr()), ctypes.c_void_p(hold_out.data_ptr()), ctypes.c_int32(hold_min),
            ctypes.c_int32(hold_max), ctypes.c_void_p(_raw_stream(keys_in.device)))
    _check(rc, "excenv_update_ref_to")


def random_state(env_id, dtype, B, props: Props, keys: torch.Tensor, state_out: Sequence[torch.Tensor], key_leaf: torch.Tensor):
    """excenv_random_state: init_state(key) for every environment in one launch (keys / key_leaf: int64 [B, 2])."""
    _require_device(keys, "vmap_init_state")
    assert keys.dtype == torch.int64 and keys.is_contiguous() and key_leaf.dtype == torch.int64 and key_leaf.is_contiguous()
    with _on_device(keys.device):
        rc = lib().excenv_random_state(
            ctypes.c_int(env_id), ctypes.c_int(dtype_id(dtype)), ctypes.c_int64(B), ctypes.byref(props),
            ctypes.c_void_p(keys.data_ptr()), _ptrs(state_out), ctypes.c_void_p(key_leaf.data_ptr()),
            ctypes.c_void_p(_raw_stream(keys.device)))
    _check(rc, "excenv_random_state")


def sim_ahead_workspace_bytes(env_id, dtype, B, K, substeps, n_control, action_layout, traj_layout,
                              with_state_traj=True) -> int:
    return int(lib().excenv_sim_ahead_workspace_bytes(
        ctypes.c_int(env_id), ctypes.c_int(dtype_id(dtype)), ctypes.c_int64(B), ctypes.c_int64(K),
        ctypes.c_int32(substeps), ctypes.c_int32(n_control), ctypes.c_int(action_layout), ctypes.c_int(traj_layout),
        ctypes.c_int(int(with_state_traj))))


def transpose(x: torch.Tensor) -> torch.Tensor:
    """out[n][m] = in[m][n] for a contiguous 2-D device tensor (the library's LDS-tiled conversion kernel)."""
    _require_device(x, "transpose")
    assert x.ndim == 2 and x.is_contiguous()
    out = torch.empty((x.shape[1], x.shape[0]), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream(x.device).cuda_stream
        rc = lib().excenv_transpose(ctypes.c_int(dtype_id(x.dtype)), ctypes.c_int64(x.shape[0]), ctypes.c_int64(x.shape[1]),
                                    ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream))
    _check(rc, "excenv_transpose")
    return out


def probe_math(which: int, x: torch.Tensor) -> torch.Tensor:
    _require_device(x, "probe_math")
    x = x.contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream(x.device).cuda_stream
        rc = lib().excenv_probe_math(ctypes.c_int(which), ctypes.c_int(dtype_id(x.dtype)), ctypes.c_int64(x.numel()),
                                     ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                     ctypes.c_void_p(stream))
    _check(rc, "excenv_probe_math")
    return out


def probe_div(num: torch.Tensor, den: torch.Tensor):
    """(InvDiv(den).div(num), num / den) element-wise on the device (tests: equal bits)."""
    _require_device(num, "probe_div")
    num, den = num.contiguous(), den.contiguous()
    assert num.shape == den.shape and num.dtype == den.dtype
    fast, ref = torch.empty_like(num), torch.empty_like(num)
    with torch.cuda.device(num.device):
        stream = torch.cuda.current_stream(num.device).cuda_stream
        rc = lib().excenv_probe_div(ctypes.c_int(dtype_id(num.dtype)), ctypes.c_int64(num.numel()), ctypes.c_void_p(num.data_ptr()),
                                    ctypes.c_void_p(den.data_ptr()), ctypes.c_void_p(fast.data_ptr()),
                                    ctypes.c_void_p(ref.data_ptr()), ctypes.c_void_p(stream))
    _check(rc, "excenv_probe_div")
    return fast, ref
