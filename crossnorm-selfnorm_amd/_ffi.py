"""ctypes binding of libcnsn_hip.so (the C ABI of include/cnsn_hip.h).  No torch types cross it:
device pointers travel as integers, the stream as a void*.  The library must be present — there is
no fallback: `lib()` raises if it is missing (build it with `python __graft_entry__.py`)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CNSN_LIB_PATH") or os.path.join(_HERE, "libcnsn_hip.so")   # (override: tuning builds)

CNSN_F32, CNSN_BF16, CNSN_F16 = 0, 1, 2
STRATEGY_AUTO, STRATEGY_TWO_PASS, STRATEGY_RESIDENT, STRATEGY_LOCAL, STRATEGY_MONO = 0, 1, 2, 3, 4
ADD_NONE, ADD_PRE, ADD_POST = 0, 1, 2
LAYOUT_NCHW, LAYOUT_NHWC = 0, 1
PATHS = {0: "streaming", 1: "packed", 2: "resident", 3: "local", 4: "mono"}
ABI_VERSION = 8
PERM_INLINE_MAX = 1024       # CNSN_PERM_INLINE_MAX
E_UNSUPPORTED = -9


class Problem(C.Structure):
    """cnsn_problem_t"""
    _fields_ = [
        ("struct_bytes", C.c_int32), ("dtype", C.c_int32),
        ("N", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("cn_active", C.c_int32), ("content_box", C.c_int32 * 4), ("style_box", C.c_int32 * 4),
        ("lam", C.c_float), ("eps_cn", C.c_float),
        ("sn_active", C.c_int32), ("sn_two", C.c_int32), ("sn_training", C.c_int32),
        ("eps_sn", C.c_float), ("eps_bn", C.c_float), ("momentum", C.c_float),
        ("strategy", C.c_int32), ("layout", C.c_int32),
        ("context", C.c_void_p), ("context_bytes", C.c_uint64),
        ("perm_host", C.c_void_p),
    ]


class Gate(C.Structure):
    """cnsn_gate_t"""
    _fields_ = [("fc_weight", C.c_void_p), ("bn_weight", C.c_void_p), ("bn_bias", C.c_void_p),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p)]


class GateGrad(C.Structure):
    """cnsn_gate_grad_t"""
    _fields_ = [("d_fc_weight", C.c_void_p), ("d_bn_weight", C.c_void_p), ("d_bn_bias", C.c_void_p)]


class BnTail(C.Structure):
    """cnsn_bn_tail_t"""
    _fields_ = [("struct_bytes", C.c_int32), ("training", C.c_int32), ("eps", C.c_float), ("momentum", C.c_float),
                ("weight", C.c_void_p), ("bias", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("num_batches_tracked", C.c_void_p)]


class ArenaStats(C.Structure):
    """cnsn_arena_stats_t"""
    _fields_ = [("struct_bytes", C.c_int32), ("device", C.c_int32), ("chunk_bytes", C.c_uint64), ("mapped_bytes", C.c_uint64),
                ("in_use_bytes", C.c_uint64), ("blocks", C.c_uint64), ("blocks_in_use", C.c_uint64), ("hits", C.c_uint64),
                ("misses", C.c_uint64), ("failed", C.c_uint64), ("probed", C.c_uint64), ("tries", C.c_uint64),
                ("evicted", C.c_uint64), ("limit_bytes", C.c_uint64), ("broken", C.c_uint64)]


class Epilogue(C.Structure):
    """cnsn_epilogue_t"""
    _fields_ = [("struct_bytes", C.c_int32), ("add_mode", C.c_int32), ("relu", C.c_int32),
                ("reserved", C.c_int32), ("addend", C.c_void_p), ("sum_out", C.c_void_p)]


# name -> (restype, argtypes); every symbol include/cnsn_hip.h declares
SIGNATURES = {
    "cnsn_abi_version": (C.c_int, []),
    "cnsn_status_string": (C.c_char_p, [C.c_int]),
    "cnsn_context_bytes": (C.c_size_t, [C.POINTER(Problem)]),
    "cnsn_context_init": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "cnsn_resident_timeouts": (C.c_int, []),
    "cnsn_resident_enable": (None, [C.c_int]),
    "cnsn_resident_rearm": (C.c_int, []),
    "cnsn_resident_degraded": (C.c_int, []),
    "cnsn_reload_env": (None, []),
    "cnsn_set_wait_ms": (None, [C.c_int]),
    "cnsn_wait_ms": (C.c_int, []),
    "cnsn_set_headroom_cus": (None, [C.c_int]),
    "cnsn_headroom_cus": (C.c_int, []),
    "cnsn_arena_map": (C.c_void_p, [C.c_size_t, C.c_int, C.c_void_p]),
    "cnsn_arena_unmap": (None, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "cnsn_arena_alloc": (C.c_void_p, [C.c_int, C.c_size_t, C.c_void_p]),
    "cnsn_arena_record_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cnsn_arena_set_limit": (C.c_int, [C.c_int, C.c_uint64]),
    "cnsn_arena_free": (C.c_int, [C.c_void_p]),
    "cnsn_arena_trim": (C.c_size_t, [C.c_int]),
    "cnsn_arena_owns": (C.c_int, [C.c_void_p]),
    "cnsn_arena_stats": (C.c_int, [C.c_int, C.POINTER(ArenaStats)]),
    "cnsn_arena_set_chunk_bytes": (C.c_int, [C.c_size_t]),
    "cnsn_arena_prospect": (C.c_int, [C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
    "cnsn_arena_block_gbps": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "cnsn_arena_set_tries": (C.c_int, [C.c_int]),
    "cnsn_saved_floats": (C.c_size_t, [C.POINTER(Problem)]),
    "cnsn_workspace_bytes": (C.c_size_t, [C.POINTER(Problem)]),
    "cnsn_forward": (C.c_int, [C.POINTER(Problem), C.c_void_p, C.c_void_p, C.c_void_p,
                               C.POINTER(Gate), C.POINTER(Gate), C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_size_t, C.c_void_p]),
    "cnsn_backward": (C.c_int, [C.POINTER(Problem), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.POINTER(Gate), C.POINTER(Gate), C.c_void_p, C.c_void_p,
                                C.POINTER(GateGrad), C.POINTER(GateGrad), C.c_void_p, C.c_size_t,
                                C.c_void_p]),
    "cnsn_forward_fused": (C.c_int, [C.POINTER(Problem), C.POINTER(Epilogue), C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.POINTER(Gate), C.POINTER(Gate), C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cnsn_backward_fused": (C.c_int, [C.POINTER(Problem), C.POINTER(Epilogue), C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.POINTER(Gate), C.POINTER(Gate),
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(GateGrad),
                                      C.POINTER(GateGrad), C.c_void_p, C.c_size_t, C.c_void_p]),
    "cnsn_which_path": (C.c_int, [C.POINTER(Problem), C.POINTER(Epilogue), C.c_int, C.c_int]),
    "cnsn_keeps_sum": (C.c_int, [C.POINTER(Problem), C.POINTER(Epilogue)]),
    "cnsn_sn_cluster_plan": (C.c_int, [C.POINTER(Problem), C.POINTER(Epilogue), C.c_int]),
    "cnsn_bnrelu_plan": (C.c_int, [C.POINTER(Problem), C.POINTER(Epilogue), C.c_int]),
    "cnsn_bn_block_plan": (C.c_int, [C.POINTER(Problem), C.POINTER(Epilogue)]),
    "cnsn_forward_bn_block": (C.c_int, [C.POINTER(Problem), C.POINTER(Epilogue), C.POINTER(BnTail), C.POINTER(BnTail), C.c_void_p,
                                        C.POINTER(Gate), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cnsn_backward_bn_block": (C.c_int, [C.POINTER(Problem), C.POINTER(Epilogue), C.POINTER(BnTail), C.POINTER(BnTail), C.c_void_p,
                                         C.c_void_p, C.POINTER(Gate), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(GateGrad), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_void_p]),
    "cnsn_forward_bnrelu": (C.c_int, [C.POINTER(Problem), C.POINTER(Epilogue), C.POINTER(BnTail), C.c_void_p,
                                      C.POINTER(Gate), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_size_t, C.c_void_p]),
    "cnsn_backward_bnrelu": (C.c_int, [C.POINTER(Problem), C.POINTER(Epilogue), C.POINTER(BnTail), C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.POINTER(Gate), C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.POINTER(GateGrad), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cnsn_jsd_workspace_bytes": (C.c_size_t, [C.c_int]),
    "cnsn_jsd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cnsn_plane_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.POINTER(C.c_int32), C.c_float, C.c_void_p, C.c_void_p]),
    "cnsn_plane_stats_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(C.c_int32), C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cnsn_plane_affine": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cnsn_plane_dot": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_void_p, C.c_void_p]),
    "cnsn_plane_dot_shifted": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cnsn_plane_combine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


class CnsnError(RuntimeError):
    pass


def lib():
    """Load (once) and return the bound library; raise loudly when it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CnsnError(
                f"{LIB_PATH} is missing: the CrossNorm/SelfNorm HIP library has not been built. "
                "Run `python __graft_entry__.py` (hipcc --offload-arch=gfx950). There is no "
                "CPU or eager fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        got = handle.cnsn_abi_version()
        if got != ABI_VERSION:
            raise CnsnError(f"libcnsn_hip.so ABI {got} != binding ABI {ABI_VERSION}")
        _lib = handle
    return _lib


GLUE_DIR = os.path.join(_HERE, "_glue")
GLUE_PATH = os.path.join(GLUE_DIR, "cnsn_glue.so")
_glue = None
_glue_tried = False


def glue():
    """The C++ autograd glue (csrc/glue/cnsn_glue.cpp) if it has been built, else None.

    It calls the same C ABI as the ctypes path; it only removes ~0.1 ms of Python per call.  Its
    undefined `cnsn_*` symbols are resolved against libcnsn_hip.so, which is therefore loaded first with
    RTLD_GLOBAL.  `CNSN_NO_GLUE=1` forces the ctypes path (tests exercise both)."""
    global _glue, _glue_tried
    if _glue_tried:
        return _glue
    _glue_tried = True
    if os.environ.get("CNSN_NO_GLUE") == "1" or not os.path.exists(GLUE_PATH):
        return None
    lib()
    C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    import importlib.machinery
    import importlib.util
    import torch  # noqa: F401  (the extension links against libtorch)
    loader = importlib.machinery.ExtensionFileLoader("cnsn_glue", GLUE_PATH)
    spec = importlib.util.spec_from_loader("cnsn_glue", loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    if mod.abi_version() != ABI_VERSION:
        raise CnsnError(f"cnsn_glue.so was built against ABI {mod.abi_version()}, binding expects {ABI_VERSION}")
    _glue = mod
    return _glue


_timeouts_reported = 0
_defer_depth = 0          # > 0: check_resident_health() does not raise (a step guard polls at the step boundary instead)


def _timeout_count() -> int:
    """Cluster launches that gave up so far in this process (a plain host read of the library's pinned word)."""
    return lib().cnsn_resident_timeouts()


_TIMEOUT_TEXT = ("{what}: an earlier cluster-resident launch timed out waiting for part of its grid ({n} so far) — the GPU "
                 "is shared with work that kept it off the device for seconds.  The outputs of the step that was in flight "
                 "are invalid: repeat it.  The library now uses the two-pass kernels (CNSN_RESIDENT=0 selects them from "
                 "the start).")


def check_resident_health(what: str):
    """Poll the library's time-out counter (a plain host read).  A cluster-resident launch that gave up leaves its
    outputs incomplete; the library has already stopped choosing that strategy (two-pass from now on) — raise ONCE
    per event so the training loop can repeat the step instead of consuming garbage.  Inside `deferred_timeouts()`
    nothing is raised here: the step guard that opened it polls at the step boundary (`poll_timeouts`), where every
    data-parallel rank can take the same decision."""
    global _timeouts_reported
    if _defer_depth:
        return
    n = _timeout_count()
    if n > _timeouts_reported:
        _timeouts_reported = n
        raise CnsnError(_TIMEOUT_TEXT.format(what=what, n=n))


class deferred_timeouts:
    """`with deferred_timeouts():` — the per-call poll of the module layer stays silent inside the block.  A CnsnError
    raised in the MIDDLE of a step takes one data-parallel rank out of the step while its peers go on to the gradient
    all-reduce: their collectives no longer pair up.  A step guard (`callers.steps.StepGuard`, `bench.py`) therefore
    lets the step run to its end — the launch that gave up has marked its outputs with NaNs, nothing is applied from
    them — and asks `poll_timeouts()` at the boundary, where all ranks agree (`data_parallel.agree_to_repeat`)."""

    def __enter__(self):
        global _defer_depth
        _defer_depth += 1
        return self

    def __exit__(self, *exc):
        global _defer_depth
        _defer_depth -= 1
        return False


def poll_timeouts() -> int:
    """Cluster launches that gave up since the last report (0 = healthy); marks them reported, never raises.  The
    caller must have waited for the work it asks about (the counter moves when the kernel runs, not when it is queued)."""
    global _timeouts_reported
    n = _timeout_count()
    new = max(0, n - _timeouts_reported)
    _timeouts_reported = max(n, _timeouts_reported)
    return new


def settle_step(device=None, raise_on_timeout: bool = True) -> int:
    """Call between `loss.backward()` and `optimizer.step()` where a step must never be applied from incomplete
    gradients: waits for the device's current stream, then polls the time-out counter.  A cluster launch that gave up
    has marked the planes it still owed with NaNs and bumped the counter by the time the stream is idle — so the
    CnsnError is raised BEFORE the optimizer consumes the gradients of that step, and the caller repeats the step (the
    library has switched to the two-pass kernels by then).  Costs one stream synchronisation per step.
    `raise_on_timeout=False`: returns the number of new time-outs instead (what `StepGuard` all-reduces over the ranks)."""
    import torch
    torch.cuda.current_stream(device).synchronize()
    if not raise_on_timeout:
        return poll_timeouts()
    check_resident_health("training step")
    return 0


_plan_caches = []       # dictionaries of remembered planning answers (functional.py registers its own): emptied when a knob moves


def forget_plans():
    """Planning answers the Python layer remembers (which kernel family takes a call, side-buffer sizes) depend on the
    CNSN_* knobs, the strategy setting and `set_resident`: whoever changes one of those calls this."""
    for c in _plan_caches:
        c.clear()


def reload_env():
    """The library reads its CNSN_* environment knobs once, when it is loaded (csrc/cnsn_env.h); call this after changing
    one through `os.environ` inside a running process (tests, A/B tools).  Not while another thread is in a launch."""
    lib().cnsn_reload_env()
    forget_plans()


_following = False


def follow_environ():
    """For tests and A/B tools that flip CNSN_* knobs through `os.environ` inside one process: from now on every
    assignment to / deletion of a CNSN_* variable is followed by `cnsn_reload_env()` (if the library is loaded).
    `os.environ` hands its changes to the module-level `os.putenv` / `os.unsetenv`; those are wrapped.  A training
    process never needs this: it sets its knobs before the import."""
    global _following
    if _following:
        return
    _following = True
    put, unset = os.putenv, os.unsetenv

    def reread(key):
        if os.fsdecode(key).startswith("CNSN_") and _lib is not None:
            _lib.cnsn_reload_env()
            forget_plans()

    def putenv(key, value):
        put(key, value)
        reread(key)

    def unsetenv(key):
        unset(key)
        reread(key)

    os.putenv, os.unsetenv = putenv, unsetenv


def default_headroom_cus() -> int:
    """compute units the persistent grids leave to RCCL under a process group: two per channel RCCL may open —
    `NCCL_MAX_NCHANNELS` when the job sets it, else 16 channels (what an 8-GPU all-reduce ring of this stack's RCCL uses per
    direction pair; `profiles/r05_exchange_hardening.md` measured the 32-CU case) —, never more than a quarter of the part.
    `CNSN_HEADROOM_CUS` in the environment overrides it (0: none)."""
    try:
        ch = int(os.environ.get("NCCL_MAX_NCHANNELS", "") or 16)
    except ValueError:
        ch = 16
    return max(0, min(2 * ch, 64))


def under_process_group_defaults() -> dict:
    """One process per GPU under an initialised torch.distributed group of more than one rank.  (1) A rank whose cluster wait
    runs out stalls its peers' collectives for as long as the bound — 2 s there instead of 5 s (`cnsn_set_wait_ms`; an explicit
    CNSN_WAIT_MS in the environment wins, also one set later and made known through `reload_env()`).  (2) RCCL's channel
    kernels HOLD compute units for the length of an all-reduce: a persistent grid sized for the whole part next to them runs
    1.5-1.7 x its quiet time (profiles/r05_exchange_hardening.md), so the grids leave `default_headroom_cus()` compute units
    free (`cnsn_set_headroom_cus`; CNSN_HEADROOM_CUS in the environment wins).  Called by `callers.steps.StepGuard` on its
    first step, by `data_parallel` and by bench.py; returns what is in force ({} outside a process group)."""
    try:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return {}
        handle = lib()
        if "CNSN_WAIT_MS" not in os.environ:
            handle.cnsn_set_wait_ms(2000)
        if "CNSN_HEADROOM_CUS" not in os.environ:
            handle.cnsn_set_headroom_cus(default_headroom_cus())
            forget_plans()
        return {"wait_ms": int(handle.cnsn_wait_ms()), "headroom_cus": int(handle.cnsn_headroom_cus())}
    except Exception:   # pragma: no cover
        return {}


def check(status: int, what: str):
    if status != 0:
        msg = lib().cnsn_status_string(status).decode()
        if status == -7:   # CNSN_E_BATCH: same exception type nn.BatchNorm1d raises
            raise ValueError(f"{what}: {msg}")
        raise CnsnError(f"{what} failed with status {status}: {msg}")


def box4(b):
    arr = (C.c_int32 * 4)(-1, -1, -1, -1)
    if b is not None:
        for i in range(4):
            arr[i] = int(b[i])
    return arr
