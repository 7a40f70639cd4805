"""torch.autograd glue over the C ABI (torch here is plumbing: device memory, streams, autograd
book-keeping).  Every forward/backward below is ONE call into libcnsn_hip.so; nothing is computed
with torch ops on the activation tensor.  HIP device tensors only — other inputs raise."""
from __future__ import annotations

import ctypes as C
import dataclasses
import os
import threading
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import torch

from . import _ffi

Box = Tuple[int, int, int, int]

_DTYPES = {torch.float32: _ffi.CNSN_F32, torch.bfloat16: _ffi.CNSN_BF16, torch.float16: _ffi.CNSN_F16}

# process-wide default for cnsn_problem_t.strategy (tests/bench switch it to force one path)
_strategy = _ffi.STRATEGY_AUTO


def set_strategy(name: str):
    """'auto' | 'two_pass' | 'resident' | 'local' | 'mono' — which kernel strategy libcnsn_hip.so uses."""
    global _strategy
    _strategy = {"auto": _ffi.STRATEGY_AUTO, "two_pass": _ffi.STRATEGY_TWO_PASS,
                 "resident": _ffi.STRATEGY_RESIDENT, "local": _ffi.STRATEGY_LOCAL, "mono": _ffi.STRATEGY_MONO}[name]
    _ffi.forget_plans()


def set_resident(enabled: bool):
    """Allow (default) or forbid CNSN_STRATEGY_AUTO to choose the cluster-resident kernels (cnsn_resident_enable);
    the environment variable CNSN_RESIDENT=0 does the same from process start."""
    global _resident_set
    _ffi.lib().cnsn_resident_enable(int(bool(enabled)))
    _resident_set = bool(enabled)
    _ffi.forget_plans()


_resident_set = None        # what set_resident() last asked for (None: never called — the environment decides)


def resident_allowed() -> bool:
    """whether CNSN_STRATEGY_AUTO may currently choose the cluster kernels as far as the SWITCH goes (`set_resident`, else
    CNSN_RESIDENT at load); a degradation after a time-out is separate (`cnsn_resident_degraded`)"""
    import os
    return _resident_set if _resident_set is not None else os.environ.get("CNSN_RESIDENT", "1") != "0"


def _require_device(x: torch.Tensor, what: str):
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{what}: expected a tensor")
    if not x.is_cuda:
        raise _ffi.CnsnError(
            f"{what}: got a {x.device.type} tensor. This implementation runs on MI355X HIP device "
            "tensors only; there is no CPU path.")
    if x.dtype not in _DTYPES:
        raise TypeError(f"{what}: dtype {x.dtype} not supported (float32, bfloat16, float16)")
    assert x.dim() == 4, "expected an (N, C, H, W) tensor"          # reference cnsn.py:12


def _dense(t: torch.Tensor) -> torch.Tensor:
    """Contiguous AND 16-byte aligned: `.contiguous()` keeps a contiguous view with a storage offset (x[1:], a
    torch.split chunk) as it is, and the kernels' 16-byte vector accesses need an aligned base — such a view is
    copied once (the reference simply works on it, models/cnsn.py:14)."""
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone(memory_format=torch.contiguous_format)
    return t


NHWC = None


def _nhwc_call(x: torch.Tensor, cfg, chan_perm) -> bool:
    """True when this call is computed where a channels-last tensor lies (`cnsn_problem_t.layout = CNSN_LAYOUT_NHWC`): strictly
    channels-last strides, no crop boxes, no channel permutation, a channel count that is a whole number of 16-byte vectors.
    Everything else is copied to NCHW, like the reference's `.contiguous()` (models/cnsn.py:14).  CNSN_NHWC=0 switches it off."""
    global NHWC
    if NHWC is None:
        import os
        NHWC = os.environ.get("CNSN_NHWC") != "0"
    if not NHWC or x.dim() != 4 or chan_perm is not None or cfg.content_box is not None or cfg.style_box is not None:
        return False
    if x.is_contiguous() or not x.is_contiguous(memory_format=torch.channels_last):
        return False
    vec = 16 // x.element_size()
    return x.shape[1] % vec == 0 and x.shape[2] * x.shape[3] >= 2 and x.data_ptr() % 16 == 0


def _dense_cl(t: torch.Tensor) -> torch.Tensor:
    t = t.contiguous(memory_format=torch.channels_last)
    if t.data_ptr() % 16:
        t = t.clone(memory_format=torch.channels_last)
    return t


def _on_device_of(fn):
    """Run a Function.forward/backward with the tensor's device current: the library sizes grids, orders its
    persistent launches and sets kernel attributes for the CURRENT device, and the stream handed over belongs
    to the tensor's device — a model on cuda:1 must not depend on the caller having called set_device(1)."""
    import functools

    @functools.wraps(fn)
    def wrapped(ctx, first, *rest):
        if not (isinstance(first, torch.Tensor) and first.is_cuda):
            return fn(ctx, first, *rest)          # (raises the "device tensors only" error itself)
        with torch.cuda.device(first.device):
            return fn(ctx, first, *rest)
    return wrapped


def _out_like(x: torch.Tensor) -> torch.Tensor:
    """a fresh tensor for an output of the op, shaped like the dense `x`: from the output arena (arena.py) from its
    threshold on, else from torch's allocator"""
    g = _ffi.glue()
    return g.out_like(x) if g is not None else torch.empty_like(x)


def _stream(x):
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(x.device.index))


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype == torch.float32 and t.is_contiguous():
        return t.detach() if t.requires_grad else t
    return t.detach().to(torch.float32).contiguous()


# The module-level caches below are shared by every thread that calls into the library — torch.nn.DataParallel, the
# reference's own multi-GPU mode (cifar.py:395, imagenet.py:533), runs one Python thread per replica.  Dictionary reads
# and writes are atomic under the GIL; the read-modify-write sequences (growing the exchange context, taking a slot of
# the pinned staging ring) hold this lock (tests/test_gpu_threads.py).
_lock = threading.RLock()

# sizes of the caller-owned side buffers per problem signature (two ctypes calls saved per launch)
_size_cache = {}
_ffi._plan_caches.append(_size_cache)


def _sizes(prob):
    key = (prob.dtype, prob.N, prob.C, prob.H, prob.W, prob.cn_active, prob.sn_active, prob.sn_two,
           prob.sn_training, prob.content_box[0] >= 0, prob.style_box[0] >= 0, prob.strategy, prob.layout)
    hit = _size_cache.get(key)
    if hit is None:
        lib = _ffi.lib()
        hit = (lib.cnsn_saved_floats(C.byref(prob)), lib.cnsn_workspace_bytes(C.byref(prob)),
               lib.cnsn_context_bytes(C.byref(prob)))
        if hit[1] > 0:          # 0 = the library rejected the problem: let the launch report why
            _size_cache[key] = hit
    return hit


# The persistent exchange context of the cluster-resident kernels (cnsn_context_init): ONE buffer per device, grown
# when a larger problem shows up, allocated and initialised on first use.  With it a resident launch needs no fill
# launch in front of it.  None while the stream is being captured into a graph (a replay would repeat the launch number).
_contexts = {}
_retired_contexts = []     # outgrown buffers are kept: launches still queued on OTHER streams may exchange through them


def _context(prob, dev: torch.device):
    need = _sizes(prob)[2]
    if need == 0 or torch.cuda.is_current_stream_capturing():
        return
    have = _contexts.get(dev.index)
    if have is None or have.numel() < need:
        with _lock:
            have = _grow_context(need, dev)
    prob.context = have.data_ptr()
    prob.context_bytes = have.numel()


def _grow_context(need, dev):
    have = _contexts.get(dev.index)              # (again, under the lock: another thread may have grown it meanwhile)
    if have is None or have.numel() < need:
        size = max(need, 2 * have.numel() if have is not None else (4 << 20))
        buf = torch.empty(size, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev)
        _ffi.check(_ffi.lib().cnsn_context_init(C.c_void_p(buf.data_ptr()), size, C.c_void_p(stream.cuda_stream)),
                   "cnsn_context_init")
        stream.synchronize()            # once per buffer: ordered before whichever stream uses it next
        if have is not None:
            _retired_contexts.append(have)
        _contexts[dev.index] = have = buf
    return have


class _PinnedRing:
    """Host->device copies of the (tiny) permutations without stalling the launch thread.

    The reference does `torch.randperm(N).to(x.device)` (cnsn.py:62): a pageable-memory copy that
    blocks the host until everything queued on the stream has run.  Here the index vector goes
    through a small ring of pinned staging buffers and an asynchronous copy; a slot is reused only
    after the event recorded behind its previous copy has completed."""
    SLOTS = 8

    def __init__(self):
        self.rings = {}

    def to_device(self, idx: torch.Tensor, dev: torch.device) -> torch.Tensor:
        if idx.is_cuda:
            return idx.to(device=dev, dtype=torch.int64).contiguous()
        n = idx.numel()
        with _lock:                                   # (one slot per caller: replicas run as threads under DataParallel)
            ring = self.rings.setdefault((dev.index, n), {"next": 0, "slots": [None] * self.SLOTS})
            k = ring["next"] % self.SLOTS
            ring["next"] += 1
        if ring["slots"][k] is None:
            ring["slots"][k] = (torch.empty(n, dtype=torch.int64).pin_memory(), torch.cuda.Event())
        else:
            ring["slots"][k][1].synchronize()
        buf, evt = ring["slots"][k]
        buf.copy_(idx.reshape(-1))
        out = buf.to(dev, non_blocking=True)
        evt.record(torch.cuda.current_stream(dev))
        return out


_h2d = _PinnedRing()


@dataclass
class GateParams:
    """One SelfNorm gate's tensors (reference cnsn.py:118-126): Conv1d weight (C,1,2), BatchNorm1d
    weight / bias / running_mean / running_var (C)."""
    fc_weight: torch.Tensor
    bn_weight: torch.Tensor
    bn_bias: torch.Tensor
    running_mean: torch.Tensor
    running_var: torch.Tensor
    # nn.BatchNorm1d.num_batches_tracked (int64 scalar, on the device) when THIS call has to count — training mode with
    # running statistics and a fixed momentum —, else None: the forward kernel adds 1 (cnsn_gate_t.num_batches_tracked)
    num_batches_tracked: Optional[torch.Tensor] = None


@dataclass
class FusedConfig:
    cn_active: bool = False
    content_box: Optional[Box] = None
    style_box: Optional[Box] = None
    lam: Optional[float] = None
    sn_active: bool = False
    sn_two: bool = False
    sn_training: bool = True
    eps_cn: float = 1e-5
    eps_sn: float = 1e-12
    eps_bn: float = 1e-5
    momentum: float = 0.1
    # residual-block epilogue (cnsn_forward_fused): y = act(CNSN(x [+ addend]) [+ addend])
    add_mode: str = "none"          # 'none' | 'pre' | 'post'
    relu: bool = False

    @property
    def has_epilogue(self):
        return self.add_mode != "none" or self.relu


_ADD_MODES = {"none": _ffi.ADD_NONE, "pre": _ffi.ADD_PRE, "post": _ffi.ADD_POST}


def _epilogue(cfg: FusedConfig, addend, sum_out=None):
    e = _ffi.Epilogue()
    e.struct_bytes = C.sizeof(_ffi.Epilogue)
    e.add_mode = _ADD_MODES[cfg.add_mode]
    e.relu = int(cfg.relu)
    e.addend = addend.data_ptr() if addend is not None else None
    e.sum_out = sum_out.data_ptr() if sum_out is not None else None
    return e


# CNSN_KEEP_SUM=0: never keep x + addend for the backward (A/B knob; cnsn_epilogue_t.sum_out, ABI 8)
_KEEP_SUM = os.environ.get("CNSN_KEEP_SUM", "1") != "0"


def which_path(x: torch.Tensor, cfg: FusedConfig, backward: bool = False, chan_perm: bool = False) -> str:
    """'streaming' | 'packed' | 'resident' | 'local' | 'mono': the kernels a call with this tensor / configuration would run
    under the current strategy setting (cnsn_which_path; nothing is launched)."""
    prob = _problem(x, cfg)
    if _nhwc_call(x, cfg, True if chan_perm else None):
        prob.layout = _ffi.LAYOUT_NHWC
    epi = _epilogue(cfg, None) if cfg.has_epilogue else None
    st = _ffi.lib().cnsn_which_path(C.byref(prob), C.byref(epi) if epi else None, int(chan_perm), int(backward))
    if st < 0:
        _ffi.check(st, "cnsn_which_path")
    return _ffi.PATHS[st]


_perm_inline_cache = {}
_ffi._plan_caches.append(_perm_inline_cache)
# A/B switch (profiles/r04_launches_per_step.md): CNSN_LEGACY_LAUNCHES=1 restores the two small launches in front of the op —
# `num_batches_tracked.add_(1)` as a torch launch and the permutation as a host-to-device copy
import os as _os  # noqa: E402
LEGACY_LAUNCHES = _os.environ.get("CNSN_LEGACY_LAUNCHES") == "1"


def perm_inline_ok(x: torch.Tensor, cfg: FusedConfig, perm, chan_perm) -> bool:
    """True when the batch permutation of this call can travel as a launch argument (cnsn_problem_t.perm_host) instead of
    through a host-to-device copy: a host int64 vector of at most CNSN_PERM_INLINE_MAX entries, no channel permutation, and
    BOTH directions of the call resolve to the cluster-resident kernels (remembered per problem signature)."""
    if LEGACY_LAUNCHES or chan_perm is not None or not isinstance(perm, torch.Tensor) or perm.is_cuda or perm.dtype != torch.int64 \
            or not perm.is_contiguous() or perm.numel() > _ffi.PERM_INLINE_MAX or perm.numel() != x.shape[0]:
        return False
    key = (tuple(x.shape), x.dtype, x.device.index, cfg.sn_active, cfg.sn_two, cfg.sn_training, cfg.content_box is not None,
           cfg.style_box is not None, cfg.add_mode, cfg.relu, _strategy)
    hit = _perm_inline_cache.get(key)
    if hit is None:
        hit = which_path(x, cfg, False) == "resident" and which_path(x, cfg, True) == "resident"
        _perm_inline_cache[key] = hit
    return hit


def sn_cluster(x: torch.Tensor, cfg: FusedConfig, backward: bool = False) -> bool:
    """True when the call runs the SelfNorm-only cluster kernels (cnsn_sn_cluster_plan)."""
    prob = _problem(x, cfg)
    epi = _epilogue(cfg, None) if cfg.has_epilogue else None
    st = _ffi.lib().cnsn_sn_cluster_plan(C.byref(prob), C.byref(epi) if epi else None, int(backward))
    if st < 0:
        _ffi.check(st, "cnsn_sn_cluster_plan")
    return st == 1


def _problem(x: torch.Tensor, cfg: FusedConfig) -> _ffi.Problem:
    p = _ffi.Problem()
    p.struct_bytes = C.sizeof(_ffi.Problem)
    p.dtype = _DTYPES[x.dtype]
    p.N, p.C, p.H, p.W = (int(s) for s in x.shape)
    p.cn_active = int(cfg.cn_active)
    p.content_box = _ffi.box4(cfg.content_box)
    p.style_box = _ffi.box4(cfg.style_box)
    p.lam = 0.0 if cfg.lam is None else float(cfg.lam)
    p.eps_cn = cfg.eps_cn
    p.sn_active = int(cfg.sn_active)
    p.sn_two = int(cfg.sn_two)
    p.sn_training = int(cfg.sn_training)
    p.eps_sn, p.eps_bn, p.momentum = cfg.eps_sn, cfg.eps_bn, cfg.momentum
    p.strategy = _strategy
    return p


class _GateBuffers:
    """float32 contiguous views/copies of a gate's tensors + the cnsn_gate_t that points at them."""

    def __init__(self, w, gamma, beta, rm, rv, nbt=None):
        self.src_rm, self.src_rv = rm, rv
        self.w, self.gamma, self.beta = _f32(w), _f32(gamma), _f32(beta)
        direct = rm.dtype == torch.float32 and rm.is_contiguous() and rv.dtype == torch.float32 \
            and rv.is_contiguous()
        self.rm = rm.detach() if direct else _f32(rm)
        self.rv = rv.detach() if direct else _f32(rv)
        self.direct = direct
        if nbt is not None and not (nbt.dtype == torch.int64 and nbt.is_cuda and nbt.numel() == 1):
            nbt.add_(1)          # (a counter the kernel cannot reach: counted here, as nn.BatchNorm1d.forward does)
            nbt = None
        self.nbt = nbt
        self.c = _ffi.Gate(_ptr(self.w), _ptr(self.gamma), _ptr(self.beta), _ptr(self.rm), _ptr(self.rv), _ptr(nbt))

    def write_back(self):
        if not self.direct:  # running buffers kept in another dtype: copy the update back
            self.src_rm.copy_(self.rm)
            self.src_rv.copy_(self.rv)


class FusedCNSN(torch.autograd.Function):
    """y = SelfNorm(CrossNorm(x)) with either half optional — cnsn_forward / cnsn_backward."""

    @staticmethod
    def forward(ctx, x, cfg: FusedConfig, perm, chan_perm, g_w, g_gamma, g_beta, g_rm, g_rv,
                f_w, f_gamma, f_beta, f_rm, f_rv, addend=None, g_nbt=None, f_nbt=None):
        _require_device(x, "cnsn_forward")
        with torch.cuda.device(x.device):
            return FusedCNSN._forward(ctx, x, cfg, perm, chan_perm, g_w, g_gamma, g_beta, g_rm, g_rv,
                                      f_w, f_gamma, f_beta, f_rm, f_rv, addend, g_nbt, f_nbt)

    @staticmethod
    def _forward(ctx, x, cfg, perm, chan_perm, g_w, g_gamma, g_beta, g_rm, g_rv,
                 f_w, f_gamma, f_beta, f_rm, f_rv, addend, g_nbt, f_nbt):
        lib = _ffi.lib()
        _ffi.check_resident_health("cnsn_forward")
        nhwc = _nhwc_call(x, cfg, chan_perm if cfg.cn_active else None)
        x = x if nhwc else _dense(x)                               # reference cnsn.py:14 (a channels-last call is computed where it lies)
        if cfg.add_mode != "none":
            _require_device(addend, "cnsn_forward(addend)")
            assert addend.shape == x.shape and addend.dtype == x.dtype, "addend must match x"
            addend = _dense_cl(addend) if nhwc else _dense(addend)
        else:
            addend = None
        prob = _problem(x, cfg)
        prob.layout = _ffi.LAYOUT_NHWC if nhwc else _ffi.LAYOUT_NCHW
        dev = x.device
        _context(prob, dev)
        perm_host = None
        if cfg.cn_active:
            if not nhwc and perm_inline_ok(x, cfg, perm, chan_perm):
                # the permutation rides in the launch arguments: no upload.  A snapshot (<= 8 KB) when a backward will read
                # it again: the caller may reuse its index buffer in between (`torch.randperm(n, out=buf)`)
                perm_host, perm = (perm.clone() if any(ctx.needs_input_grad) else perm), None
                prob.perm_host = perm_host.data_ptr()
            else:
                perm = _h2d.to_device(perm, dev)
                if chan_perm is not None:
                    chan_perm = _h2d.to_device(chan_perm, dev)
        gate_g = _GateBuffers(g_w, g_gamma, g_beta, g_rm, g_rv, g_nbt if cfg.sn_training else None) if cfg.sn_active else None
        gate_f = _GateBuffers(f_w, f_gamma, f_beta, f_rm, f_rv, f_nbt if cfg.sn_training else None) \
            if (cfg.sn_active and cfg.sn_two) else None
        y = _out_like(x)
        need_bwd = any(ctx.needs_input_grad)
        saved_floats, ws_bytes = _sizes(prob)[:2]
        saved = torch.empty(saved_floats, dtype=torch.float32, device=dev) if need_bwd else None
        ws = torch.empty(ws_bytes // 4 + 4, dtype=torch.float32, device=dev)
        epi = _epilogue(cfg, addend) if cfg.has_epilogue else None
        # A PRE add in front of a channels-last call (two tensor passes each way): the forward KEEPS X = x + addend and the
        # backward reads that one tensor instead of two, twice (cnsn_epilogue_t.sum_out).  X is what the reference's in-place
        # `out += identity` leaves (resnet_cnsn.py:117) and what its autograd saves; x and the addend are not saved here.
        xsum = None
        if need_bwd and nhwc and cfg.add_mode == "pre" and _KEEP_SUM and lib.cnsn_keeps_sum(C.byref(prob), C.byref(epi)) == 1:
            xsum = _out_like(x)
            epi.sum_out = xsum.data_ptr()

        def launch():
            return lib.cnsn_forward_fused(C.byref(prob), C.byref(epi) if epi else None, _ptr(x),
                                          _ptr(perm if cfg.cn_active else None),
                                          _ptr(chan_perm if cfg.cn_active else None),
                                          C.byref(gate_g.c) if gate_g else None, C.byref(gate_f.c) if gate_f else None,
                                          _ptr(y), _ptr(saved), _ptr(ws), ws_bytes, _stream(x))

        st = launch()
        if st == _ffi.E_UNSUPPORTED and perm_host is not None:   # (the plan changed under us: upload and call again)
            perm, perm_host = _h2d.to_device(perm_host, dev), None
            prob.perm_host = None
            st = launch()
        _ffi.check(st, "cnsn_forward")
        if cfg.sn_active and cfg.sn_training:
            gate_g.write_back()
            if gate_f:
                gate_f.write_back()
        if need_bwd:
            ctx.cfg, ctx.prob = cfg, prob
            ctx.gates = (gate_g, gate_f)
            ctx.param_dtypes = tuple(t.dtype if t is not None else None
                                     for t in (g_w, g_gamma, g_beta, f_w, f_gamma, f_beta))
            ctx.perm_host = perm_host                  # (a CPU tensor: kept alive for the backward's launch argument)
            ctx.kept_sum = xsum is not None
            if xsum is not None:                       # the backward is the backward of the op WITHOUT the add, on X
                ctx.cfg = dataclasses.replace(cfg, add_mode="none")
                ctx.save_for_backward(xsum, saved, perm if cfg.cn_active else None,
                                      chan_perm if cfg.cn_active else None, None)
            else:
                ctx.save_for_backward(x, saved, perm if cfg.cn_active else None,
                                      chan_perm if cfg.cn_active else None, addend)
        return y

    @staticmethod
    @_on_device_of
    def backward(ctx, gy):
        lib = _ffi.lib()
        x, saved, perm, chan_perm, addend = ctx.saved_tensors
        cfg, prob = ctx.cfg, ctx.prob
        gate_g, gate_f = ctx.gates
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        gy = _dense_cl(gy) if prob.layout == _ffi.LAYOUT_NHWC else _dense(gy)
        dev = x.device
        dx = _out_like(x)
        ws_bytes = _sizes(prob)[1]
        _context(prob, dev)     # (the buffer may have grown since the forward; None under graph capture)
        if torch.cuda.is_current_stream_capturing():
            prob.context, prob.context_bytes = None, 0
        ws = torch.empty(ws_bytes // 4 + 4, dtype=torch.float32, device=dev)
        Cn = x.shape[1]

        def grads():  # one allocation, three views: d_fc_weight (C,1,2), d_bn_weight (C), d_bn_bias (C)
            flat = torch.empty(4 * Cn, dtype=torch.float32, device=dev)
            dw, dgam, dbet = flat[:2 * Cn].view(Cn, 1, 2), flat[2 * Cn:3 * Cn], flat[3 * Cn:]
            return (dw, dgam, dbet), _ffi.GateGrad(_ptr(dw), _ptr(dgam), _ptr(dbet))

        gg = gf = None
        gg_c = gf_c = None
        if cfg.sn_active:
            gg, gg_c = grads()
            if cfg.sn_two:
                gf, gf_c = grads()
        epi = _epilogue(cfg, addend) if cfg.has_epilogue else None
        d_add = None
        if cfg.add_mode == "post":      # gradient of a POST addend: grad_y behind the ReLU mask
            d_add = _out_like(x) if cfg.relu else gy
        perm_host = ctx.perm_host
        prob.perm_host = perm_host.data_ptr() if perm_host is not None else None

        def launch(perm_dev):
            return lib.cnsn_backward_fused(C.byref(prob), C.byref(epi) if epi else None, _ptr(gy), _ptr(x), _ptr(perm_dev),
                                           _ptr(chan_perm), C.byref(gate_g.c) if gate_g else None,
                                           C.byref(gate_f.c) if gate_f else None, _ptr(saved), _ptr(dx),
                                           _ptr(d_add) if (cfg.add_mode == "post" and cfg.relu) else None,
                                           C.byref(gg_c) if gg_c else None, C.byref(gf_c) if gf_c else None,
                                           _ptr(ws), ws_bytes, _stream(x))

        st = launch(perm)
        if st == _ffi.E_UNSUPPORTED and perm_host is not None:   # (another strategy by now: it wants the device array)
            prob.perm_host = None
            st = launch(_h2d.to_device(perm_host, dev))
        _ffi.check(st, "cnsn_backward")
        if cfg.add_mode == "pre" or ctx.kept_sum:       # d(x + addend) reaches both terms unchanged
            d_add = dx
        pd = ctx.param_dtypes
        out_g = [None] * 3 if gg is None else [t if t.dtype == pd[i] else t.to(pd[i]) for i, t in enumerate(gg)]
        out_f = [None] * 3 if gf is None else [t if t.dtype == pd[3 + i] else t.to(pd[3 + i])
                                               for i, t in enumerate(gf)]
        #      x   cfg  perm  chan  g_w..g_beta   g_rm g_rv   f_w..f_beta  f_rm f_rv  addend  g_nbt f_nbt
        return (dx, None, None, None, *out_g, None, None, *out_f, None, None, d_add, None, None)


def bnrelu_plan(x: torch.Tensor, cfg: FusedConfig, backward: bool = False) -> bool:
    """True when a fused kernel evaluates `y = CNSN(x [+ addend]); z = relu(BatchNorm2d(y))` in one launch for this
    tensor / configuration (cnsn_bnrelu_plan)."""
    prob = _problem(x, cfg)
    epi = _epilogue(cfg, None) if cfg.has_epilogue else None
    st = _ffi.lib().cnsn_bnrelu_plan(C.byref(prob), C.byref(epi) if epi else None, int(backward))
    if st < 0:
        _ffi.check(st, "cnsn_bnrelu_plan")
    return st == 1


class FusedCNSNTail(torch.autograd.Function):
    """(y, z) = (SelfNorm(x [+ addend]), relu(BatchNorm2d(SelfNorm(x [+ addend])))) — cnsn_forward_bnrelu /
    cnsn_backward_bnrelu: the end of one WideResNet block and the start of the next (wideresnet_cnsn.py:93-96,
    :76-77) in one launch per direction.  `want_y` False: only z is produced (the next block's widths differ, :69-70)."""

    @staticmethod
    def forward(ctx, x, cfg: FusedConfig, addend, want_y, g_w, g_gamma, g_beta, g_rm, g_rv, bn_w, bn_b, bn_rm, bn_rv,
                bn_training, bn_eps, bn_momentum, g_nbt=None, bn_nbt=None):
        _require_device(x, "cnsn_forward_bnrelu")
        with torch.cuda.device(x.device):
            lib = _ffi.lib()
            _ffi.check_resident_health("cnsn_forward_bnrelu")
            x = _dense(x)
            if cfg.add_mode != "none":
                _require_device(addend, "cnsn_forward_bnrelu(addend)")
                assert addend.shape == x.shape and addend.dtype == x.dtype, "addend must match x"
                addend = _dense(addend)
            else:
                addend = None
            prob = _problem(x, cfg)
            dev = x.device
            gate = _GateBuffers(g_w, g_gamma, g_beta, g_rm, g_rv, g_nbt if cfg.sn_training else None)
            bw, bb = _f32(bn_w), _f32(bn_b)
            if bn_nbt is not None and not (bn_training and bn_nbt.dtype == torch.int64 and bn_nbt.is_cuda):
                if bn_training:
                    bn_nbt.add_(1)
                bn_nbt = None
            direct = (bn_rm.dtype == torch.float32 and bn_rm.is_contiguous() and bn_rv.dtype == torch.float32
                      and bn_rv.is_contiguous())
            rm = bn_rm.detach() if direct else _f32(bn_rm)
            rv = bn_rv.detach() if direct else _f32(bn_rv)
            tail = _ffi.BnTail(C.sizeof(_ffi.BnTail), int(bn_training), float(bn_eps), float(bn_momentum), bw.data_ptr(),
                               bb.data_ptr(), rm.data_ptr(), rv.data_ptr(), _ptr(bn_nbt))
            y = _out_like(x) if want_y else None
            z = _out_like(x)
            need_bwd = any(ctx.needs_input_grad)
            saved_floats, ws_bytes = _sizes(prob)[:2]
            saved = torch.empty(saved_floats, dtype=torch.float32, device=dev) if need_bwd else None
            stats = torch.empty(2 * x.shape[1], dtype=torch.float32, device=dev)
            ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
            epi = _epilogue(cfg, addend) if cfg.has_epilogue else None
            st = lib.cnsn_forward_bnrelu(C.byref(prob), C.byref(epi) if epi else None, C.byref(tail), _ptr(x),
                                         C.byref(gate.c), _ptr(y), _ptr(z), _ptr(saved), _ptr(stats), _ptr(ws), ws_bytes,
                                         _stream(x))
            _ffi.check(st, "cnsn_forward_bnrelu")
            if cfg.sn_training:
                gate.write_back()
            if bn_training and not direct:
                bn_rm.copy_(rm)
                bn_rv.copy_(rv)
            ctx.set_materialize_grads(False)    # an unused y hands None to the backward, not a tensor of zeros
            if need_bwd:
                ctx.cfg, ctx.prob, ctx.gate, ctx.want_y = cfg, prob, gate, want_y
                ctx.tail_cfg = (bool(bn_training), float(bn_eps), float(bn_momentum))
                ctx.param_dtypes = (g_w.dtype, g_gamma.dtype, g_beta.dtype, bn_w.dtype, bn_b.dtype)
                ctx.bn_buffers = (bw, bb, rm, rv)
                ctx.save_for_backward(x, saved, addend, stats)
            return (y, z) if want_y else z

    @staticmethod
    def backward(ctx, *grads):
        gy, gz = grads if ctx.want_y else (None, grads[0])
        x, saved, addend, stats = ctx.saved_tensors
        with torch.cuda.device(x.device):
            lib = _ffi.lib()
            cfg, prob, gate = ctx.cfg, ctx.prob, ctx.gate
            dev = x.device
            if gz is None:
                gz = torch.zeros_like(x)
            gz = _dense(gz.to(x.dtype))
            if gy is not None:
                gy = _dense(gy.to(x.dtype))
            bw, bb, rm, rv = ctx.bn_buffers
            tr, eps, mom = ctx.tail_cfg
            tail = _ffi.BnTail(C.sizeof(_ffi.BnTail), int(tr), eps, mom, bw.data_ptr(), bb.data_ptr(), rm.data_ptr(),
                               rv.data_ptr(), None)
            Cn = x.shape[1]
            dx = _out_like(x)
            flat = torch.empty(6 * Cn, dtype=torch.float32, device=dev)
            dw, dgam, dbet = flat[:2 * Cn].view(Cn, 1, 2), flat[2 * Cn:3 * Cn], flat[3 * Cn:4 * Cn]
            dbw, dbb = flat[4 * Cn:5 * Cn], flat[5 * Cn:]
            gg = _ffi.GateGrad(_ptr(dw), _ptr(dgam), _ptr(dbet))
            ws_bytes = _sizes(prob)[1]
            ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
            epi = _epilogue(cfg, addend) if cfg.has_epilogue else None
            st = lib.cnsn_backward_bnrelu(C.byref(prob), C.byref(epi) if epi else None, C.byref(tail), _ptr(gy), _ptr(gz),
                                          _ptr(x), C.byref(gate.c), _ptr(saved), _ptr(stats), _ptr(dx), C.byref(gg),
                                          _ptr(dbw), _ptr(dbb), _ptr(ws), ws_bytes, _stream(x))
            _ffi.check(st, "cnsn_backward_bnrelu")
            pd = ctx.param_dtypes
            outs = [t if t.dtype == pd[i] else t.to(pd[i]) for i, t in enumerate((dw, dgam, dbet, dbw, dbb))]
            d_add = dx if cfg.add_mode == "pre" else None
            #       x   cfg   addend want_y  g_w      g_gamma  g_beta  g_rm  g_rv  bn_w     bn_b    bn_rm bn_rv  tr   eps  mom  nbt nbt
            return (dx, None, d_add, None, outs[0], outs[1], outs[2], None, None, outs[3], outs[4], None, None, None, None, None,
                    None, None)


# ------------------------------------------------------------------------------------------------
# the block's last BatchNorm2d in front of the op: y = act(SelfNorm(BatchNorm2d(conv_out) + identity))
# (cnsn_forward_bn_block / cnsn_backward_bn_block, csrc/cnsn_nhwc_bnhead_kernels.h)
# ------------------------------------------------------------------------------------------------
_bn_block_plan_cache = {}
_ffi._plan_caches.append(_bn_block_plan_cache)
# CNSN_BN_BLOCK=0: never fold the block's last BatchNorm2d into the op's launch (A/B knob)
_BN_BLOCK = os.environ.get("CNSN_BN_BLOCK", "1") != "0"


def bn_block_plan(x: torch.Tensor, cfg: FusedConfig) -> bool:
    """True when ONE launch per direction evaluates `act(CNSN(BatchNorm2d(conv_out) + identity))` for this tensor / configuration
    (cnsn_bn_block_plan: channels-last, SelfNorm alone with one gate in training mode, N <= 256, no stream capture) —
    remembered per (shape, dtype, configuration, strategy)."""
    if not _BN_BLOCK or not x.is_cuda or x.dim() != 4 or x.dtype not in _DTYPES or cfg.cn_active or not cfg.sn_training:
        return False
    if not x.is_contiguous(memory_format=torch.channels_last) or x.is_contiguous():
        return False
    if _ffi.lib().cnsn_resident_degraded():      # (a persistent launch gave up and nobody re-armed since: the un-fused sequence)
        return False
    key = (tuple(x.shape), x.dtype, x.device.index, cfg.relu, cfg.sn_two, _strategy)
    hit = _bn_block_plan_cache.get(key)
    if hit is None:
        prob = _problem(x, cfg)
        prob.layout = _ffi.LAYOUT_NHWC
        epi = _epilogue(cfg, None)
        hit = _ffi.lib().cnsn_bn_block_plan(C.byref(prob), C.byref(epi)) == 1
        _bn_block_plan_cache[key] = hit
    return hit


class _Bn2dBuffers:
    """float32 contiguous views / copies of an nn.BatchNorm2d's tensors + the cnsn_bn_tail_t that points at them (training mode)"""

    def __init__(self, w, b, rm, rv, eps, momentum, nbt):
        self.src_rm, self.src_rv = rm, rv
        self.w, self.b = _f32(w), _f32(b)
        if nbt is not None and not (nbt.dtype == torch.int64 and nbt.is_cuda and nbt.numel() == 1):
            nbt.add_(1)          # (a counter the kernel cannot reach: counted here, as nn.BatchNorm2d.forward does)
            nbt = None
        self.direct = rm.dtype == torch.float32 and rm.is_contiguous() and rv.dtype == torch.float32 and rv.is_contiguous()
        self.rm = rm.detach() if self.direct else _f32(rm)
        self.rv = rv.detach() if self.direct else _f32(rv)
        self.eps, self.momentum = float(eps), float(momentum)
        self.c = self.struct(nbt)

    def struct(self, nbt=None):
        return _ffi.BnTail(C.sizeof(_ffi.BnTail), 1, self.eps, self.momentum, self.w.data_ptr(), self.b.data_ptr(), self.rm.data_ptr(),
                           self.rv.data_ptr(), _ptr(nbt))

    def write_back(self):
        if not self.direct:
            self.src_rm.copy_(self.rm)
            self.src_rv.copy_(self.rv)


class FusedBnBlock(torch.autograd.Function):
    """y = act(SelfNorm(BatchNorm2d(conv_out) + identity)) — the tail of a ResNet bottleneck (resnet_cnsn.py:108-122, pos='post')
    in one launch per direction; with `bn2_*` the identity is itself `BatchNorm2d(skip convolution)` (the block's downsample,
    :99-100) and `identity` is that convolution's output.  Saves conv_out and identity (what the BatchNorm2d layers and the add
    would have saved), never writes a BatchNorm2d's output or the sum."""

    @staticmethod
    def forward(ctx, conv_out, identity, cfg: FusedConfig, g_w, g_gamma, g_beta, g_rm, g_rv, bn_w, bn_b, bn_rm, bn_rv, bn_eps,
                bn_momentum, g_nbt=None, bn_nbt=None, bn2_w=None, bn2_b=None, bn2_rm=None, bn2_rv=None, bn2_eps=0.0,
                bn2_momentum=0.0, bn2_nbt=None):
        _require_device(conv_out, "cnsn_forward_bn_block")
        with torch.cuda.device(conv_out.device):
            lib = _ffi.lib()
            _ffi.check_resident_health("cnsn_forward_bn_block")
            x = _dense_cl(conv_out)
            _require_device(identity, "cnsn_forward_bn_block(identity)")
            assert identity.shape == x.shape and identity.dtype == x.dtype, "identity must match conv_out"
            idt = _dense_cl(identity)
            prob = _problem(x, cfg)
            prob.layout = _ffi.LAYOUT_NHWC
            dev = x.device
            _context(prob, dev)
            gate = _GateBuffers(g_w, g_gamma, g_beta, g_rm, g_rv, g_nbt)
            head = _Bn2dBuffers(bn_w, bn_b, bn_rm, bn_rv, bn_eps, bn_momentum, bn_nbt)
            skip = _Bn2dBuffers(bn2_w, bn2_b, bn2_rm, bn2_rv, bn2_eps, bn2_momentum, bn2_nbt) if bn2_w is not None else None
            y = _out_like(x)
            need_bwd = any(ctx.needs_input_grad)
            saved_floats, ws_bytes = _sizes(prob)[:2]
            saved = torch.empty(saved_floats, dtype=torch.float32, device=dev) if need_bwd else None
            stats = torch.empty((8 if skip else 4) * x.shape[1], dtype=torch.float32, device=dev)
            ws = torch.empty(ws_bytes // 4 + 4, dtype=torch.float32, device=dev)
            epi = _epilogue(cfg, idt)
            st = lib.cnsn_forward_bn_block(C.byref(prob), C.byref(epi), C.byref(head.c), C.byref(skip.c) if skip else None, _ptr(x),
                                           C.byref(gate.c), _ptr(y), _ptr(saved), _ptr(stats), _ptr(ws), ws_bytes, _stream(x))
            _ffi.check(st, "cnsn_forward_bn_block")
            gate.write_back()
            head.write_back()
            if skip:
                skip.write_back()
            if need_bwd:
                ctx.cfg, ctx.prob, ctx.gate, ctx.head, ctx.skip = cfg, prob, gate, head, skip
                ctx.param_dtypes = (g_w.dtype, g_gamma.dtype, g_beta.dtype, bn_w.dtype, bn_b.dtype,
                                    bn2_w.dtype if skip else None, bn2_b.dtype if skip else None)
                ctx.save_for_backward(x, idt, saved, stats)
            return y

    @staticmethod
    def backward(ctx, gy):
        x, idt, saved, stats = ctx.saved_tensors
        with torch.cuda.device(x.device):
            lib = _ffi.lib()
            cfg, prob, gate, head, skip = ctx.cfg, ctx.prob, ctx.gate, ctx.head, ctx.skip
            dev = x.device
            gy = _dense_cl(gy if gy.dtype == x.dtype else gy.to(x.dtype))
            hs = head.struct()
            ss = skip.struct() if skip else None
            Cn = x.shape[1]
            d_conv, d_idt = _out_like(x), _out_like(x)
            flat = torch.empty(8 * Cn, dtype=torch.float32, device=dev)
            dw, dgam, dbet = flat[:2 * Cn].view(Cn, 1, 2), flat[2 * Cn:3 * Cn], flat[3 * Cn:4 * Cn]
            dbw, dbb, d2w, d2b = flat[4 * Cn:5 * Cn], flat[5 * Cn:6 * Cn], flat[6 * Cn:7 * Cn], flat[7 * Cn:]
            gg = _ffi.GateGrad(_ptr(dw), _ptr(dgam), _ptr(dbet))
            ws_bytes = _sizes(prob)[1]
            _context(prob, dev)
            if torch.cuda.is_current_stream_capturing():
                prob.context, prob.context_bytes = None, 0
            ws = torch.empty(ws_bytes // 4 + 4, dtype=torch.float32, device=dev)
            epi = _epilogue(cfg, idt)
            st = lib.cnsn_backward_bn_block(C.byref(prob), C.byref(epi), C.byref(hs), C.byref(ss) if ss else None, _ptr(gy), _ptr(x),
                                            C.byref(gate.c), _ptr(saved), _ptr(stats), _ptr(d_conv), _ptr(d_idt), C.byref(gg),
                                            _ptr(dbw), _ptr(dbb), _ptr(d2w) if skip else None, _ptr(d2b) if skip else None, _ptr(ws),
                                            ws_bytes, _stream(x))
            _ffi.check(st, "cnsn_backward_bn_block")
            pd = ctx.param_dtypes
            outs = [t if t.dtype == pd[i] else t.to(pd[i]) for i, t in enumerate((dw, dgam, dbet, dbw, dbb))]
            o2 = [None, None] if not skip else [t if t.dtype == pd[5 + i] else t.to(pd[5 + i]) for i, t in enumerate((d2w, d2b))]
            #       conv   identity cfg  g_w      g_gamma  g_beta  g_rm  g_rv  bn_w     bn_b    bn_rm bn_rv eps   mom   nbt   nbt
            return (d_conv, d_idt, None, outs[0], outs[1], outs[2], None, None, outs[3], outs[4], None, None, None, None, None, None,
                    o2[0], o2[1], None, None, None, None, None)


def _glue_cfg(cfg: FusedConfig, need_bwd: bool):
    cb = cfg.content_box if cfg.content_box is not None else (-1, -1, -1, -1)
    sb = cfg.style_box if cfg.style_box is not None else (-1, -1, -1, -1)
    icfg = [int(cfg.cn_active), *(int(v) for v in cb), *(int(v) for v in sb), int(cfg.sn_active), int(cfg.sn_two),
            int(cfg.sn_training), _strategy, int(need_bwd), _ADD_MODES[cfg.add_mode], int(cfg.relu), 0]
    fcfg = [0.0 if cfg.lam is None else float(cfg.lam), cfg.eps_cn, cfg.eps_sn, cfg.eps_bn, cfg.momentum]
    return icfg, fcfg


_tail_plan_cache = {}
_ffi._plan_caches.append(_tail_plan_cache)


def fused_cnsn_tail(x, cfg: FusedConfig, addend, want_y: bool, g: GateParams, bn_w, bn_b, bn_rm, bn_rv, bn_training: bool,
                    bn_eps: float, bn_momentum: float, bn_nbt=None):
    """(y or None, z): FusedCNSNTail through the C++ glue when it is built (same C ABI calls, without the Python
    per-call overhead — WideResNet's step is launch-bound), else through ctypes."""
    glue = _ffi.glue()
    if glue is None:
        out = FusedCNSNTail.apply(x, cfg, addend, bool(want_y), g.fc_weight, g.bn_weight, g.bn_bias, g.running_mean,
                                  g.running_var, bn_w, bn_b, bn_rm, bn_rv, bn_training, bn_eps, bn_momentum,
                                  g.num_batches_tracked, bn_nbt)
        return out if want_y else (None, out)
    _require_device(x, "cnsn_forward_bnrelu")
    _ffi.check_resident_health("cnsn_forward_bnrelu")
    need_bwd = torch.is_grad_enabled() and (x.requires_grad or any(
        t is not None and t.requires_grad for t in (g.fc_weight, g.bn_weight, g.bn_bias, bn_w, bn_b, addend)))
    icfg, fcfg = _glue_cfg(cfg, need_bwd)
    out = glue.fused_cnsn_tail(x, icfg, fcfg, g.fc_weight, g.bn_weight, g.bn_bias, g.running_mean, g.running_var,
                               addend if cfg.add_mode != "none" else None, bn_w, bn_b, bn_rm, bn_rv,
                               [int(want_y), int(bn_training)], [float(bn_eps), float(bn_momentum)],
                               g.num_batches_tracked if cfg.sn_training else None, bn_nbt if bn_training else None)
    return (out[0], out[1]) if want_y else (None, out[0])


def bnrelu_plan_cached(x: torch.Tensor, cfg: FusedConfig, need_bwd: bool) -> bool:
    """bnrelu_plan for both directions, remembered per (shape, dtype, configuration, strategy): a per-call ctypes
    round trip is what the fused tail is there to save"""
    key = (tuple(x.shape), x.dtype, x.device.index, cfg.add_mode, cfg.sn_training, cfg.sn_two, _strategy, need_bwd)
    hit = _tail_plan_cache.get(key)
    if hit is None:
        hit = bnrelu_plan(x, cfg) and (not need_bwd or bnrelu_plan(x, cfg, backward=True))
        _tail_plan_cache[key] = hit
    return hit


def fused_cnsn(x, cfg: FusedConfig, perm=None, chan_perm=None, g: Optional[GateParams] = None,
               f: Optional[GateParams] = None, addend: Optional[torch.Tensor] = None):
    ga = (g.fc_weight, g.bn_weight, g.bn_bias, g.running_mean, g.running_var) if g else (None,) * 5
    fa = (f.fc_weight, f.bn_weight, f.bn_bias, f.running_mean, f.running_var) if f else (None,) * 5
    g_nbt = g.num_batches_tracked if (g and cfg.sn_training) else None
    f_nbt = f.num_batches_tracked if (f and cfg.sn_training) else None
    glue = _ffi.glue()
    if glue is None:
        return FusedCNSN.apply(x, cfg, perm, chan_perm, *ga, *fa, addend, g_nbt, f_nbt)
    # C++ glue: same C ABI calls, without the Python per-call overhead
    _require_device(x, "cnsn_forward")
    _ffi.check_resident_health("cnsn_forward")
    if cfg.add_mode == "none":
        addend = None
    need_bwd = torch.is_grad_enabled() and (x.requires_grad or any(
        t is not None and t.requires_grad for t in (*ga[:3], *fa[:3], addend)))
    cb = cfg.content_box if cfg.content_box is not None else (-1, -1, -1, -1)
    sb = cfg.style_box if cfg.style_box is not None else (-1, -1, -1, -1)
    inline = cfg.cn_active and perm_inline_ok(x, cfg, perm, chan_perm)
    icfg = [int(cfg.cn_active), *(int(v) for v in cb), *(int(v) for v in sb), int(cfg.sn_active), int(cfg.sn_two),
            int(cfg.sn_training), _strategy, int(need_bwd), _ADD_MODES[cfg.add_mode], int(cfg.relu), int(inline)]
    fcfg = [0.0 if cfg.lam is None else float(cfg.lam), cfg.eps_cn, cfg.eps_sn, cfg.eps_bn, cfg.momentum]
    return glue.fused_cnsn(x, icfg, fcfg, perm if cfg.cn_active else None, chan_perm if cfg.cn_active else None,
                           *ga, *fa, addend, g_nbt, f_nbt)


# ------------------------------------------------------------------------------------------------
# building blocks exposed for the stand-alone functions of the reference (cnsn.py:8-29)
# ------------------------------------------------------------------------------------------------
def _dims(x):
    return tuple(int(s) for s in x.shape)


class PlaneStats(torch.autograd.Function):
    """(mean, std) of every plane — cnsn_plane_stats / cnsn_plane_stats_backward."""

    @staticmethod
    @_on_device_of
    def forward(ctx, x, eps, box, keep_fp32=False):
        _require_device(x, "calc_ins_mean_std")
        lib = _ffi.lib()
        x = _dense(x)
        n, c, h, w = _dims(x)
        ms = torch.empty(2, n * c, dtype=torch.float32, device=x.device)
        st = lib.cnsn_plane_stats(_ptr(x), _DTYPES[x.dtype], n, c, h, w, _ffi.box4(box) if box else None,
                                  float(eps), _ptr(ms), _stream(x))
        _ffi.check(st, "cnsn_plane_stats")
        ctx.box = box
        ctx.save_for_backward(x, ms)
        out_dtype = torch.float32 if keep_fp32 else x.dtype
        mean = ms[0].view(n, c, 1, 1).to(out_dtype)
        std = ms[1].view(n, c, 1, 1).to(out_dtype)
        return mean, std

    @staticmethod
    @_on_device_of
    def backward(ctx, gmean, gstd):
        lib = _ffi.lib()
        x, ms = ctx.saved_tensors
        n, c, h, w = _dims(x)
        gm = _f32(gmean).view(-1)
        gs = _f32(gstd).view(-1)
        dx = _out_like(x)
        st = lib.cnsn_plane_stats_backward(_ptr(x), _DTYPES[x.dtype], n, c, h, w,
                                           _ffi.box4(ctx.box) if ctx.box else None, _ptr(ms[0]), _ptr(ms[1]),
                                           _ptr(gm), _ptr(gs), _ptr(dx), _stream(x))
        _ffi.check(st, "cnsn_plane_stats_backward")
        return dx, None, None, None


class PlaneAffine(torch.autograd.Function):
    """y = scale[n,c]*x + shift[n,c] — cnsn_plane_affine (+ cnsn_plane_dot for the backward)."""

    @staticmethod
    @_on_device_of
    def forward(ctx, x, scale, shift):
        _require_device(x, "plane_affine")
        lib = _ffi.lib()
        x = _dense(x)
        n, c, h, w = _dims(x)
        sc, sh = _f32(scale).view(-1), _f32(shift).view(-1)
        y = _out_like(x)
        st = lib.cnsn_plane_affine(_ptr(x), _DTYPES[x.dtype], n, c, h, w, _ptr(sc), _ptr(sh), _ptr(y), _stream(x))
        _ffi.check(st, "cnsn_plane_affine")
        ctx.save_for_backward(x, sc)
        ctx.meta = (scale.shape, scale.dtype, shift.shape, shift.dtype)
        return y

    @staticmethod
    @_on_device_of
    def backward(ctx, gy):
        lib = _ffi.lib()
        x, sc = ctx.saved_tensors
        n, c, h, w = _dims(x)
        gy = _dense(gy.to(x.dtype))
        dt = _DTYPES[x.dtype]
        dx = dscale = dshift = None
        if ctx.needs_input_grad[0]:
            dx = _out_like(x)
            zero = torch.zeros_like(sc)
            st = lib.cnsn_plane_affine(_ptr(gy), dt, n, c, h, w, _ptr(sc), _ptr(zero), _ptr(dx), _stream(x))
            _ffi.check(st, "cnsn_plane_affine(backward)")
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            sums = torch.empty(2, n * c, dtype=torch.float32, device=x.device)
            st = lib.cnsn_plane_dot(_ptr(gy), _ptr(x), dt, n, c, h, w, _ptr(sums), _stream(x))
            _ffi.check(st, "cnsn_plane_dot")
            s_shape, s_dtype, t_shape, t_dtype = ctx.meta
            dscale = sums[1].view(s_shape).to(s_dtype)
            dshift = sums[0].view(t_shape).to(t_dtype)
        return dx, dscale, dshift


class InstanceNorm(torch.autograd.Function):
    """nn.InstanceNorm2d (biased variance + eps, optional affine) with a dedicated backward: forward = plane statistics
    + per-plane affine (3 tensor passes), backward = per-plane sums of G and G*(x - mean) + ONE apply launch
    dx = r*w*(G - mean(G) - xhat*mean(G*xhat)) (5 passes; composing PlaneStats / PlaneAffine needs 9).
    Reference: models/imagenet/resnet_ibn_cnsn.py:24-44 instantiates nn.InstanceNorm2d(half, affine=True)."""

    @staticmethod
    @_on_device_of
    def forward(ctx, x, weight, bias, eps):
        _require_device(x, "instance_norm")
        lib = _ffi.lib()
        x = _dense(x)
        n, c, h, w = _dims(x)
        m = h * w
        assert m > 1, "InstanceNorm2d needs more than one value per plane"
        dt = _DTYPES[x.dtype]
        ms = torch.empty(2, n * c, dtype=torch.float32, device=x.device)
        # the kernel returns sqrt(unbiased var + e): with e = eps*M/(M-1),  biased var + eps = std_u^2 * (M-1)/M
        st = lib.cnsn_plane_stats(_ptr(x), dt, n, c, h, w, None, float(eps) * m / (m - 1.0), _ptr(ms), _stream(x))
        _ffi.check(st, "cnsn_plane_stats")
        mean = ms[0].view(n, c)
        rstd = 1.0 / (ms[1].view(n, c) * ((m - 1.0) / m) ** 0.5)          # 1 / sqrt(biased var + eps), (N, C)
        wf = weight.detach().float().view(1, c) if weight is not None else None
        scale = rstd * wf if wf is not None else rstd
        shift = -mean * scale
        if bias is not None:
            shift = shift + bias.detach().float().view(1, c)
        scale, shift = scale.contiguous().view(-1), shift.contiguous().view(-1)
        y = _out_like(x)
        st = lib.cnsn_plane_affine(_ptr(x), dt, n, c, h, w, _ptr(scale), _ptr(shift), _ptr(y), _stream(x))
        _ffi.check(st, "cnsn_plane_affine")
        ctx.save_for_backward(x, mean, rstd, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @_on_device_of
    def backward(ctx, gy):
        lib = _ffi.lib()
        x, mean, rstd, weight = ctx.saved_tensors
        n, c, h, w = _dims(x)
        m = float(h * w)
        dt = _DTYPES[x.dtype]
        gy = _dense(gy.to(x.dtype))
        mean64 = mean.double().contiguous().view(-1)
        sums = torch.empty(2, n * c, dtype=torch.float32, device=x.device)
        st = lib.cnsn_plane_dot_shifted(_ptr(gy), _ptr(x), dt, n, c, h, w, _ptr(mean64), _ptr(sums), _stream(x))
        _ffi.check(st, "cnsn_plane_dot_shifted")
        s1 = sums[0].view(n, c).double()
        # the kernel shifts by float(mean): sum G*(x - mean) = s2 + (float(mean) - mean) * s1 — mean IS a float here
        sgx = sums[1].view(n, c).double() * rstd.double()                 # sum G * xhat
        wd = weight.detach().double().view(1, c) if weight is not None else torch.ones(1, c, dtype=torch.float64, device=x.device)
        rw = rstd.double() * wd
        coef = torch.stack([rw, -rw * rstd.double() * sgx / m, mean.double(), -rw * s1 / m]).float().contiguous()
        dx = _out_like(x)
        st = lib.cnsn_plane_combine(_ptr(gy), _ptr(x), dt, n, c, h, w, _ptr(coef), _ptr(dx), _stream(x))
        _ffi.check(st, "cnsn_plane_combine")
        dweight = dbias = None
        if weight is not None and ctx.needs_input_grad[1]:
            dweight = sgx.sum(0).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            dbias = s1.sum(0).to(weight.dtype if weight is not None else torch.float32)
        return dx, dweight, dbias, None


class JsdConsistency(torch.autograd.Function):
    """Jensen-Shannon consistency of three (B, K) logit tensors — cnsn_jsd: loss and gradient in one launch
    (reference imagenet.py:367-381, cifar.py:173-186)."""

    @staticmethod
    @_on_device_of
    def forward(ctx, l0, l1, l2):
        lib = _ffi.lib()
        for t in (l0, l1, l2):
            if not (isinstance(t, torch.Tensor) and t.is_cuda):
                raise _ffi.CnsnError("jsd_consistency: HIP device tensors only")
        assert l0.dim() == 2 and l0.shape == l1.shape == l2.shape and l0.dtype == l1.dtype == l2.dtype
        if l0.dtype not in _DTYPES:
            raise TypeError(f"jsd_consistency: dtype {l0.dtype} not supported")
        l0, l1, l2 = _dense(l0), _dense(l1), _dense(l2)
        b, k = int(l0.shape[0]), int(l0.shape[1])
        need = any(ctx.needs_input_grad)
        loss = torch.empty((), dtype=torch.float32, device=l0.device)
        grads = torch.empty((3, b, k), dtype=l0.dtype, device=l0.device) if need else None
        wsb = lib.cnsn_jsd_workspace_bytes(b)
        ws = torch.empty(wsb // 4 + 1, dtype=torch.float32, device=l0.device)
        st = lib.cnsn_jsd(_ptr(l0), _ptr(l1), _ptr(l2), _DTYPES[l0.dtype], b, k, _ptr(loss),
                          _ptr(grads[0]) if need else None, _ptr(grads[1]) if need else None,
                          _ptr(grads[2]) if need else None, _ptr(ws), wsb, _stream(l0))
        _ffi.check(st, "cnsn_jsd")
        if need:
            ctx.save_for_backward(grads)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        (grads,) = ctx.saved_tensors
        g = grads * gloss.to(grads.dtype)          # the loss is a scalar: chain rule is one scale of 3*B*K values
        return g[0], g[1], g[2]


def jsd_consistency(logits_clean, logits_aug1, logits_aug2):
    return JsdConsistency.apply(logits_clean, logits_aug1, logits_aug2)
