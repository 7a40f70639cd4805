"""Training-step structure around the hot path, restated from the reference's trainers (which cannot
be imported: torchvision missing, argparse at import time):

  cifar.py:117-145    train_cn               r < cn_prob -> net(x, aug=True)
  cifar.py:148-208    train_cn_consistency   clean + 2 CrossNorm views, CE + consist_wt * JSD
  imagenet.py:195-250 train_cn_image         CrossNorm applied to the IMAGE batch (cn_op on (B,3,H,W))
  imagenet.py:253-334 train_cn_image_consist clean + 2 image-space CrossNorm views, CE + wt * JSD
  imagenet.py:337-406 train_cn_image_augmix  one CrossNorm call on the concatenated 3-view batch

Only the arithmetic and the RNG draw order are reproduced; data loading, meters and logging are not."""
import numpy as np
import torch
import torch.nn.functional as F


def jsd_consistency(logits_clean, logits_aug1, logits_aug2):
    """Jensen-Shannon consistency of three views (cifar.py:173-186, imagenet.py:367-381):
    mixture = clamp(mean of the 3 softmaxes, 1e-7, 1).log(); mean of the three KL(mixture || p_i),
    each with reduction 'batchmean'."""
    from ..functional import jsd_consistency as _jsd      # one fused launch (cnsn_jsd): loss and gradient;
    return _jsd(logits_clean, logits_aug1, logits_aug2)   # HIP device tensors only, like the op itself


class StepGuard:
    """One training step that is applied from complete gradients or not at all — on every data-parallel rank alike.

    A cluster-resident launch whose bounded wait runs out ("when a bounded wait runs out": the GPU was shared
    with something that kept part of a persistent grid off the device for seconds) gives up, marks the planes it still
    owed with NaNs and bumps a host counter.  What that attempt touched besides the gradients:
      * every BatchNorm running statistic downstream ((1-m)*rm + m*NaN = NaN) and SelfNorm's gate buffers,
      * every `num_batches_tracked` (incremented once more by the repeat),
      * the host RNG streams CrossNorm draws from (torch CPU generator, numpy global: cnsn.py:62-76) and `r < cn_prob`.
    `run()` snapshots those before the attempt (ONE multi-tensor copy per dtype), lets the attempt run to its END with
    the module layer's per-call poll silenced (`_ffi.deferred_timeouts`: a CnsnError in the middle of a forward would
    take this rank out of the step while its peers go on to DDP's gradient all-reduce), waits for the stream, and
    all-reduces "did a launch of mine give up" over the ranks (`data_parallel.agree_to_repeat`, 4 bytes, MAX) BEFORE the
    optimizer runs.  If any rank says yes, every rank restores its snapshot, switches the cluster kernels off
    (`data_parallel.degrade_all`) and repeats: the ranks issue the same collectives in the same order whatever happens.
    The reference's loops (cifar.py:136-138, imagenet.py:240-244) have no counterpart: eager PyTorch cannot time out."""

    def __init__(self, *modules, group=None, restore_rng=True, max_attempts=3, optimizer=None, rearm_after=20):
        """optimizer: ALSO snapshot the parameters and the optimizer's state tensors (`save` / `restore` then cover a whole
        WINDOW of applied steps — what a launch-bound loop uses to poll once per window instead of synchronising the
        stream in every step: bench.py's model workloads; `run()` itself never needs it, it polls before the optimizer)."""
        self.slots = [(m, name) for mod in modules for m in mod.modules() for name, b in m._buffers.items() if b is not None]
        self.modules, self.optimizer = modules, optimizer
        self.group, self.restore_rng, self.max_attempts = group, restore_rng, max_attempts
        self.snap, self.rng = None, None
        # The way back: after `rearm_after` applied steps without a time-out the cluster kernels are tried again
        # (`data_parallel.rearm_all`; 20 by default — a degraded step costs 1.3-1.6 x a healthy one, so the way back should be
        # short; round 5 waited 200); a relapse doubles the interval.  Degradations are rank-agreed and every rank counts
        # the same applied steps, so all ranks re-arm at the same step without a collective.  None / 0: never.
        self.rearm_after = rearm_after
        self.clean_steps = 0             # applied steps since the last degradation
        self.degraded = False
        self.rearms = 0                  # times this guard switched the cluster kernels back on
        self.repeats = 0                 # steps repeated so far (all ranks count the same)
        self.local_timeouts = 0          # launches of THIS rank that gave up
        self._defaults_done = False

    # The guard lives in the network's __dict__ (`_guard_of`): `copy.deepcopy(net)` (EMA / SWA copies) and `torch.save(net)`
    # would otherwise drag along the snapshot clones of every buffer (and of the parameters and optimizer state when an
    # optimizer was passed) and the process-group handle, which does not pickle.  A copy gets no guard; `_guard_of` builds it
    # a fresh one on its first step.
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (type(None), ())

    def _live(self):
        live = [m._buffers[name] for m, name in self.slots]
        if self.optimizer is not None:
            live += [p.data for mod in self.modules for p in mod.parameters()]
            live += [v for st in self.optimizer.state.values() for v in st.values() if torch.is_tensor(v)]
        return live

    def save(self):
        live = self._live()
        if self.snap is None or len(self.snap) != len(live) or any(
                s.shape != b.shape or s.dtype != b.dtype or s.device != b.device for s, b in zip(self.snap, live)):
            self.snap = [torch.empty_like(b) for b in live]
        if live:
            with torch.no_grad():
                torch._foreach_copy_(self.snap, live)
        if self.restore_rng:
            self.rng = (np.random.get_state(), torch.get_rng_state())

    def restore(self):
        live = self._live()
        assert len(live) == len(self.snap), "StepGuard.restore: tensors appeared since save() (optimizer state created mid-window)"
        if live:
            with torch.no_grad():
                torch._foreach_copy_(live, self.snap)
        if self.restore_rng and self.rng is not None:
            np.random.set_state(self.rng[0])
            torch.set_rng_state(self.rng[1])

    def degrade(self):
        """every rank, at the same step: cluster kernels off (a relapse after a re-arm doubles the way back)"""
        from .. import data_parallel as dp
        dp.degrade_all()
        if self.rearms and self.rearm_after:
            self.rearm_after = min(self.rearm_after * 2, 1 << 20)
        self.degraded, self.clean_steps = True, 0

    def step_applied(self, k=1):
        """count `k` applied steps; re-arm the cluster kernels when the clean stretch since a degradation is long enough"""
        if not self.degraded or not self.rearm_after:
            return
        self.clean_steps += k
        if self.clean_steps >= self.rearm_after:
            from .. import data_parallel as dp
            dp.rearm_all()
            self.degraded, self.clean_steps = False, 0
            self.rearms += 1

    def run(self, compute_loss, optimizer):
        """compute_loss(): the forward(s) of the step, returns the loss (RNG draws included: a repeat re-draws the same
        values).  Then zero_grad / backward [gradient all-reduce inside] / settle / agree / optimizer.step."""
        from .. import _ffi
        from .. import data_parallel as dp
        if not self._defaults_done:
            _ffi.under_process_group_defaults()      # (a 2 s bound on cluster waits when peers would wait with us)
            self._defaults_done = True
        for _ in range(self.max_attempts):
            self.save()
            with _ffi.deferred_timeouts():
                loss = compute_loss()
                optimizer.zero_grad()
                loss.backward()
            dev = loss.device if loss.is_cuda else None
            new = _ffi.settle_step(dev, raise_on_timeout=False) if dev is not None else _ffi.poll_timeouts()
            self.local_timeouts += new
            if dp.agree_to_repeat(new, dev, self.group) == 0:
                optimizer.step()
                self.step_applied()
                return loss.detach()
            self.repeats += 1                        # some rank's cluster launch gave up: nobody applies this attempt
            self.degrade()
            self.restore()
        raise _ffi.CnsnError(f"training step: cluster-resident launches still time out after {self.max_attempts} attempts "
                             "(every rank raises this together)")


def _guard_of(net):
    """the StepGuard of a network: built on first use and kept ON the network (an attribute outside `_modules` /
    `_buffers`, so state_dict and `.modules()` do not see it) — it dies with the network.  A dictionary keyed by the
    network, even a weak one, would keep the network alive through the guard's own references to its modules."""
    g = net.__dict__.get("_cnsn_step_guard")
    if g is None:
        g = net.__dict__["_cnsn_step_guard"] = StepGuard(net)
    return g


def _apply(net, optimizer, compute_loss, guard):
    """zero_grad / backward / step of the reference's loops (cifar.py:136-138) around `compute_loss()`; with `guard` through
    the network's `StepGuard`: gradients of a cluster launch that gave up never reach the weights, the buffers and RNG
    streams the failed attempt moved are put back, and data-parallel ranks repeat together."""
    if guard:
        try:
            on_device = next(net.parameters()).is_cuda
        except StopIteration:
            on_device = True
        if on_device:
            return _guard_of(net).run(compute_loss, optimizer)
    # (a network on the host — the step-structure tests run on host restatements of the modules — has no cluster launch that could give up:
    #  nothing to snapshot, nothing to poll, and libcnsn_hip.so need not even be built)
    loss = compute_loss()
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.detach()


def repeat_on_timeout(step_fn, *args, **kwargs):
    """For a loop that does NOT go through `StepGuard` (`guard=False` steps, a user's own step function) on ONE process:
    run the step; when the library reports that a cluster-resident launch gave up ("repeat the step") run it once more —
    by then the library uses the two-pass kernels.  Only the weights are protected, and only if `step_fn` settles the
    stream before its optimizer runs (`_ffi.settle_step`); BatchNorm running statistics the failed attempt moved stay
    moved.  Under an initialised process group this refuses to run: one rank repeating alone would issue a second set of
    gradient all-reduces its peers never match — use `StepGuard` (the steps' default `guard=True`), whose repeat is
    agreed by all ranks (`data_parallel.agree_to_repeat`)."""
    import torch.distributed as dist
    from .._ffi import CnsnError
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        raise CnsnError("repeat_on_timeout is for single-process loops; under a process group use callers.steps.StepGuard "
                        "(all ranks repeat together)")
    try:
        return step_fn(*args, **kwargs)
    except CnsnError as e:
        if "repeat" not in str(e):
            raise
        return step_fn(*args, **kwargs)


def train_step_cn(net, x, target, optimizer, cn_prob, guard=True):
    """One feature-level CrossNorm step (cifar.py:123-140): draw r first, then forward with aug."""
    def compute_loss():
        r = np.random.rand(1)
        return F.cross_entropy(net(x, aug=bool(r < cn_prob)), target)
    return _apply(net, optimizer, compute_loss, guard)


def train_step_cn_consistency(net, x, target, optimizer, consist_wt, cn_prob=1.0, jsd=jsd_consistency, guard=True):
    """Consistency step (cifar.py:163-196).  Draw order of the reference: `r = np.random.rand(1)` FIRST; only when
    `r < cn_prob` the clean view + two independently armed CrossNorm views + JSD run (:165-187), otherwise plain
    cross-entropy on `net(x, aug=False)` (:188-190).  `jsd`: the consistency term (tests of the step structure on
    host tensors pass a host restatement)."""
    def compute_loss():
        r = np.random.rand(1)
        if r < cn_prob:
            logits_clean = net(x, aug=False)
            loss = F.cross_entropy(logits_clean, target)
            logits_aug1 = net(x, aug=True)
            logits_aug2 = net(x, aug=True)
            return loss + consist_wt * jsd(logits_clean, logits_aug1, logits_aug2)
        return F.cross_entropy(net(x, aug=False), target)
    return _apply(net, optimizer, compute_loss, guard)


def image_space_crossnorm(images, cn_prob, beta, crop, cn_op):
    """imagenet.py:211-215: r = np.random.rand(1); if r < cn_prob: images = cn_op(images, beta, crop)."""
    r = np.random.rand(1)
    if r < cn_prob:
        images = cn_op(images, crop=crop, beta=beta)
    return images


def train_step_image_cn_views(net, views, target, optimizer, cn_prob, beta, crop, cn_op, jsd_wt=12.0,
                              jsd=jsd_consistency, guard=True):
    """AugMix-style 3-view step (imagenet.py:352-381): concatenate the views, ONE image-space CrossNorm
    call on the (3B,3,H,W) batch with probability cn_prob, one forward, CE on the clean third + 12*JSD."""
    b = views[0].size(0)

    def compute_loss():
        batch = image_space_crossnorm(torch.cat(views, 0), cn_prob, beta, crop, cn_op)
        l_clean, l_a1, l_a2 = torch.split(net(batch), b)
        return F.cross_entropy(l_clean, target) + jsd_wt * jsd(l_clean, l_a1, l_a2)
    return _apply(net, optimizer, compute_loss, guard)


class _IdleBlock(torch.nn.Module):
    """One WideResNet block in the form `WideResNetCNSN.forward` calls it with the fused tail, as a tensor-in / tensor-out
    module (what torch.cuda.make_graphed_callables captures).  Shares the block's (and the next BatchNorm's) parameters."""

    def __init__(self, blk, next_bn, want_y, has_pre):
        super().__init__()
        self.blk, self.next_bn, self.want_y, self.has_pre = blk, next_bn, want_y, has_pre

    def forward(self, h, pre=None):
        y, z = self.blk(h, pre if self.has_pre else None, next_bn=self.next_bn, want_y=self.want_y)
        assert z is not None and (y is not None) == self.want_y
        return (y, z) if self.want_y else z


class _GraphedBlocks:
    """An ARMED WideResNet step cannot be replayed as a whole (which sites fire, their permutations and crop boxes are drawn
    per step and are launch ARGUMENTS), but 16 of its 18 blocks are idle all the same.  Every block is captured once in its
    idle form — forward and backward, through torch.cuda.make_graphed_callables — and an armed step replays the idle ones and
    runs the two armed ones eagerly: ~700 launches become ~60.  Same arithmetic as `WideResNetCNSN.forward(x, aug=True)`,
    same RNG draws (`_enable_cross_norm` is called first, exactly as there).
    MEASURED AND LEFT OFF (bench.py --workload wrn40, MI355X, ROCm 7.2, torch 2.10): 8.45 ms per step against 7.11 with the
    armed steps fully eager — 36 graph launches per step, each with its input copies into the callable's static buffers and
    the graphed autograd function's own book-keeping, cost more than the ~640 launches they replace.  Kept as an option
    (`GraphedIdleStep(..., graph_blocks=True)`, `bench.py --block-graphs`) and under test: the trajectory equals the eager one."""

    def __init__(self, net, x):
        from . import _sites
        assert _sites.FUSE_TAIL and net.pos == "post", "per-block graphs follow the fused-tail form of WideResNetCNSN.forward"
        self.net = net
        self.blocks = [b for stage in (net.block1, net.block2, net.block3) for b in stage.layer]
        n = len(self.blocks)
        self.next_bn = [net.bn1 if i + 1 == n else self.blocks[i + 1].bn1 for i in range(n)]
        self.want_y = []
        for i, blk in enumerate(self.blocks):
            last = i + 1 == n
            w = (not last) and self.blocks[i + 1].same_width and self.blocks[i + 1].pos != "pre"
            self.want_y.append(w or not blk.cnsn_tail_ok())
        # sample inputs of every block: one eager pass in the idle form
        samples, h, pre = [], net.conv1(x), None
        with torch.no_grad():
            for i, blk in enumerate(self.blocks):
                samples.append((h, pre))
                y, z = blk(h, pre, next_bn=self.next_bn[i], want_y=self.want_y[i])
                assert z is not None, "per-block graphs need the fused tail at every site (CNSN.forward_block_bn)"
                h, pre = (y if y is not None else z), z
        mods, args = [], []
        for i, (hh, pp) in enumerate(samples):
            mods.append(_IdleBlock(self.blocks[i], self.next_bn[i], self.want_y[i], pp is not None).to(x.device).train())
            a = (hh.detach().clone().requires_grad_(),) + ((pp.detach().clone().requires_grad_(),) if pp is not None else ())
            args.append(a)
        self.graphed = torch.cuda.make_graphed_callables(tuple(mods), tuple(args), allow_unused_input=True)

    def forward(self, x):
        net = self.net
        net._enable_cross_norm()
        h, pre = net.conv1(x), None
        z = None
        for i, blk in enumerate(self.blocks):
            cn = getattr(blk.cnsn, "crossnorm", None)
            if cn is not None and cn.active:               # armed: eager, fresh draws
                y, z = blk(h, pre, next_bn=self.next_bn[i], want_y=self.want_y[i])
                if z is None:                               # (no tail at this site this step)
                    h, pre = y, None
                    if i + 1 == len(self.blocks):
                        z = net.relu(net.bn1(y))
                else:
                    h, pre = (y if y is not None else z), z
            else:                                           # idle: the block's captured forward (and, later, backward)
                out = self.graphed[i](h, pre) if pre is not None else self.graphed[i](h)
                y, z = out if self.want_y[i] else (None, out)
                h, pre = (y if y is not None else z), z
        return net.fc(F.avg_pool2d(z, 8).view(z.size(0), -1))


class GraphedIdleStep:
    """Launch-bound networks (WideResNet-40-2: ~700 small launches per step for 8 ms of host time against ~3 ms of GPU
    work): a training step whose CrossNorm sites are all IDLE (`aug=False`: SelfNorm only — nothing is drawn on the host)
    is captured ONCE into a HIP graph — forward, loss, backward, optimizer — and replayed; a step with armed sites
    (host-drawn permutation and boxes are launch arguments) runs eagerly as before.  Same arithmetic, same RNG draws:
    `r = np.random.rand(1)` is still drawn first every step (cifar.py:127-131).  Static shapes: `x`, `target` are copied into
    the captured buffers.  The op itself needs nothing special for this: every entry point only enqueues work on the
    caller's stream ("HIP graphs")."""

    def __init__(self, net, optimizer, x, target, warmup=3, graph_blocks=False):
        """graph_blocks: ARMED steps replay one small graph per idle block as well (`_GraphedBlocks` below) — WideResNet with
        the fused tail only; the armed blocks, the head, the loss and the optimizer stay eager."""
        import copy
        self.net, self.opt = net, optimizer
        self.blocks = None
        self.x, self.y = x.clone(), target.clone()
        for grp in optimizer.param_groups:    # the restore below turns warm-up-created state into "no step taken yet" by zeroing
            if grp.get("dampening", 0) != 0:  # it: SGD's first step is buf = grad, a zeroed buffer gives (1 - dampening) * grad
                raise ValueError("GraphedIdleStep: SGD with dampening != 0 is not supported (the first real step after the "
                                 "warm-up would differ from the reference loop's)")
        for m in net.modules():     # float(num_batches_tracked) would synchronise under capture
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.momentum is None and m.track_running_stats:
                raise ValueError("GraphedIdleStep: BatchNorm with momentum=None (cumulative average) cannot be captured")
        # The warm-up and the capture pass run REAL steps (weights, momentum buffers, BatchNorm running statistics,
        # num_batches_tracked): snapshot everything first and put it back afterwards, so that building the graph does
        # not move the training trajectory away from the reference loop's.
        model_state = copy.deepcopy(net.state_dict())
        opt_before = {p: {k: (v.clone() if torch.is_tensor(v) else copy.deepcopy(v)) for k, v in st.items()}
                      for p, st in optimizer.state.items()}
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):                       # warm-up off the capture stream (allocator, optimizer state)
            # a fresh optimizer takes one real step at least: its state tensors (momentum buffers) must exist before the
            # capture — captured, SGD's first-step `buf = clone(grad)` would be replayed as the first step for ever
            for _ in range(warmup if len(optimizer.state) else max(warmup, 1)):
                self._body()
        torch.cuda.current_stream(x.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._body()
        torch.cuda.synchronize(x.device)
        if graph_blocks:
            self.blocks = _GraphedBlocks(net, self.x)       # (its warm-up passes move BatchNorm statistics too: restored below)
        with torch.no_grad():                               # in place: the graph holds the parameters' / buffers' addresses
            live = net.state_dict()
            for k, v in model_state.items():
                live[k].copy_(v)
            for p, st in optimizer.state.items():
                old = opt_before.get(p)
                for name, val in st.items():
                    if not torch.is_tensor(val):
                        if old is not None and name in old:
                            st[name] = old[name]
                        elif isinstance(val, (int, float)) and not isinstance(val, bool):
                            st[name] = type(val)(0)     # a counter the warm-up created (a float `step`): not taken yet
                        elif val is not None:
                            raise ValueError(f"GraphedIdleStep: optimizer state {name!r} of type {type(val).__name__} was "
                                             "created by the warm-up and cannot be put back to its initial value")
                        continue
                    if old is not None and torch.is_tensor(old.get(name)):
                        val.copy_(old[name])                # the state the optimizer had before the warm-up
                    else:
                        val.zero_()                         # created by the warm-up: back to "no step taken yet" (SGD's first
                                                            # `buf = grad` and `0 * buf + grad` agree; Adam starts from zeros)

    def _body(self):
        loss = F.cross_entropy(self.net(self.x, aug=False), self.y)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        self.opt.step()
        return loss.detach()

    def step(self, x, target, cn_prob):
        from .. import _ffi
        _ffi.check_resident_health("GraphedIdleStep.step")  # a replay has no per-call poll of its own
        r = np.random.rand(1)                               # drawn first, armed or not (cifar.py:127)
        if r < cn_prob:                                     # armed: eager (fresh draws per step)
            logits = self.blocks.forward(x) if self.blocks is not None else self.net(x, aug=True)
            loss = F.cross_entropy(logits, target)
            self.opt.zero_grad(set_to_none=True)
            loss.backward()
            self.opt.step()
            return loss.detach()
        if x is not self.x:
            self.x.copy_(x)
            self.y.copy_(target)
        self.graph.replay()
        return self.loss
