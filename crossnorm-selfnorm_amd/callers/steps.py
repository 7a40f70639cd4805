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


def train_step_cn(net, x, target, optimizer, cn_prob):
    """One feature-level CrossNorm step (cifar.py:123-140): draw r first, then forward with aug."""
    r = np.random.rand(1)
    logits = net(x, aug=bool(r < cn_prob))
    loss = F.cross_entropy(logits, target)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.detach()


def train_step_cn_consistency(net, x, target, optimizer, consist_wt, cn_prob=1.0, jsd=jsd_consistency):
    """Consistency step (cifar.py:163-196).  Draw order of the reference: `r = np.random.rand(1)` FIRST; only when
    `r < cn_prob` the clean view + two independently armed CrossNorm views + JSD run (:165-187), otherwise plain
    cross-entropy on `net(x, aug=False)` (:188-190).  `jsd`: the consistency term (tests of the step structure on
    host tensors pass a host restatement)."""
    r = np.random.rand(1)
    if r < cn_prob:
        logits_clean = net(x, aug=False)
        loss = F.cross_entropy(logits_clean, target)
        logits_aug1 = net(x, aug=True)
        logits_aug2 = net(x, aug=True)
        loss = loss + consist_wt * jsd(logits_clean, logits_aug1, logits_aug2)
    else:
        loss = F.cross_entropy(net(x, aug=False), target)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.detach()


def image_space_crossnorm(images, cn_prob, beta, crop, cn_op):
    """imagenet.py:211-215: r = np.random.rand(1); if r < cn_prob: images = cn_op(images, beta, crop)."""
    r = np.random.rand(1)
    if r < cn_prob:
        images = cn_op(images, crop=crop, beta=beta)
    return images


def train_step_image_cn_views(net, views, target, optimizer, cn_prob, beta, crop, cn_op, jsd_wt=12.0,
                              jsd=jsd_consistency):
    """AugMix-style 3-view step (imagenet.py:352-381): concatenate the views, ONE image-space CrossNorm
    call on the (3B,3,H,W) batch with probability cn_prob, one forward, CE on the clean third + 12*JSD."""
    b = views[0].size(0)
    batch = image_space_crossnorm(torch.cat(views, 0), cn_prob, beta, crop, cn_op)
    logits = net(batch)
    l_clean, l_a1, l_a2 = torch.split(logits, b)
    loss = F.cross_entropy(l_clean, target) + jsd_wt * jsd(l_clean, l_a1, l_a2)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.detach()


class GraphedIdleStep:
    """Launch-bound networks (WideResNet-40-2: ~700 small launches per step for 8 ms of host time against ~3 ms of GPU
    work): a training step whose CrossNorm sites are all IDLE (`aug=False`: SelfNorm only — nothing is drawn on the host)
    is captured ONCE into a HIP graph — forward, loss, backward, optimizer — and replayed; a step with armed sites
    (host-drawn permutation and boxes are launch arguments) runs eagerly as before.  Same arithmetic, same RNG draws:
    `r = np.random.rand(1)` is still drawn first every step (cifar.py:127-131).  Static shapes: `x`, `target` are copied into
    the captured buffers.  The op itself needs nothing special for this: every entry point only enqueues work on the
    caller's stream ("HIP graphs")."""

    def __init__(self, net, optimizer, x, target, warmup=3):
        self.net, self.opt = net, optimizer
        self.x, self.y = x.clone(), target.clone()
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):                       # warm-up off the capture stream (allocator, momentum buffers)
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream(x.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._body()

    def _body(self):
        loss = F.cross_entropy(self.net(self.x, aug=False), self.y)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        self.opt.step()
        return loss.detach()

    def step(self, x, target, cn_prob):
        r = np.random.rand(1)                               # drawn first, armed or not (cifar.py:127)
        if r < cn_prob:                                     # armed: eager (fresh draws per step)
            loss = F.cross_entropy(self.net(x, aug=True), target)
            self.opt.zero_grad(set_to_none=True)
            loss.backward()
            self.opt.step()
            return loss.detach()
        if x is not self.x:
            self.x.copy_(x)
            self.y.copy_(target)
        self.graph.replay()
        return self.loss
