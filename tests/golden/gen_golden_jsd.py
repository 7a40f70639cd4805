#!/usr/bin/env python3
"""G8: golden vectors of the reference's Jensen-Shannon consistency arithmetic (build container only).

    python tests/golden/gen_golden_jsd.py     # writes tests/golden/g8_jsd.npz

The reference's trainers cannot be IMPORTED here (torchvision / tensorboardX are absent and they parse arguments at
import time), but their source can be PARSED: this script reads /root/reference/imagenet.py and cifar.py at run time,
picks — with `ast`, by assignment target — the three statements between the logits and `consist_loss`
(imagenet.py:367-376 inside train_cn_image_augmix / train_cn_image_consist, cifar.py:173-182 inside
train_cn_consistency), compiles exactly those nodes and executes them on seeded logits with autograd.  Nothing of the
reference's text is stored: the fixture holds inputs, the loss and the logit gradients (fp32 and fp64).  The script
also checks that the statements of all the trainer functions are the same computation (identical outputs)."""
import ast
import os
import sys

sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
TARGETS = (("p_clean", "p_aug1", "p_aug2"), ("p_mixture",), ("consist_loss",))


def _names(t):
    if isinstance(t, ast.Name):
        return (t.id,)
    if isinstance(t, ast.Tuple):
        return tuple(e.id for e in t.elts if isinstance(e, ast.Name))
    return ()


def reference_jsd_programs():
    """{(file, function): code object} — the reference's own statements, compiled from its own AST nodes."""
    progs = {}
    for path in ("/root/reference/imagenet.py", "/root/reference/cifar.py"):
        tree = ast.parse(open(path).read(), filename=path)
        for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
            picked = []
            for node in ast.walk(fn):
                if isinstance(node, ast.Assign) and len(node.targets) == 1 and _names(node.targets[0]) in TARGETS:
                    picked.append(node)
            if {_names(n.targets[0]) for n in picked} == set(TARGETS):
                picked.sort(key=lambda n: n.lineno)
                mod = ast.Module(body=picked, type_ignores=[])
                progs[(os.path.basename(path), fn.name, picked[0].lineno, picked[-1].end_lineno)] = compile(mod, path, "exec")
    return progs


def run(code, logits, dtype):
    zs = [torch.from_numpy(z).to(dtype).requires_grad_() for z in logits]
    ns = {"F": F, "torch": torch, "logits_clean": zs[0], "logits_aug1": zs[1], "logits_aug2": zs[2]}
    exec(code, ns)                                   # the reference's own three statements
    loss = ns["consist_loss"]
    loss.backward()
    return loss.detach().numpy(), [z.grad.numpy() for z in zs]


CASES = {   # name: (B, K, logit scale, seed, classes forced to -40 so the mixture falls under the 1e-7 clamp)
    "small": (8, 10, 2.0, 1, 0),
    "cifar100": (16, 100, 0.5, 2, 0),
    "imagenet": (12, 1000, 4.0, 3, 0),
    "clamped": (4, 6, 1.0, 4, 1),
    "one_class": (3, 1, 1.0, 5, 0),
}


def main():
    torch.set_num_threads(4)
    progs = reference_jsd_programs()
    assert len(progs) >= 2, progs.keys()
    print("reference statements found in:", *(f"{f}:{a}-{b} ({fn})" for (f, fn, a, b) in progs), sep="\n  ")
    out = {"sources": np.array([f"{f}:{a}-{b}:{fn}" for (f, fn, a, b) in progs])}
    for name, (b, k, scale, seed, dead) in CASES.items():
        g = torch.Generator().manual_seed(seed)
        logits = [(torch.randn(b, k, generator=g, dtype=torch.float64) * scale).float().numpy() for _ in range(3)]
        for z in logits:
            z[:, :dead] = -40.0
        for tag, dtype in (("f32", torch.float32), ("f64", torch.float64)):
            results = [run(code, logits, dtype) for code in progs.values()]
            for loss, grads in results[1:]:          # every trainer function computes the same thing, bit for bit
                assert np.array_equal(loss, results[0][0]) and all(np.array_equal(x, y) for x, y in zip(grads, results[0][1]))
            loss, grads = results[0]
            out[f"{name}_{tag}_loss"] = loss
            for i, gr in enumerate(grads):
                out[f"{name}_{tag}_grad{i}"] = gr
        for i, z in enumerate(logits):
            out[f"{name}_logits{i}"] = z
    path = os.path.join(HERE, "g8_jsd.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
