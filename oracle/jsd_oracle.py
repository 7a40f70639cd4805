"""TEST INFRASTRUCTURE — CPU oracle of the Jensen-Shannon consistency loss.

Restates, op for op, what the reference's trainers compute between the logits and `consist_loss`
(imagenet.py:367-376 and :291-300, cifar.py:173-182 and :233-242).  PINNED: those files cannot be imported in this
image (torchvision / tensorboardX are absent and they parse arguments at import time), but
tests/golden/gen_golden_jsd.py parses them, compiles the reference's own three statements from their AST nodes and
executes them on seeded logits; this restatement reproduces the resulting fixture (tests/golden/g8_jsd.npz: loss and
logit gradients, fp32 and fp64) bit for bit — tests/test_oracle_golden.py::test_jsd_oracle_reproduces_reference.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module."""
import torch
import torch.nn.functional as F


def jsd_consistency(logits_clean, logits_aug1, logits_aug2):
    p_clean, p_aug1, p_aug2 = (F.softmax(logits_clean, dim=1), F.softmax(logits_aug1, dim=1),
                               F.softmax(logits_aug2, dim=1))                                   # :367-370
    p_mixture = torch.clamp((p_clean + p_aug1 + p_aug2) / 3., 1e-7, 1).log()                  # :373
    return (F.kl_div(p_mixture, p_clean, reduction='batchmean') +                              # :374-376
            F.kl_div(p_mixture, p_aug1, reduction='batchmean') +
            F.kl_div(p_mixture, p_aug2, reduction='batchmean')) / 3.
