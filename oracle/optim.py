"""TEST INFRASTRUCTURE - the optimiser of the train step as plain tensor arithmetic.

The reference builds ``torch.optim.SGD(params, lr, momentum=0.9, nesterov=True)`` with per-group weight decay
(run.py:175-186), steps it once per iteration (train.py:147) and scales the learning rate by
``(1 - it / max_it) ** power`` (utils/scheduler.py:3-10, train.py:150-151).  torch's update rule (torch/optim/sgd.py,
dampening 0):   g' = g + wd p ;  m' = mu m + g'  (first step: m' = g') ;  d = g' + mu m' (Nesterov) | m' ;  p' = p - lr d.
``sgd_step`` evaluates it in float64 and rounds to fp32 where the HIP kernel (ucd_amd/csrc/sgd.hip) rounds, so the kernel can
be held to it bit for bit.  PARITY: pinned against ``torch.optim.SGD`` itself - the reference's optimiser IS that class -
on the CPU over several steps and hyper-parameter sets (tests/test_optim.py::test_float64_rule_is_torch_sgd)."""


def sgd_step(p, g, m, lr, mu, wd, nesterov):
    """One update of one tensor: (p', m').  ``m`` zero-initialised (mu*0 + g' = g' is torch's first step)."""
    g1 = (g.double() + wd * p.double()).float() if wd != 0 else g
    if mu != 0:
        mb = mu * m.double() + g1.double()
        m_new = mb.float()
        g2 = (g1.double() + mu * mb).float() if nesterov else m_new
    else:
        m_new, g2 = m, g1
    return (p.double() - lr * g2.double()).float(), m_new


def poly_lr(base_lr, iteration, max_iters, power=0.9):
    return base_lr * (1 - iteration / max_iters) ** power
