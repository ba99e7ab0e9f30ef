"""TEST INFRASTRUCTURE - CPU restatement of the cross-rank batch-norm statistics of InPlaceABNSync.

The reference takes them from the inplace-abn extension (segmentation_module.py:17 `InPlaceABNSync`; third-party, not
vendored under /root/reference: parity pinned to "statistics of the concatenated global batch", i.e. what
torch.nn.SyncBatchNorm / F.batch_norm over the whole batch produce).  Each rank contributes (mean_r, M2_r) over its
own M rows; equal M per rank (DistributedSampler with drop_last, run.py:147-149).
"""
import numpy as np


def rank_moments(x):
    """x: [M, C] rows of one rank -> (mean_r [C], M2_r [C]) in float64."""
    x = np.asarray(x, dtype=np.float64)
    mean = x.mean(axis=0)
    return mean, ((x - mean) ** 2).sum(axis=0)


def combine_rank_moments(gathered, m_local):
    """gathered: [world, 2, C] = per-rank (mean_r, M2_r) -> (mean [C], biased var [C], M2 [C]) of the global batch
    (Chan et al.'s pairwise update, closed form for equal counts)."""
    g = np.asarray(gathered, dtype=np.float64)
    world = g.shape[0]
    mean = g[:, 0].mean(axis=0)
    m2 = g[:, 1].sum(axis=0) + m_local * ((g[:, 0] - mean) ** 2).sum(axis=0)
    return mean, m2 / (world * m_local), m2
