"""Train-time metrics of the reference (utils/metrics.py:31-45,112-135).

``accuracy``/``dice`` keep the reference signatures for (H,W)/(B,H,W) tensors.  The
trainer's hot path does not call them per image: it reads the sums produced by
the ``wesup_seg_metrics`` kernel and applies the same formulas
(``accuracy_from_sums``/``dice_from_sums``) -- one host sync per step instead of four."""
import numpy as np
import torch


def accuracy(P, G):
    if torch.is_tensor(P) and torch.is_tensor(G):
        return (P == G).float().mean().item()
    return (np.array(P) == np.array(G)).mean()


def dice(S, G, epsilon=1e-7):
    if torch.is_tensor(S) and torch.is_tensor(G):
        S = S.unsqueeze(0) if len(S.size()) == 2 else S
        G = G.unsqueeze(0) if len(G.size()) == 2 else G
        S, G = S.float(), G.float()
        d = 2 * (G * S).sum(dim=(1, 2)) / (G.sum(dim=(1, 2)) + S.sum(dim=(1, 2)) + epsilon)
        return d.mean().item()
    S, G = np.array(S), np.array(G)
    S = np.expand_dims(S, 0) if len(S.shape) == 2 else S
    G = np.expand_dims(G, 0) if len(G.shape) == 2 else G
    d = 2 * (G * S).sum(axis=(1, 2)) / (G.sum(axis=(1, 2)) + S.sum(axis=(1, 2)) + epsilon)
    return d.mean()


def accuracy_from_sums(sums, n_pixels):
    """sums (B,4) = {#(P==G), sum(P*G), sum(P), sum(G)} per image -> mean accuracy over the batch."""
    return float(np.mean(sums[:, 0] / n_pixels))


def dice_from_sums(sums, epsilon=1e-7):
    return float(np.mean(2 * sums[:, 1] / (sums[:, 3] + sums[:, 2] + epsilon)))


# ---------------------------------------------------------------------------------------------------------------------
# GlaS challenge metrics (utils/metrics.py:48-281).  Evaluation-time, CPU, numpy -- as in the
# reference.  Restated around ONE contingency table of the two labelled maps (pixels per (segmented object, ground
# truth object) pair) instead of the reference's per-object boolean masks: the "corresponding object" of an object
# is the arg-max of its row / column (the mode of the labels it overlaps, smallest label on ties like scipy.stats.mode).
# Connected components: scipy.ndimage.label with 8-connectivity (skimage.measure.label's default for 2-D input).
# ---------------------------------------------------------------------------------------------------------------------
def _to_numpy(a):
    return a.detach().cpu().numpy() if torch.is_tensor(a) else np.array(a)


def label(mask):
    """Connected components of the non-zero pixels, 8-connected, numbered 1..n in raster order of first pixel."""
    from scipy import ndimage
    lab, _ = ndimage.label(np.asarray(mask) != 0, structure=np.ones((3, 3), dtype=np.int32))
    return lab


def _contingency(S, G):
    nS, nG = int(S.max()), int(G.max())
    C = np.bincount((S.astype(np.int64) * (nG + 1) + G).ravel(), minlength=(nS + 1) * (nG + 1))
    return C.reshape(nS + 1, nG + 1), nS, nG


def _partner(C):
    """For every row s >= 1: the column g >= 1 with the largest overlap (0 when the object overlaps nothing)."""
    if C.shape[1] <= 1:
        return np.zeros(C.shape[0], dtype=np.int64)
    best = C[:, 1:].argmax(1) + 1
    best[C[:, 1:].max(1) == 0] = 0
    best[0] = 0
    return best


def detection_f1(S, G, overlap_threshold=0.5, epsilon=1e-7):
    """F1 of object detection: a segmented object is a true positive when it covers more than ``overlap_threshold``
    of the ground-truth object it overlaps most (utils/metrics.py:49-109)."""
    S, G = label(_to_numpy(S)), label(_to_numpy(G))
    C, nS, nG = _contingency(S, G)
    if nS == 0 and nG == 0:
        return 1
    if nS == 0 or nG == 0:
        return 0
    partner = _partner(C)
    area_G = C.sum(0)
    s = np.arange(1, nS + 1)
    g = partner[1:]
    hit = (g > 0) & (C[s, g] / np.maximum(area_G[g], 1) > overlap_threshold)
    TP = int(hit.sum())
    FP = nS - TP
    FN = nG - TP
    precision = TP / (TP + FP)
    recall = TP / (TP + FN)
    return (2 * precision * recall) / (precision + recall + epsilon)


def _weighted_pair_dice(C, epsilon=1e-7):
    """sum_s (area_s / total) * dice(object s, its partner) over the rows of C."""
    partner = _partner(C)
    area_row, area_col = C.sum(1), C.sum(0)
    total = area_row[1:].sum()
    s = np.arange(1, C.shape[0])
    g = partner[1:]
    inter = np.where(g > 0, C[s, g], 0)
    other = np.where(g > 0, area_col[g], 0)
    d = 2 * inter / (area_row[1:] + other + epsilon)
    return float(((area_row[1:] / total) * d).sum())


def object_dice(S, G):
    """Object-level Dice (utils/metrics.py:139-196)."""
    S, G = label(_to_numpy(S)), label(_to_numpy(G))
    C, nS, nG = _contingency(S, G)
    if nS == 0 and nG == 0:
        return 1
    if nS == 0 or nG == 0:
        return 0
    return (_weighted_pair_dice(C) + _weighted_pair_dice(C.T)) / 2


def hausdorff(S, G):
    """Symmetric Hausdorff distance between the non-zero pixels of two masks (utils/metrics.py:199-222)."""
    from scipy.spatial.distance import directed_hausdorff
    S, G = _to_numpy(S), _to_numpy(G)
    if S.sum() == 0 and G.sum() == 0:
        return 0
    if S.sum() == 0 or G.sum() == 0:
        return np.inf
    Sc, Gc = np.column_stack(np.nonzero(S > 0)), np.column_stack(np.nonzero(G > 0))
    return max(directed_hausdorff(Sc, Gc)[0], directed_hausdorff(Gc, Sc)[0])


def _weighted_pair_hausdorff(A, B, C):
    """sum over the objects a of A of (area_a / total) * Hausdorff(a, partner in B), or to the nearest object of B
    when a overlaps none (utils/metrics.py:247-261)."""
    partner = _partner(C)
    area = C.sum(1)
    total = area[1:].sum()
    nB = C.shape[1] - 1
    acc = 0.0
    for a in range(1, C.shape[0]):
        Ai = A == a
        if partner[a] > 0:
            acc += area[a] / total * hausdorff(Ai, B == partner[a])
        elif nB > 0:
            acc += area[a] / total * min(hausdorff(Ai, B == b) for b in range(1, nB + 1))
    return acc


def object_hausdorff(S, G):
    """Object-level Hausdorff distance (utils/metrics.py:225-281)."""
    S, G = label(_to_numpy(S)), label(_to_numpy(G))
    C, nS, nG = _contingency(S, G)
    return (_weighted_pair_hausdorff(S, G, C) + _weighted_pair_hausdorff(G, S, C.T)) / 2
