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
