"""Data contract of the training step (reference: utils/data.py:135-152,459-512).

Real dataset readers / augmentation are host-side I/O outside the hot path
(8(f) rank 2).  What the trainer needs is the tensor contract:
an item is (img f32 (3,H,W) in [0,1], pixel_mask (C,H,W) one-hot, point_mask (C,H,W)
radius-0 dots, segments (H,W) int32 label map).  ``SyntheticGlasDataset`` produces
GlaS-shaped items of that contract; ``get_dataset`` accepts 'synthetic:H:W:g:n'."""
import torch

from .. import synth


class SyntheticGlasDataset(torch.utils.data.Dataset):
    def __init__(self, H=480, W=480, g=24, n=16, seed=0, frac=0.2, with_points=True):
        self.H, self.W, self.g, self.n, self.seed, self.frac = H, W, g, n, seed, frac
        self.with_points = with_points

    def __len__(self):
        return self.n

    def summary(self, logger=None):
        msg = f'SyntheticGlasDataset: {self.n} samples of {self.H}x{self.W}, {self.g * self.g} superpixels'
        (logger.info if logger else print)(msg)

    def __getitem__(self, i):
        s = self.seed * 100003 + i
        img = torch.from_numpy(synth.synth_image(s, self.H, self.W))
        seg = torch.from_numpy(synth.voronoi_labels(s, self.H, self.W, self.g))
        pix = torch.from_numpy(synth.pixel_mask(s, self.H, self.W)).long()
        if not self.with_points:
            return img, pix, torch.zeros(0), seg
        pts = torch.from_numpy(synth.point_mask(s, seg.numpy(), self.frac, 2)).long()
        return img, pix, pts, seg


def get_dataset(root_dir, train=True, proportion=1.0, multiscale_range=None, rescale_factor=None):
    root = str(root_dir)
    if 'synthetic:' in root:
        spec = root[root.index('synthetic:'):].split('/')[0].split(':')[1:]
        H, W, g, n = (int(v) for v in (spec + ['480', '480', '24', '16'][len(spec):])[:4])
        return SyntheticGlasDataset(H, W, g, max(1, int(n * proportion)), seed=0 if train else 1)
    raise NotImplementedError('dataset readers/augmentation are outside the MI355X hot path (SURVEY.md 8(f) rank 2); '
                              "use 'synthetic:H:W:g:n' or feed tensors to trainer.train_one_iteration directly")
