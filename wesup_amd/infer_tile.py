"""Window-based inference on images larger than the network input (reference infer_tile.py:23-181 and
pixel_infer_tile.py:41-60).

Same strategy as the reference: ``ceil(H / patch) x ceil(W / patch)`` windows whose top-left corners are spread evenly
with ``np.linspace`` (so neighbouring windows overlap when the size is not a multiple of the patch), every window goes
through the model on its own, and overlapping predictions are merged by the mean over the covering windows.  The
window functions are numpy and pinned to the reference's own outputs (tests/golden/tiles.npz); the per-window forward
is the HIP path (``trainer.preprocess`` with the GPU SLIC -> ``WESUP.forward`` -> ``postprocess``, or
``WESUPPixelInference`` for the pixel-wise variant)."""
import argparse
import math
from pathlib import Path

import numpy as np
import torch

from .models import initialize_trainer


def window_grid(height, width, patch_size):
    """Rows and columns of the window lattice: ``ceil(size / patch)`` windows per axis, first window at 0, last window
    flush with the far border, the others spread evenly in between (positions truncated to integers, which is what
    ``np.linspace(..., dtype=int)`` does in infer_tile.py:26-29)."""
    def axis(size):
        if size < patch_size:
            raise ValueError(f'image {height}x{width} is smaller than the patch size {patch_size}')
        return np.linspace(0, size - patch_size, math.ceil(size / patch_size), dtype=int)
    return axis(height), axis(width)


def _get_top_left_coordinates(height, width, patch_size):
    """(top, left) of every window, row-major (the order of infer_tile.py:23-31)."""
    tops, lefts = window_grid(height, width, patch_size)
    return [(int(t), int(l)) for t in tops for l in lefts]


def divide_image_to_patches(img, patch_size):
    """(H, W, 3) uint8 image -> (N, patch_size, patch_size, 3) possibly overlapping windows (infer_tile.py:34-57)."""
    if img.ndim != 3 or img.shape[-1] != 3:
        raise AssertionError('expected an (H, W, 3) image')            # the reference asserts (infer_tile.py:46)
    tops, lefts = window_grid(img.shape[0], img.shape[1], patch_size)
    span = np.arange(patch_size)
    rows = (tops[:, None] + span)[:, None, :, None]                    # (n_h, 1, p, 1)
    cols = (lefts[:, None] + span)[None, :, None, :]                   # (1, n_w, 1, p)
    windows = img[rows, cols]                                          # one gather: (n_h, n_w, p, p, 3)
    return windows.reshape(-1, patch_size, patch_size, 3).astype(np.uint8)


def combine_patches_to_image(patches, target_height, target_width):
    """Merge window predictions (N, p, p[, C]) into one (H, W[, C]) map: the value of a pixel is the mean over the
    windows that cover it (the reference keeps it as a running mean with a per-pixel overlap count,
    infer_tile.py:60-91; here: per-pixel sum and count, divided once -- equal up to float rounding)."""
    patches = np.asarray(patches, dtype=np.float64)
    flat = patches.ndim == 3
    if flat:
        patches = patches[..., None]
    p = patches.shape[1]
    total = np.zeros((target_height, target_width, patches.shape[-1]))
    cover = np.zeros((target_height, target_width, 1))
    for window, (top, left) in zip(patches, _get_top_left_coordinates(target_height, target_width, p)):
        total[top:top + p, left:left + p] += window
        cover[top:top + p, left:left + p] += 1.0
    merged = total / cover                                              # every pixel lies in at least one window
    return merged[..., 0] if flat else np.squeeze(merged)


def _to_tensor(patch, device):
    """uint8 (h, w, 3) -> float (1, 3, h, w) in [0, 1] (torchvision's to_tensor, infer_tile.py:111)."""
    return torch.from_numpy(np.ascontiguousarray(patch)).to(device).permute(2, 0, 1).float().div_(255.).unsqueeze(0)


def predict_array(trainer, img, patch_size, device='cuda'):
    """Window-based superpixel prediction of one (H, W, 3) uint8 image -> (H, W) float map (infer_tile.py:94-119)."""
    patches = divide_image_to_patches(img, patch_size)
    predictions = []
    with torch.no_grad():
        for patch in patches:
            input_, _ = trainer.preprocess(_to_tensor(patch, device))
            prediction = trainer.postprocess(trainer.model(input_))
            predictions.append(prediction.detach().cpu().numpy()[..., np.newaxis])
    predictions = np.concatenate(predictions)
    return combine_patches_to_image(predictions, img.shape[0], img.shape[1])


def pixel_predict_array(model, img, patch_size, device='cuda'):
    """Window-based pixel-wise prediction with WESUPPixelInference -> (H, W) class-1 probability
    (pixel_infer_tile.py:41-60; the caller rounds)."""
    patches = divide_image_to_patches(img, patch_size)
    predictions = []
    with torch.no_grad():
        for patch in patches:
            pred = model(_to_tensor(patch, device))
            predictions.append(np.expand_dims(pred.detach().cpu().numpy()[..., 1], 0))
    predictions = np.concatenate(predictions)
    return combine_patches_to_image(predictions, img.shape[0], img.shape[1])


def predict(trainer, img_path, patch_size, device='cuda'):
    from PIL import Image
    return predict_array(trainer, np.asarray(Image.open(img_path).convert('RGB')), patch_size, device=device)


def save_predictions(predictions, img_paths, output_dir='predictions'):
    from PIL import Image
    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    for pred, img_path in zip(predictions, img_paths):
        Image.fromarray(pred.astype('uint8') * 255).save(output_dir / Path(img_path).name)


def infer(trainer, data_dir, patch_size, output_dir=None, device='cuda'):
    """Window-based inference on ``data_dir/images`` (infer_tile.py:143-162)."""
    trainer.model.eval()
    data_dir = Path(data_dir).expanduser()
    img_paths = sorted((data_dir / 'images').iterdir())
    predictions = [predict(trainer, p, patch_size, device=device) for p in img_paths]
    if output_dir is not None:
        save_predictions(predictions, img_paths, output_dir)
    return predictions


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('data_dir')
    ap.add_argument('--model-type', default='wesup')
    ap.add_argument('--patch-size', type=int, default=464)          # infer_tile.py:165
    ap.add_argument('--checkpoint')
    ap.add_argument('--output-dir')
    ap.add_argument('--device', default='cuda')
    a = ap.parse_args(argv)
    output_dir = a.output_dir
    if output_dir is None and a.checkpoint is not None:
        output_dir = Path(a.checkpoint).expanduser().parent.parent / 'results'
    trainer = initialize_trainer(a.model_type, device=a.device)
    if a.checkpoint is not None:
        trainer.load_checkpoint(a.checkpoint)
    infer(trainer, a.data_dir, a.patch_size, output_dir, device=a.device)


if __name__ == '__main__':
    main()
