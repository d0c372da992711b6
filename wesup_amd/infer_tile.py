"""Window-based inference on images larger than the network input (reference infer_tile.py:23-181 and
pixel_infer_tile.py:41-60).

Same strategy as the reference: ``ceil(H / patch) x ceil(W / patch)`` windows whose top-left corners are spread evenly
with ``np.linspace`` (so neighbouring windows overlap when the size is not a multiple of the patch), every window goes
through the model on its own, and overlapping predictions are merged by the reference's running average.  The three
window functions are numpy and pinned to the reference's own outputs (tests/golden/tiles.npz); the per-window forward
is the HIP path (``trainer.preprocess`` with the GPU SLIC -> ``WESUP.forward`` -> ``postprocess``, or
``WESUPPixelInference`` for the pixel-wise variant)."""
import argparse
import math
from itertools import product
from pathlib import Path

import numpy as np
import torch

from .models import initialize_trainer


def _get_top_left_coordinates(height, width, patch_size):
    """Top-left corners of the windows (infer_tile.py:23-31)."""
    n_h = math.ceil(height / patch_size)
    n_w = math.ceil(width / patch_size)
    tops = np.linspace(0, height - patch_size, n_h, dtype=int)
    lefts = np.linspace(0, width - patch_size, n_w, dtype=int)
    return product(tops, lefts)


def divide_image_to_patches(img, patch_size):
    """(H, W, 3) uint8 image -> (N, patch_size, patch_size, 3) possibly overlapping windows (infer_tile.py:34-57)."""
    assert len(img.shape) == 3 and img.shape[-1] == 3
    height, width, _ = img.shape
    if height < patch_size or width < patch_size:
        raise ValueError(f'image {height}x{width} is smaller than the patch size {patch_size}')
    patches = [img[top:top + patch_size, left:left + patch_size]
               for top, left in _get_top_left_coordinates(height, width, patch_size)]
    return np.array(patches).astype('uint8')


def combine_patches_to_image(patches, target_height, target_width):
    """Merge window predictions (N, h, w[, C]) into one (H, W[, C]) map, averaging where windows overlap
    (infer_tile.py:60-91: a running mean kept with a per-pixel overlap count)."""
    counter = 0
    patch_size = patches.shape[1]
    if len(patches.shape) == 3:          # channel dimension is missing
        patches = np.expand_dims(patches, -1)
    combined = np.zeros((target_height, target_width, patches.shape[-1] + 1))
    for top, left in _get_top_left_coordinates(target_height, target_width, patch_size):
        patch = combined[top:top + patch_size, left:left + patch_size, :-1]
        overlaps = combined[top:top + patch_size, left:left + patch_size, -1:]
        patch = (patch * overlaps + patches[counter]) / (overlaps + 1)
        combined[top:top + patch_size, left:left + patch_size, :-1] = patch
        overlaps += 1.
        counter += 1
    return np.squeeze(combined[..., :-1])


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
