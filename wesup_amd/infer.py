"""Superpixel inference on a directory of images (reference infer.py:24-153).

Same flow as the reference: resize (fixed ``input_size`` or every factor in ``scales``), ``trainer.preprocess`` (GPU
SLIC here) -> model forward -> ``postprocess`` (round) -> nearest upsample to the original size; multi-scale
predictions are averaged and rounded, then opened with the reference's 9x9 cross.  Everything stays on the GPU until
the final mask; the opening runs on the CPU (scipy.ndimage instead of skimage.morphology, which is absent: parity
of that step unpinned).  ``evaluate_predictions`` scores masks with the challenge metrics (utils/metrics.py)."""
import argparse
from math import ceil
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

from .models import initialize_trainer
from .utils import metrics as M
from .utils.data import SegmentationDataset


def predict_single_image(trainer, img, mask, output_size):
    """img (1,3,h,w), mask (1,C,h,w) or a 0-dim tensor -> (1,1,H,W) {0,1} at ``output_size`` (infer.py:24-35)."""
    data = (img, mask.long()) if mask.dim() == 4 else (img,)
    input_, target = trainer.preprocess(*data)
    with torch.no_grad():
        pred = trainer.model(input_)
    pred, _ = trainer.postprocess(pred, target)
    pred = pred.float().unsqueeze(0)
    return F.interpolate(pred, size=output_size, mode='nearest')


def _cross(size=9):
    """The reference's structuring element (infer.py:84-90): a cross through index (size+1)/2 -- one past the centre."""
    assert size % 2 == 1
    selem = np.zeros((size, size))
    center = int((size + 1) / 2)
    selem[center, :] = 1
    selem[:, center] = 1
    return selem


def predict(trainer, dataset, input_size=None, scales=(0.5,), device='cuda'):
    """Predict every image of ``dataset`` (raw items of utils.data.SegmentationDataset).  Returns a list of (H,W) masks."""
    from scipy import ndimage
    predictions = []
    for i in range(len(dataset)):
        raw = dataset[i]
        img, mask = dataset.to_reference_item(raw)
        img = img.unsqueeze(0).to(device)
        mask = mask.unsqueeze(0).to(device).float() if mask.dim() == 3 else mask
        orig_size = (img.size(2), img.size(3))

        def resized(size):
            im = F.interpolate(img, size=size, mode='bilinear')
            mk = F.interpolate(mask, size=size, mode='nearest') if mask.dim() == 4 else mask
            return im, mk

        if input_size is not None:
            prediction = predict_single_image(trainer, *resized(input_size), orig_size)
        else:
            multi = []
            for scale in scales:
                # (the reference rescales the already rescaled image at the second scale, infer.py:74-76; every scale
                #  starts from the original here)
                target_size = [ceil(s * scale) for s in orig_size]
                multi.append(predict_single_image(trainer, *resized(target_size), orig_size))
            prediction = torch.cat(multi).mean(dim=0).round()
        prediction = prediction.squeeze().cpu().numpy()
        if input_size is None and len(scales) > 1:
            prediction = ndimage.grey_opening(prediction, footprint=_cross(9))
        predictions.append(prediction)
    return predictions


def save_predictions(predictions, dataset, output_dir='predictions'):
    from PIL import Image
    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    for pred, img_path in zip(predictions, dataset.img_paths):
        Image.fromarray(pred.astype('uint8') * 255).save(output_dir / f'{img_path.stem}.png')


def evaluate_predictions(predictions, dataset):
    """Challenge metrics of the predictions against the dataset's masks (scripts/evaluate_glas.py:29-69)."""
    rows = []
    for i, pred in enumerate(predictions):
        gt = dataset[i][1].numpy()
        gt = (gt == 1).astype(np.uint8)
        rows.append({'accuracy': M.accuracy(pred, gt), 'dice': M.dice(pred, gt), 'detection_f1': M.detection_f1(pred, gt),
                     'object_dice': M.object_dice(pred, gt),
                     'object_hausdorff': M.object_hausdorff(pred, gt) if pred.any() and gt.any() else float('nan')})
    keys = rows[0].keys() if rows else []
    return {k: float(np.nanmean([r[k] for r in rows])) for k in keys}, rows


def infer(trainer, data_dir, output_dir=None, input_size=None, scales=(0.5,), device='cuda'):
    trainer.model.eval()
    dataset = SegmentationDataset(data_dir, train=False)
    predictions = predict(trainer, dataset, input_size=input_size, scales=scales, device=device)
    if output_dir is not None:
        save_predictions(predictions, dataset, output_dir)
    return predictions


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('data_dir')
    ap.add_argument('--model-type', default='wesup')
    ap.add_argument('--checkpoint')
    ap.add_argument('--output-dir')
    ap.add_argument('--input-size', type=int, nargs=2)
    ap.add_argument('--scales', type=float, nargs='+', default=[0.5])
    ap.add_argument('--device', default='cuda')
    a = ap.parse_args(argv)
    output_dir = a.output_dir
    if output_dir is None and a.checkpoint is not None:
        output_dir = Path(a.checkpoint).parent.parent / 'results'
    trainer = initialize_trainer(a.model_type, device=a.device)
    if a.checkpoint is not None:
        trainer.load_checkpoint(a.checkpoint)
    infer(trainer, a.data_dir, output_dir, input_size=a.input_size, scales=tuple(a.scales), device=a.device)


if __name__ == '__main__':
    main()
