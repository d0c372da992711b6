"""GlaS evaluation: small-region post-processing of predicted masks, the challenge metrics over a directory of
predictions, and the test-set driver (reference scripts/evaluate_glas.py:29-98 and test_glas.py:13-61;
SURVEY.md 8(f) row 4).

  python -m wesup_amd.evaluate PRED_ROOT --gt-root ~/data/GLAS_all        # post-process + score testA / testB
  python -m wesup_amd.evaluate --test -c CKPT --scales 0.6,0.55,0.5,0.45,0.4 --data-root ~/data/GLAS_all

Evaluation-time code on the CPU, as in the reference (numpy / scipy; the training step does not touch it)."""
import argparse
import csv
from pathlib import Path

import numpy as np

from .utils import metrics as M

MIN_REGION = 2000        # pixels (scripts/evaluate_glas.py:33,39)


def remove_small_regions(pred, min_size=MIN_REGION):
    """Post-processing of a binary prediction (scripts/evaluate_glas.py:29-43): connected foreground regions smaller
    than ``min_size`` pixels are erased, then connected background regions (holes) smaller than ``min_size`` are
    filled -- the second pass sees the result of the first.  8-connected components (skimage.measure.label's default
    for 2-D input).  The reference walks the regions one boolean mask at a time; here one ``bincount`` of the label map
    gives every region's area and one table lookup rewrites the mask."""
    out = (np.asarray(pred) != 0).astype(np.float64)
    for value in (1.0, 0.0):                       # erase small foreground regions, then fill small holes
        regions = M.label(out == value)
        area = np.bincount(regions.ravel())
        small = area < min_size
        small[0] = False                           # label 0 is "everything else", not a region of this pass
        out[small[regions]] = 1.0 - value
    return out


def _read_mask(path):
    from PIL import Image
    a = np.asarray(Image.open(path))
    return a[..., 0] if a.ndim == 3 else a


def score(predictions, gts, binarize_gt=False):
    """Per-image rows and means of accuracy, Dice and the three object-level challenge metrics
    (scripts/evaluate_glas.py:46-69).  The reference hands the ground-truth OBJECT maps (ids 0..n) to ``accuracy`` and
    ``dice`` as they are -- for GlaS that compares a {0, 1} prediction with instance ids; kept by default so that the
    numbers are the reference's, ``binarize_gt=True`` scores against ``gt > 0`` instead."""
    rows = []
    for pred, gt in zip(predictions, gts):
        gt = np.asarray(gt)
        flat = (gt > 0).astype(pred.dtype) if binarize_gt else gt
        rows.append({'accuracy': float(M.accuracy(pred, flat)),
                     'dice': float(M.dice(pred, flat)),
                     'detection_f1': float(M.detection_f1(pred, gt)),
                     'object_dice': float(M.object_dice(pred, gt)),
                     'object_hausdorff': float(M.object_hausdorff(pred, gt)) if pred.any() and gt.any() else float('nan')})
    means = {k: float(np.nanmean([r[k] for r in rows])) for k in (rows[0] if rows else {})}
    return rows, means


def evaluate_split(pred_dir, gt_dir, new_pred_dir=None, csv_path=None, min_size=MIN_REGION, log=print):
    """One test split: read predictions (0/255 images) and ground-truth object maps in sorted order, post-process,
    optionally save the new predictions and the per-image csv (columns of scripts/evaluate_glas.py:62-66)."""
    from PIL import Image
    exts = ('*.bmp', '*.png')
    pred_paths = sorted(p for e in exts for p in Path(pred_dir).glob(e))
    gt_paths = sorted(p for e in exts for p in Path(gt_dir).glob(e))
    if len(pred_paths) != len(gt_paths):
        raise ValueError(f'{len(pred_paths)} predictions in {pred_dir} but {len(gt_paths)} masks in {gt_dir}')
    predictions = [remove_small_regions(_read_mask(p) / 255, min_size) for p in pred_paths]
    gts = [_read_mask(p) for p in gt_paths]
    if new_pred_dir is not None:
        Path(new_pred_dir).mkdir(parents=True, exist_ok=True)
        for pred, path in zip(predictions, pred_paths):
            Image.fromarray((pred * 255).astype('uint8')).save(Path(new_pred_dir) / path.name)
    rows, means = score(predictions, gts)
    for name, key in (('Accuracy', 'accuracy'), ('Dice', 'dice'), ('Detection F1', 'detection_f1'),
                      ('Object Dice', 'object_dice'), ('Object Hausdorff', 'object_hausdorff')):
        log(f'{name}: {means.get(key, float("nan"))}')
    if csv_path is not None:
        with open(csv_path, 'w', newline='') as fp:
            out = csv.writer(fp)
            out.writerow(['', 'detection_f1', 'object_dice', 'object_hausdorff'])
            for path, r in zip(pred_paths, rows):
                out.writerow([path.name, r['detection_f1'], r['object_dice'], r['object_hausdorff']])
    return rows, means


def evaluate_glas(pred_root, gt_root='~/data/GLAS_all', min_size=MIN_REGION, log=print):
    """scripts/evaluate_glas.py: testA and testB under ``pred_root`` against ``gt_root/<split>/masks``; post-processed
    predictions go to ``<pred_root>-new/<split>``, per-image metrics to ``pred_root/<split>.csv``."""
    pred_root, gt_root = Path(pred_root).expanduser(), Path(gt_root).expanduser()
    new_root = pred_root.parent / (pred_root.name + '-new')
    result = {}
    for split, title in (('testA', 'Test A'), ('testB', '\nTest B')):
        if not (pred_root / split).exists():
            continue
        log(title)
        result[split] = evaluate_split(pred_root / split, gt_root / split / 'masks', new_root / split,
                                       pred_root / f'{split}.csv', min_size, log)[1]
    return result


def test(ckpt_path, model_type='wesup', input_size=None, scales=(0.5,), device='cuda', data_root='~/data/GLAS_all'):
    """test_glas.py:13-38: load a checkpoint, predict test sets A and B into ``<record_dir>/results`` (fixed input
    size) or ``<record_dir>/results-<n>scale`` (multi-scale)."""
    from .infer import infer
    from .models import initialize_trainer
    ckpt_path = Path(ckpt_path)
    trainer = initialize_trainer(model_type, device=device)
    trainer.load_checkpoint(ckpt_path)
    record_dir = ckpt_path.parent.parent
    results_dir = record_dir / ('results' if input_size is not None else f'results-{len(scales)}scale')
    results_dir.mkdir(parents=True, exist_ok=True)
    data_root = Path(data_root).expanduser()
    for split in ('testA', 'testB'):
        if (data_root / split).exists():
            print(f'\nTesting on test set {split[-1]} ...')
            infer(trainer, data_root / split, results_dir / split, input_size, scales, device=device)
    return results_dir


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('pred_root', nargs='?')
    ap.add_argument('--gt-root', default='~/data/GLAS_all')
    ap.add_argument('--min-size', type=int, default=MIN_REGION)
    ap.add_argument('--test', action='store_true', help='run the test-set driver (test_glas.py) instead of scoring')
    ap.add_argument('-m', '--model', default='wesup')
    ap.add_argument('-c', '--checkpoint')
    ap.add_argument('--input-size')
    ap.add_argument('--scales', default='0.6,0.55,0.5,0.45,0.4')          # test_glas.py:48
    ap.add_argument('--data-root', default='~/data/GLAS_all')
    ap.add_argument('-d', '--device', default='cuda')
    a = ap.parse_args(argv)
    if a.test:
        size = [int(s) for s in a.input_size.split(',')] if a.input_size else None
        test(a.checkpoint, a.model, size, tuple(float(s) for s in a.scales.split(',')), a.device, a.data_root)
    else:
        evaluate_glas(a.pred_root, a.gt_root, a.min_size)


if __name__ == '__main__':
    main()
