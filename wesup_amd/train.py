"""Training module (mirror of the reference's train.py:14-32; `fire` is replaced by argparse).

  python -m wesup_amd.train synthetic:480:480:24:16 --epochs 1 --batch_size 4 [--smoke]
"""
import argparse
import logging
from shutil import rmtree

from .models import initialize_trainer
from .utils.metrics import accuracy
from .utils.metrics import dice


def fit(dataset_path, model='wesup', **kwargs):
    logger = logging.getLogger('Train')
    logger.setLevel(logging.DEBUG)
    if not logger.handlers:
        logger.addHandler(logging.StreamHandler())
    trainer = initialize_trainer(model, logger=logger, **kwargs)
    try:
        trainer.train(dataset_path, metrics=[accuracy, dice], **kwargs)
    finally:
        if kwargs.get('smoke'):
            rmtree(trainer.record_dir, ignore_errors=True)
    return trainer


def _parse_value(v):
    for cast in (int, float):
        try:
            return cast(v)
        except ValueError:
            pass
    return {'True': True, 'False': False}.get(v, v)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('dataset_path')
    ap.add_argument('--model', default='wesup')
    args, rest = ap.parse_known_args()
    kw = {}
    it = iter(rest)
    for tok in it:
        if tok.startswith('--'):
            key, _, val = tok[2:].partition('=')
            if not val:
                nxt = next(it, None)
                val = 'True' if nxt is None or nxt.startswith('--') else nxt
            kw[key] = _parse_value(val)
    fit(args.dataset_path, model=args.model, **kw)
