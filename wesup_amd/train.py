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
    if v.startswith('(') and v.endswith(')'):                 # tuples such as multiscale_range (0.3,0.4)
        return tuple(_parse_value(x.strip()) for x in v[1:-1].split(',') if x.strip())
    return {'True': True, 'False': False, 'None': None}.get(v, v)


def parse_cli_kwargs(tokens):
    """``--key value``, ``--key=value`` and bare ``--flag`` (= True) -> kwargs, the way ``fire.Fire(fit)`` reads them
    (train.py:32).  A bare flag never consumes the option that follows it."""
    kw, i = {}, 0
    while i < len(tokens):
        tok = tokens[i]
        i += 1
        if not tok.startswith('--'):
            raise SystemExit(f'unexpected argument {tok!r} (options look like --key value)')
        key, eq, val = tok[2:].partition('=')
        if not eq:
            if i < len(tokens) and not tokens[i].startswith('--'):
                val = tokens[i]
                i += 1
            else:
                val = 'True'
        kw[key.replace('-', '_')] = _parse_value(val)
    return kw


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('dataset_path')
    ap.add_argument('--model', default='wesup')
    args, rest = ap.parse_known_args()
    fit(args.dataset_path, model=args.model, **parse_cli_kwargs(rest))
