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
    """train.py:14-27 of the reference.  Under a multi-process launcher (``python -m torch.distributed.run --nproc-per-node N
    -m wesup_amd.train DATA ...``: WORLD_SIZE / RANK / LOCAL_RANK in the environment) every process takes the GPU of its
    local rank, joins the RCCL process group and trains data-parallel: batches sharded by image, gradients all-reduced
    during backward (wesup_amd/ddp.py)."""
    import os
    logger = logging.getLogger('Train')
    logger.setLevel(logging.DEBUG)
    if not logger.handlers:
        logger.addHandler(logging.StreamHandler())
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        import torch
        import torch.distributed as dist
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        kwargs.setdefault('device', f'cuda:{local_rank}')
        if not dist.is_initialized():
            torch.cuda.set_device(local_rank)
            dist.init_process_group(kwargs.pop('dist_backend', 'nccl'))
    trainer = initialize_trainer(model, logger=logger, **kwargs)
    if world > 1:
        trainer.enable_data_parallel()
    try:
        trainer.train(dataset_path, metrics=[accuracy, dice], **kwargs)
    finally:
        if kwargs.get('smoke') and trainer.rank == 0 and trainer.record_dir is not None:
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
    kw = parse_cli_kwargs(rest)
    # The training SCRIPT (not the library) opts into the two process-wide / trust-based conveniences of the step runner: the
    # reference trains multi-scale (utils/data.py:98-101: a new shape almost every step), where they pay (wesup_amd/runner.py)
    kw.setdefault('gc_freeze', True)
    kw.setdefault('trust_first_recording_after', 1)
    fit(args.dataset_path, model=args.model, **kw)
