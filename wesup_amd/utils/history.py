"""Epoch bookkeeping of the trainer.

Contract shared with the reference's ``utils/history.py`` ("Metrics / logging"): the trainer calls
``start_new_epoch(lr)``, ``train()`` / ``eval()``, ``step(metrics)`` once per iteration, ``log()`` once per phase,
``save()`` once per epoch and ``report()`` at the end; ``history[key]`` is the list of this epoch's values of a metric;
validation metrics carry the ``val_`` prefix; ``history.csv`` has one row per epoch with the epoch means in sorted-key
order followed by ``lr``.  Everything else here is this package's own (no pandas).
"""
import csv
from collections import defaultdict
from pathlib import Path
from statistics import fmean

from . import underline

VAL_PREFIX = 'val_'
_NOT_REPORTED = ('lr', 'loss', VAL_PREFIX + 'loss')


def _fmt(pairs, template):
    return ', '.join(template.format(name, value) for name, value in pairs)


class HistoryTracker:
    def __init__(self, save_path=None):
        self.save_path = save_path
        self.learning_rate = None
        self.is_train = True
        self.history = defaultdict(list)

    # ---- phase / epoch switches
    def start_new_epoch(self, lr):
        self.learning_rate = lr
        self.history.clear()

    def train(self):
        self.is_train = True

    def eval(self):
        self.is_train = False

    # ---- per iteration
    def step(self, metrics):
        prefix = '' if self.is_train else VAL_PREFIX
        named = [(prefix + key, value) for key, value in metrics.items()]
        for key, value in named:
            self.history[key].append(value)
        return _fmt(named, '{} = {:.4f}')

    # ---- per phase / per epoch
    def epoch_means(self, phase=None):
        """{key: mean over this epoch} in sorted key order; phase 'train' / 'val' keeps that phase's keys only."""
        keys = sorted(self.history)
        if phase is not None:
            keys = [k for k in keys if k.startswith(VAL_PREFIX) == (phase == 'val')]
        return {k: (fmean(self.history[k]) if self.history[k] else 0) for k in keys}

    def log(self):
        means = self.epoch_means('train' if self.is_train else 'val')
        return _fmt(means.items(), 'average {} = {:.4f}').capitalize()

    def save(self):
        if self.save_path is None:
            raise RuntimeError('cannot save history without setting save_path.')
        means = self.epoch_means()
        path = Path(self.save_path)
        first = not path.exists()
        with path.open('a', newline='') as fp:
            out = csv.writer(fp)
            if first:
                out.writerow([*means, 'lr'])
            out.writerow([*means.values(), self.learning_rate])

    def report(self, last_n_epochs=5):
        with open(self.save_path, newline='') as fp:
            tail = list(csv.DictReader(fp))[-last_n_epochs:]
        lines = []
        for column in (tail[0] if tail else {}):
            if column in _NOT_REPORTED:
                continue
            values = [float(row[column]) for row in tail if row[column] not in ('', None)]
            lines.append(f'{column:20s} {fmean(values) if values else float("nan"):.4f}')
        return underline('\nTraining Summary (Avg over last 5 epochs)', style='=') + '\n' + '\n'.join(lines)
