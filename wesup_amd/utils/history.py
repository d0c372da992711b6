"""Per-epoch metric bookkeeping (mirror of utils/history.py:11-81; pandas-free)."""
import csv
import os
from collections import defaultdict

from . import underline


class HistoryTracker:
    def __init__(self, save_path=None):
        self.history = defaultdict(list)
        self.learning_rate = None
        self.save_path = save_path
        self.is_train = True

    def start_new_epoch(self, lr):
        self.history.clear()
        self.learning_rate = lr

    def train(self):
        self.is_train = True

    def eval(self):
        self.is_train = False

    def step(self, metrics):
        reports = []
        for k, v in metrics.items():
            k = k if self.is_train else f'val_{k}'
            self.history[k].append(v)
            reports.append('{} = {:.4f}'.format(k, v))
        return ', '.join(reports)

    def log(self):
        metrics = {k: (sum(v) / len(v) if v else 0) for k, v in sorted(self.history.items())
                   if k.startswith('val_') != self.is_train}
        return ', '.join('average {} = {:.4f}'.format(n, v) for n, v in metrics.items()).capitalize()

    def save(self):
        if self.save_path is None:
            raise RuntimeError('cannot save history without setting save_path.')
        keys = [k for k, _ in sorted(self.history.items())]
        metrics = [sum(v) / len(v) for _, v in sorted(self.history.items())]
        new = not os.path.exists(self.save_path)
        with open(self.save_path, 'w' if new else 'a') as fp:
            writer = csv.writer(fp)
            if new:
                writer.writerow(keys + ['lr'])
            writer.writerow(metrics + [self.learning_rate])

    def report(self, last_n_epochs=5):
        with open(self.save_path) as fp:
            rows = list(csv.DictReader(fp))
        rows = rows[-last_n_epochs:]
        keys = [k for k in (rows[0].keys() if rows else []) if k not in ('lr', 'loss', 'val_loss')]
        lines = [f'{k:20s} {sum(float(r[k]) for r in rows) / len(rows):.4f}' for k in keys]
        return underline('\nTraining Summary (Avg over last 5 epochs)', style='=') + '\n' + '\n'.join(lines)
