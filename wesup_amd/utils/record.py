"""Run-directory bookkeeping of the trainer (what the reference keeps in utils/record.py:16-107): a fresh directory per
run under ``$RECORD_ROOT`` (default ``~/records``) with ``checkpoints/``, the run's parameters, a snapshot of the
package's sources and, after every epoch, learning curves drawn from ``history.csv``."""
import csv
import json
import os
import shutil
import time
from pathlib import Path

PACKAGE_ROOT = Path(__file__).resolve().parent.parent


def prepare_record_dir():
    root = Path(os.environ.get('RECORD_ROOT') or Path.home() / 'records').expanduser()      # utils/record.py:22-24
    record_dir = root / f"{time.strftime('%Y%m%d-%H%M%S')}-{os.getpid()}"
    (record_dir / 'checkpoints').mkdir(parents=True, exist_ok=True)
    return record_dir


def save_params(record_dir, params):
    """One json per (re)start of the run: params/0.json, params/1.json, ... (utils/record.py:41-52)."""
    params_dir = Path(record_dir) / 'params'
    params_dir.mkdir(parents=True, exist_ok=True)
    n = len(list(params_dir.iterdir()))
    with open(params_dir / f'{n}.json', 'w') as fp:
        json.dump(params, fp, indent=4)


def copy_source_files(record_dir):
    """Snapshot of the code that produced the run (utils/record.py:55-68): the package's Python sources, the HIP
    sources and the C-ABI header -- not the built library."""
    dst = Path(record_dir) / 'source'
    if dst.exists():
        shutil.rmtree(dst)
    keep = {'.py', '.hip', '.hpp', '.h'}
    for path in sorted(PACKAGE_ROOT.rglob('*')):
        if path.is_file() and path.suffix in keep and '__pycache__' not in path.parts:
            out = dst / 'wesup_amd' / path.relative_to(PACKAGE_ROOT)
            out.parent.mkdir(parents=True, exist_ok=True)
            shutil.copyfile(path, out)
    header = PACKAGE_ROOT.parent / 'include' / 'wesup_hip.h'
    if header.exists():
        (dst / 'include').mkdir(parents=True, exist_ok=True)
        shutil.copyfile(header, dst / 'include' / header.name)


def plot_learning_curves(history_path):
    """One figure per metric with its training and validation curve over the epochs, next to ``history.csv`` under
    ``curves/`` (utils/record.py:71-107).  Needs matplotlib; without it the curves are skipped (the csv is the record)."""
    history_path = Path(history_path)
    if not history_path.exists():
        return []
    try:
        import matplotlib
        matplotlib.use('Agg')
        import matplotlib.pyplot as plt
    except Exception:
        return []
    with open(history_path, newline='') as fp:
        rows = list(csv.DictReader(fp))
    if not rows:
        return []
    out_dir = history_path.parent / 'curves'
    out_dir.mkdir(exist_ok=True)
    epochs = list(range(1, len(rows) + 1))
    written = []
    for key in rows[0]:
        if key.startswith('val_') or key == 'lr':
            continue
        fig, ax = plt.subplots(figsize=(5, 3.5))
        ax.plot(epochs, [float(r[key]) if r[key] else float('nan') for r in rows], label=f'train {key}')
        if f'val_{key}' in rows[0]:
            ax.plot(epochs, [float(r[f'val_{key}']) if r[f'val_{key}'] else float('nan') for r in rows], label=f'val {key}')
        ax.set_xlabel('epoch')
        ax.set_title(key)
        ax.legend()
        fig.tight_layout()
        fig.savefig(out_dir / f'{key}.png')
        plt.close(fig)
        written.append(out_dir / f'{key}.png')
    return written
