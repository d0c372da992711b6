"""Run-directory bookkeeping (minimal equivalent of utils/record.py:16-68; plots are out of scope)."""
import json
import os
import time
from pathlib import Path


def prepare_record_dir():
    root = Path(os.environ.get('RECORD_ROOT', str(Path.home() / 'records')))      # utils/record.py:22-24
    record_dir = root / f"{time.strftime('%Y%m%d-%H%M%S')}-{os.getpid()}"
    (record_dir / 'checkpoints').mkdir(parents=True, exist_ok=True)
    return record_dir


def save_params(record_dir, params):
    with open(Path(record_dir) / 'params.json', 'w') as fp:
        json.dump(params, fp, indent=4)


def copy_source_files(record_dir):
    """The reference snapshots its sources here (utils/record.py:55-68); not needed for the hot path."""


def plot_learning_curves(history_path):
    """matplotlib curves in the reference (utils/record.py:71-107); out of scope."""
