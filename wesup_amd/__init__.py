"""MI355X-native WESUP training-step hot path (see README.md / DESIGN.md)."""
import os
import sys
import warnings

# One hardware queue per stream of the step even when a torch.distributed process group adds its own streams
# (DESIGN.md 7).  HIP reads the variable once, when its runtime starts: importing this package first (as bench.py and the
# trainer's CLI do) is enough; an embedder that already touched the GPU keeps the default of 4 queues, and with a process
# group alive two of the step's three streams then share a queue (+1.2 ms per step) -- say so instead of losing it quietly.
if 'GPU_MAX_HW_QUEUES' not in os.environ:
    os.environ['GPU_MAX_HW_QUEUES'] = '6'
    _torch = sys.modules.get('torch')
    if _torch is not None and getattr(_torch, 'cuda', None) is not None and _torch.cuda.is_initialized():
        warnings.warn('wesup_amd was imported after the HIP runtime started: GPU_MAX_HW_QUEUES=6 cannot take effect any more. '
                      'Import wesup_amd (or export GPU_MAX_HW_QUEUES=6) before the first GPU call when you train data-parallel.',
                      RuntimeWarning, stacklevel=2)
