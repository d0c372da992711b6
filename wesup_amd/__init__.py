"""MI355X-native WESUP training-step hot path (see README.md / DESIGN.md)."""
import os

# One hardware queue per stream of the step even when a torch.distributed process group adds its own streams
# (DESIGN.md 7); only effective when the package is imported before the HIP runtime starts, harmless otherwise.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '6')
