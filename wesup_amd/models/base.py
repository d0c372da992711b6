"""Mirror of the reference's models/base.py (BaseConfig, BaseTrainer) driving the HIP step.

Kept: names, signatures, config defaults, metric keys, checkpoint dict keys, error behaviour
(NaN loss -> ValueError before backward; RuntimeError inside an iteration is logged and the
batch skipped, models/base.py:202-203,234-237).  Re-done: batches of B images, data-parallel
gradient exchange, and ONE host sync per training iteration instead of four
(`.item()` x3 + isnan, models/wesup.py:88,523-524, models/base.py:202-205)."""
import logging
import math
import os
import time
from abc import ABC, abstractmethod
from collections import defaultdict
from pathlib import Path

import numpy as np
import torch

from ..utils import underline, record, is_empty_tensor
from ..utils.history import HistoryTracker
from ..utils import metrics as M


class BaseConfig:
    """A base model configuration class (models/base.py:16-36)."""

    batch_size = 1
    epochs = 10
    epsilon = 1e-7

    def __str__(self):
        return '\n'.join(f'{attr:<32s}{getattr(self, attr)}' for attr in dir(self) if not attr.startswith('_'))

    def to_dict(self):
        return {attr: getattr(self, attr) for attr in dir(self) if not attr.startswith('_') and attr != 'to_dict'}


class BaseTrainer(ABC):
    """A base trainer class (models/base.py:39-360)."""

    def __init__(self, model, **kwargs):
        self.device = kwargs.get('device', 'cuda' if torch.cuda.is_available() else 'cpu')
        self.model = model.to(self.device)
        self.kwargs = kwargs
        if kwargs.get('logger'):
            self.logger = kwargs.get('logger')
        else:
            self.logger = logging.getLogger('Train')
            self.logger.setLevel(logging.DEBUG)
            if not self.logger.handlers:
                self.logger.addHandler(logging.StreamHandler())
        self.initial_epoch = 1
        self.record_dir = None
        self.tracker = HistoryTracker()
        self.dataloaders = None
        self.optimizer, self.scheduler = None, None
        self.metric_funcs = []
        self.reducer = None
        self._host_buf = None            # pinned staging buffer of the per-step read-back
        self._runner = None
        self.world_size = 1
        self.rank = 0

    # ------------------------------------------------------------------ hooks (reference interface)
    @abstractmethod
    def get_default_dataset(self, root_dir, train=True, proportion=1.0):
        """Get default dataset for training/validation."""

    def get_default_optimizer(self):
        return torch.optim.SGD(self.model.parameters(), lr=1e-3), None

    def preprocess(self, *data):
        return [datum.to(self.device) for datum in data]

    @abstractmethod
    def compute_loss(self, pred, target, metrics=None):
        """Compute objective function."""

    def postprocess(self, pred, target=None):
        if target is not None:
            return pred, target
        return pred

    def post_epoch_hook(self, epoch):
        pass

    # ------------------------------------------------------------------ checkpoints (models/base.py:124-166)
    def load_checkpoint(self, ckpt_path=None):
        if ckpt_path is not None:
            self.record_dir = Path(ckpt_path).parent.parent
            self.logger.info(f'Loading checkpoint from {ckpt_path}.')
            checkpoint = torch.load(ckpt_path, map_location=self.device)
            self.initial_epoch = checkpoint['epoch'] + 1
            self.model.load_state_dict(checkpoint['model_state_dict'])
            if self.optimizer is not None:
                self.optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
            if self.scheduler is not None and 'scheduler_state_dict' in checkpoint:
                self.scheduler.load_state_dict(checkpoint['scheduler_state_dict'])
        else:
            # one record directory per job: rank 0 creates it, the other ranks learn its path
            path = [str(record.prepare_record_dir())] if self.rank == 0 else [None]
            if self.world_size > 1:
                import torch.distributed as dist
                dist.broadcast_object_list(path, src=0)
            self.record_dir = Path(path[0])
            if self.rank == 0:
                record.copy_source_files(self.record_dir)

    def save_checkpoint(self, ckpt_path, **kwargs):
        checkpoint = {
            'model_state_dict': self.model.state_dict(),
            'optimizer_state_dict': self.optimizer.state_dict(),
            **kwargs,
        }
        if self.scheduler is not None:
            checkpoint['scheduler_state_dict'] = self.scheduler.state_dict()
        Path(ckpt_path).parent.mkdir(parents=True, exist_ok=True)
        torch.save(checkpoint, ckpt_path)

    # ------------------------------------------------------------------ data parallel
    def enable_data_parallel(self, group=None, bucket_bytes=16 << 20):
        """One process per GPU: broadcast rank 0's parameters, all-reduce gradients during backward,
        average inside the SGD kernel."""
        import torch.distributed as dist
        from .. import ddp
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        ddp.broadcast_parameters(self.model, 0, group)
        self.reducer = ddp.attach(self.model, group, bucket_bytes)
        self.reducer.force = bool(self.kwargs.get('force_allreduce', False))
        if self.optimizer is not None and hasattr(self.optimizer, 'grad_scale'):
            self.optimizer.grad_scale = 1.0 / self.world_size

    # ------------------------------------------------------------------ one iteration (models/base.py:184-211)
    def _fast_metrics(self):
        names = [f.__name__ for f in (self.metric_funcs or [])]
        return all(n in ('accuracy', 'dice') for n in names), names

    def prefetch_segment_fn(self):
        """Optional callable the input pipeline runs one batch ahead on its copy stream (utils/data.py
        DevicePrefetcher); its result is appended to the data tuple.  None: nothing to precompute."""
        return None

    def _loss_flag(self, loss):
        """Data parallel only: a NaN loss on ONE rank must stop EVERY rank before the weights are touched (its NaN
        gradients reach all ranks through the all-reduce, but only that rank sees a NaN loss).  The ranks agree through
        a MAX all-reduce of a one-element flag, queued right behind the loss kernels on a stream of its own -- neither
        the backward pass on the main stream nor the side branch waits for it, so the rendezvous does not expose rank
        skew in the middle of the step; it comes back with the loss in the step's one read-back."""
        import torch.distributed as dist
        flag = torch.isnan(loss.detach()).float().reshape(1)
        if getattr(self, '_flag_stream', None) is None:
            self._flag_stream = torch.cuda.Stream(device=flag.device)
        aux = self._flag_stream
        aux.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(aux):
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.reducer.group)
        flag.record_stream(aux)
        return flag, aux

    def step_runner(self):
        """The recordable walk of the standard training iteration (wesup_amd/runner.py), or None when it is switched off
        (``native_step=False``) or the model has no engine."""
        if self._runner is None and self.kwargs.get('native_step', True) and hasattr(self.model, 'prefetch_weights'):
            from ..runner import StepRunner
            self._runner = StepRunner(self)
        return self._runner

    def train_one_iteration(self, phase, *data):
        from .. import ops
        runner = self.step_runner()
        if runner is not None:
            parsed = runner.parse(phase, data)
            if parsed is not None:
                return runner.run(parsed)
        if self.reducer is not None:
            self.reducer.reset()             # nothing may be left over from an iteration that raised
        if hasattr(self.model, 'prefetch_weights'):      # weight repacking and superpixel preprocessing go to the side
            self.model.prefetch_weights(train=(phase == 'train'))      # stream: conv1_1 only waits for its own panel
            with self.model.engine.side_stream():
                input_, target = self.preprocess(*data)
        else:
            input_, target = self.preprocess(*data)

        self.optimizer.zero_grad()
        metrics = dict()
        seg = None
        fast, names = self._fast_metrics()
        pixel_mask = target[0] if isinstance(target, (tuple, list)) else None
        has_gt = pixel_mask is not None and torch.is_tensor(pixel_mask) and not is_empty_tensor(pixel_mask)

        with torch.set_grad_enabled(phase == 'train'):
            pred = self.model(input_)
            if fast and has_gt and names:
                seg = ops.seg_metrics(pred.detach().contiguous(), pixel_mask.to(torch.uint8).contiguous())
            if phase == 'train':
                loss = self.compute_loss(pred, target, metrics=metrics)
                # The ONE host sync of the step.  The copy to pinned memory is queued behind the loss kernels, the
                # backward pass is queued behind it, and only then does the host wait -- for the copy, i.e. for the
                # forward pass -- so the GPU does not idle while the host wakes up and walks the backward schedule.
                # A NaN loss still raises before the weights are touched (models/base.py:202-203): only the gradient
                # buffers have been written by then.
                flag = self._loss_flag(loss) if (self.reducer is not None and self.world_size > 1) else None
                pending = self._stage_read_back(loss, metrics, seg, flag)
                if self.reducer is not None and self.reducer.profile:
                    self.reducer.t_backward = torch.cuda.Event(enable_timing=True)
                    self.reducer.t_backward.record()
                loss.backward()
                if self.reducer is not None:
                    self.reducer.finish()
                host = self._finish_read_back(pending, metrics)
                if math.isnan(host['loss']) or host.get('nan_anywhere', 0.0) > 0:
                    raise ValueError('Loss is nan!')
                metrics['loss'] = host['loss']
                self.optimizer.step()
            else:
                host = self._read_back(None, metrics, seg)

        if fast and has_gt and names:
            sums = host['seg']
            ev = {}
            if 'accuracy' in names:
                ev['accuracy'] = M.accuracy_from_sums(sums, pred.shape[-1] * pred.shape[-2])
            if 'dice' in names:
                ev['dice'] = M.dice_from_sums(sums)
        elif has_gt:
            pred_, target_ = self.postprocess(pred.detach(), target)
            ev = self.evaluate(pred_, target_)
        else:
            ev = {}
        self.tracker.step({**metrics, **ev})

    def _read_back(self, loss, metrics, seg):
        """Bring loss, the per-image loss terms and the segmentation sums to the host in one copy."""
        return self._finish_read_back(self._stage_read_back(loss, metrics, seg), metrics)

    def _stage_read_back(self, loss, metrics, seg, flag=None):
        """Queue the one device-to-host copy of the step (pinned buffer + event); no host wait here.  With a
        data-parallel NaN flag (see _loss_flag) the copy runs on the side stream behind the flag's all-reduce."""
        if flag is not None:
            flag, side = flag
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                return self._stage_read_back_on_current(loss, metrics, seg, flag)
        return self._stage_read_back_on_current(loss, metrics, seg, None)

    def _stage_read_back_on_current(self, loss, metrics, seg, flag):
        parts, layout = [], []
        if loss is not None:
            parts.append(loss.detach().reshape(1)); layout.append(('loss', 1))
        dt = metrics.pop('_device_terms', None)
        if dt is not None:
            terms, meta = dt
            parts += [terms.reshape(-1), meta.n_sp.float(), meta.n_l.float()]
            layout += [('terms', terms.numel()), ('n_sp', meta.B), ('n_l', meta.B)]
        if seg is not None:
            parts.append(seg.reshape(-1)); layout.append(('seg', seg.numel()))
        if flag is not None:
            parts.append(flag); layout.append(('nan_anywhere', 1))
        if not parts:
            return None
        flat = torch.cat(parts)
        n = flat.numel()
        if self._host_buf is None or self._host_buf.numel() < n:
            self._host_buf = torch.empty(max(4096, n), dtype=torch.float32).pin_memory()
        self._host_buf[:n].copy_(flat, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        return (layout, n, done)

    def _finish_read_back(self, pending, metrics):
        out = {}
        if pending is None:
            return out
        layout, n, done = pending
        done.synchronize()                                           # <- host sync
        flat = self._host_buf[:n].numpy().astype(np.float64)
        o = 0
        for name, n in layout:
            out[name] = flat[o:o + n]
            o += n
        if 'loss' in out:
            out['loss'] = float(out['loss'][0])
        if 'nan_anywhere' in out:
            out['nan_anywhere'] = float(out['nan_anywhere'][0])
        self._metrics_from_terms(out, metrics)
        if 'seg' in out:
            out['seg'] = out['seg'].reshape(-1, 4)
        return out

    def _metrics_from_terms(self, out, metrics):
        """labeled_sp_ratio / propagated_labels / propagate_loss (models/wesup.py:520-524) from the per-image loss terms."""
        if 'terms' in out:
            t = out['terms'].reshape(-1, 8)
            n_sp, n_l = out['n_sp'], out['n_l']
            weak = n_l < n_sp
            if weak.any():
                metrics['labeled_sp_ratio'] = float(np.mean(n_l[weak] / n_sp[weak]))
                if self.kwargs.get('enable_propagation'):
                    metrics['propagated_labels'] = float(np.mean(t[weak, 4]))
                    metrics['propagate_loss'] = float(np.mean(np.where(t[weak, 3] > 0, t[weak, 2] / np.maximum(t[weak, 3], 1), 0.0)))

    def train_one_epoch(self, no_val=False):
        phases = ['train'] if no_val else ['train', 'val']
        for phase in phases:
            self.logger.info(f'{phase.capitalize()} phase:')
            start = time.time()
            if phase == 'train':
                self.model.train()
                self.tracker.train()
            else:
                self.model.eval()
                self.tracker.eval()
            for data in self.dataloaders[phase]:
                try:
                    self.train_one_iteration(phase, *data)
                except RuntimeError as ex:
                    # models/base.py:234-237: log and go on with the next batch.  Not under data parallelism: the other
                    # ranks are already inside this iteration's collectives, so a rank that skips a batch leaves them
                    # hanging or reducing mismatched buckets -- there the error ends the job (fail-stop; the launcher
                    # tears the other ranks down) after the reducer has been brought back to a clean state.
                    if self.reducer is not None and self.world_size > 1:
                        self.reducer.reset()
                        raise
                    self.logger.exception(ex)
                    if getattr(self.model, 'engine', None) is not None:
                        self.model.engine.ctx = None
            self.logger.info(f'Took {time.time() - start:.2f}s.')
            self.logger.info(self.tracker.log())

    def train(self, data_root, **kwargs):
        """Start training process (models/base.py:252-333)."""
        self.kwargs = {**self.kwargs, **kwargs}
        self.optimizer, self.scheduler = self.get_default_optimizer()
        if self.reducer is not None and hasattr(self.optimizer, 'grad_scale'):
            self.optimizer.grad_scale = 1.0 / self.world_size
        self.load_checkpoint(self.kwargs.get('checkpoint'))
        if self.kwargs.get('checkpoint') is None and getattr(self.model, 'pretrained_backbone', True) is False:
            self.logger.warning('WARNING: the VGG16 backbone starts from RANDOM weights; the reference starts from the '
                                'ImageNet weights (vgg16(pretrained=True), models/wesup.py:199) -- pass '
                                'backbone_weights=<file with a torchvision vgg16 state_dict> to reproduce its accuracy')
        if self.rank == 0:
            self.logger.addHandler(logging.FileHandler(self.record_dir / 'train.log'))
        serializable_kwargs = {k: v for k, v in self.kwargs.items() if isinstance(v, (int, float, str, tuple))}
        if self.rank == 0:
            record.save_params(self.record_dir, serializable_kwargs)
        self.logger.info(str(serializable_kwargs) + '\n')
        self.tracker.save_path = self.record_dir / 'history.csv'
        data_root = Path(data_root)
        train_path = data_root / 'train'
        val_path = data_root / 'val'
        train_dataset = self.get_default_dataset(train_path, proportion=self.kwargs.get('proportion', 1))
        train_dataset.summary(logger=self.logger)

        sampler = None
        if self.world_size > 1:
            from ..ddp import ShardSampler
            sampler = ShardSampler(len(train_dataset), self.rank, self.world_size, seed=0)
        workers = self.kwargs.get('num_workers', min(8, os.cpu_count() or 1))
        self.dataloaders = {
            'train': torch.utils.data.DataLoader(train_dataset, batch_size=self.kwargs.get('batch_size'),
                                                 shuffle=(sampler is None), sampler=sampler, num_workers=workers)
        }
        has_val = val_path.exists() or 'synthetic:' in str(data_root)
        if has_val and not self.kwargs.get('no_val'):
            val_dataset = self.get_default_dataset(val_path, train=False)
            val_dataset.summary(logger=self.logger)
            self.dataloaders['val'] = torch.utils.data.DataLoader(val_dataset, batch_size=1, num_workers=workers)
        else:
            has_val = False
        # datasets read from disk hand out raw uint8 items: H2D through pinned memory one batch ahead, augmentation /
        # ToTensor / one-hot on the GPU (utils/data.py: DevicePrefetcher)
        from ..utils import data as D
        for phase, ds in (('train', train_dataset), ('val', val_dataset if has_val else None)):
            if isinstance(ds, D.SegmentationDataset):
                self.dataloaders[phase] = D.DevicePrefetcher(
                    self.dataloaders[phase], self.device, train=(phase == 'train'),
                    with_points=isinstance(ds, D.PointSupervisionDataset), has_masks=ds.mask_paths is not None,
                    n_classes=ds.n_classes, seed=self.rank, segment_fn=self.prefetch_segment_fn())

        self.logger.info(underline('\nTraining Stage', '='))
        self.metric_funcs = self.kwargs.get('metrics')
        epochs = self.kwargs.get('epochs')
        total_epochs = epochs + self.initial_epoch - 1
        for epoch in range(self.initial_epoch, total_epochs + 1):
            self.logger.info(underline('\nEpoch {}/{}'.format(epoch, total_epochs), '-'))
            self.tracker.start_new_epoch(self.optimizer.param_groups[0]['lr'])
            if sampler is not None:
                sampler.set_epoch(epoch)                 # a new order and a new partition every epoch
            self.train_one_epoch(no_val=(not has_val))
            self.post_epoch_hook(epoch)
            if self.rank == 0:
                self.tracker.save()
                record.plot_learning_curves(self.tracker.save_path)
                ckpt_path = self.record_dir / 'checkpoints' / f'ckpt.{epoch:04d}.pth'
                self.save_checkpoint(ckpt_path, epoch=epoch, optimizer_state_dict=self.optimizer.state_dict())
                for old in sorted((self.record_dir / 'checkpoints').glob('*.pth'))[:-1]:
                    os.remove(old)
        if self.rank == 0:
            self.logger.info(self.tracker.report())

    def evaluate(self, pred, target=None, verbose=False):
        """Running several metrics to evaluate model performance (models/base.py:335-360)."""
        if target is None:
            return dict()
        metrics = defaultdict(list)
        for P, G in zip(pred, target):
            for func in self.metric_funcs:
                metrics[func.__name__].append(func(P, G))
        return {k: np.mean(v) for k, v in metrics.items()}
