"""Drop-in for the reference's `train` module: `TrainingState` and `Trainer` with the reference's public
surface (train.py:14-23,27,54,59-61,129,145,320-339) and checkpoint layout
`<models_root>/<name>/<name>{best_model,last_model,_epoch_N}.pth = torch.save({'state': TrainingState,
'model': module})`, driving the HIP model/loss of this package.

Differences that are deliberate:
  * `torch.load(..., weights_only=False)`: under torch >= 2.6 the reference's own `_load` cannot read its own
    whole-module pickles;
  * the scheduler assert accepts `LRScheduler` (StepLR is no longer an `_LRScheduler` subclass instance);
  * tensorboardX is optional (absent in this image): scalars go to a no-op writer when it cannot be imported;
  * `eval_cpu=True` is refused: this path is HIP-only;
  * tile geometry of `predict_tiled` is a parameter (defaults = the reference's literals 192/48/72);
  * data parallelism (main.py:61 wraps the model in nn.DataParallel so that this loop is data-parallel without knowing it): here the
    loop itself is rank-aware.  Launched as one process per GPU (`torch.distributed.run`, process group initialised before
    `Trainer.train`), every rank takes its slice of each batch the loader yields (what DataParallel's scatter does), the criteria
    form the GLOBAL-batch loss (`data_parallel`, loss.py:114-115 sums over the whole batch), the parameter gradients are summed
    over the ranks after `loss.backward()` (DataParallel's reduce_add) and every replica takes the same optimizer step; rank 0
    alone writes checkpoints and scalars.  `shard_batches = False` if the loader already yields per-rank shards.
"""
from __future__ import annotations

import os
import shutil
import time

import numpy as np
import torch

from . import tiling
from .parallel import all_reduce_gradients, world_info


class TrainingState(object):
    """train.py:14-23."""

    def __init__(self):
        self.epoch = 0
        self.train_metric = dict()
        self.val_metric = dict()
        self.global_step = 0           # number of processed batches
        self.best_val = 0
        self.optimizer_state = None
        self.cuda = True


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass


def _make_writer(logdir):
    try:
        from tensorboardX import SummaryWriter
        return SummaryWriter(logdir=logdir)
    except Exception:
        return _NullWriter()


def _log_metric(writer, metric, prefix, epoch):
    """metrics.print_metrics analogue (metrics.py:273-280): one scalar per entry of metric.get()."""
    val = np.atleast_1d(np.asarray(metric.get(), dtype=np.float64))
    for i, v in enumerate(val):
        writer.add_scalar("%s%s-%d" % (prefix, metric.name, i), float(v), epoch)
    if world_info()[0] == 0:
        print("%s%s: %s" % (prefix, metric.name, np.array2string(val, precision=4)))


class Trainer(object):
    def __init__(self, name, models_root, model=None, rewrite=False, connect_tb=True):
        assert isinstance(model, (list, tuple, torch.nn.Module)) or model is None
        self.model = model
        self.name = name
        self.models_root = models_root
        self.model_path = os.path.join(models_root, self.name)
        self.logs_path = os.path.join(self.model_path, "logs")
        self.state = TrainingState()
        self.resume_training = False
        self.shard_batches = True          # data parallel: slice each loader batch per rank (nn.DataParallel's scatter)
        self.hip_optimizer = True          # Trainer.train(optimizer=torch.optim.Adam, ...) -> brats2019_amd.optim.Adam (same surface, HIP kernel)
        self.fuse_criteria = True          # criterion=[Dice_loss_joint, BCE_Loss] -> one sums pass + one gradient pass (loss.fuse_criterion_list)
        self.log_every = 50                # loss scalars are queued on the device and written every `log_every` steps: no per-step host sync
        self._pending_scalars = []
        rank, world = world_info()
        if rank == 0:
            if os.path.exists(self.model_path):
                if rewrite:
                    shutil.rmtree(self.model_path)
                else:
                    self.resume_training = True
            if not os.path.exists(self.model_path):
                os.makedirs(self.logs_path)
        if world > 1:                      # every rank follows rank 0's view of the directory
            import torch.distributed as dist
            flag = [self.resume_training]
            dist.broadcast_object_list(flag, src=0)
            self.resume_training = bool(flag[0])
        self.tb_writer = _make_writer(self.logs_path) if (connect_tb and rank == 0) else _NullWriter()
        self.tile_shape, self.center_shape, self.border = (192, 192, 192), (48, 48, 48), (72, 72, 72)   # train.py:154-156

    def cuda(self):
        if self.model is not None:
            self.model.cuda()
        self.state.cuda = True

    # ------------------------------------------------------------------ training (train.py:59-127)
    def train(self, criterion, optimizer, optimizer_params, scheduler, scheduler_params, training_data_loader,
              evaluation_data_loader, split_into_tiles, pretrained_weights, train_metrics, val_metrics,
              track_metric, epoches, default_val, comparator, eval_cpu, continue_form_pretraining):
        if eval_cpu:
            raise NotImplementedError("eval_cpu=True needs a CPU model; this engine is HIP-only")
        self.eval_cpu = eval_cpu
        assert isinstance(criterion, (tuple, list, torch.nn.Module))
        rank, world = world_info()
        if world > 1:
            # one process per GPU: each rank sees its shard only, so the criteria must reduce their partial sums over the ranks to
            # form the loss nn.DataParallel computes on GPU 0 over the gathered global batch (main.py:61, train.py:201-208)
            for c in (criterion if isinstance(criterion, (tuple, list)) else [criterion]):
                if not hasattr(c, "data_parallel"):
                    raise RuntimeError("data-parallel training needs criteria that reduce over the global batch "
                                       "(brats2019_amd.loss modules); got %s" % type(c).__name__)
                c.data_parallel = True
        if self.resume_training:
            self.load_latest()
        elif pretrained_weights is not None:
            self.model.load_state_dict(pretrained_weights)
        elif continue_form_pretraining:
            print("Continue from pretraining")
        else:
            self.state.best_val = default_val
        if self.state.cuda:
            self.model.cuda()
        optimizer, scheduler = self._make_optimizer(optimizer, optimizer_params, scheduler, scheduler_params)
        if self.state.optimizer_state is not None and not continue_form_pretraining:
            optimizer.load_state_dict(self.state.optimizer_state)
            print("Loaded optimizer state")
        if not self.state.train_metric:
            for m in train_metrics:
                self.state.train_metric[m.name] = []
            for m in val_metrics:
                self.state.val_metric[m.name] = []
        for epoch in range(self.state.epoch, epoches):
            tic = time.time()
            self.state.global_step = self._train_one_epoch(criterion, optimizer, training_data_loader, train_metrics,
                                                           self.state.train_metric, epoch, self.state.global_step, scheduler)
            self._evaluate_and_save(evaluation_data_loader, split_into_tiles, val_metrics, track_metric,
                                    self.state.val_metric, epoch, comparator)
            print("Epoch %d, time %s \n" % (epoch, time.time() - tic))
            self._save(suffix="_epoch_" + str(self.state.epoch))
            self._save(suffix="last_model")
            self.state.epoch = self.state.epoch + 1

    def _make_optimizer(self, optimizer, optimizer_params, scheduler, scheduler_params):
        """train.py:82-90: instantiate the optimizer / scheduler classes the caller handed over."""
        if isinstance(optimizer, type):
            # main.py:134 passes the CLASS torch.optim.Adam: on the HIP path it is instantiated as brats2019_amd.optim.Adam -- the same
            # torch.optim.Optimizer surface and state_dict() layout (a checkpoint of either resumes under the other, train.py:92-94),
            # with the update as one ru_adam_step launch per contiguous run of the flat parameter buffer.  `hip_optimizer = False` keeps torch's.
            made = None
            if optimizer is torch.optim.Adam and self.hip_optimizer and self.state.cuda:
                from . import optim as hip_optim
                try:
                    made = hip_optim.Adam(params=self.model.parameters(), **optimizer_params)
                except NotImplementedError:              # options only torch's own Adam takes (fused=, foreach=, ...): keep the caller's class
                    made = None
            optimizer = made if made is not None else optimizer(params=self.model.parameters(), **optimizer_params)
        if scheduler is not None and isinstance(scheduler, type):
            scheduler = scheduler(optimizer=optimizer, **scheduler_params)
        assert isinstance(optimizer, torch.optim.Optimizer)
        assert scheduler is None or isinstance(scheduler, torch.optim.lr_scheduler.LRScheduler)
        return optimizer, scheduler

    def _to_device(self, tensors):
        return [t.cuda(non_blocking=True) for t in tensors] if self.state.cuda else list(tensors)

    def _shard(self, tensors):
        """This rank's slice of every tensor of a loader batch: samples [r*B/W, (r+1)*B/W)."""
        rank, world = world_info()
        if world <= 1 or not self.shard_batches:
            return list(tensors)
        out = []
        for t in tensors:
            b = int(t.shape[0])
            if b % world != 0:
                raise ValueError("global batch %d is not divisible by the %d data-parallel ranks" % (b, world))
            out.append(t[rank * (b // world):(rank + 1) * (b // world)])
        return out

    @staticmethod
    def _reduce_metric(m):
        """Equal shards: the global-batch mean a metric accumulates per update is the mean of the ranks' shard means."""
        rank, world = world_info()
        if world <= 1 or not hasattr(m, "accumulator"):
            return
        import torch.distributed as dist
        acc = m.accumulator
        acc = acc.detach().to(torch.float64) if isinstance(acc, torch.Tensor) else torch.as_tensor(np.asarray(acc, dtype=np.float64))
        acc = acc.cuda() if (torch.cuda.is_available() and dist.get_backend() == "nccl") else acc.cpu()
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        acc = (acc / world).cpu().numpy()
        m.accumulator = acc if acc.ndim else float(acc)

    def _train_one_epoch(self, criterion, optimizer, loader, train_metrics, results, epoch, global_step, scheduler):
        for m in train_metrics:
            m.reset()
        if self.state.cuda:
            self.model.cuda()
        self.model.train()
        optimizer.zero_grad()
        from . import loss as hip_loss
        fused = hip_loss.fuse_criterion_list(criterion) if self.fuse_criteria else None
        for batch in loader:
            assert isinstance(batch[0], list) and isinstance(batch[1], list)
            data, target = self._to_device(self._shard(batch[0])), self._to_device(self._shard(batch[1]))
            output = self.model(data)                                        # train.py:201
            if fused is not None:
                loss, loss_val = fused(output, target)                       # the same list of values and their mean, one pass per phase
            elif isinstance(criterion, (tuple, list)):
                loss_val = [c(output, target) for c in criterion]            # train.py:203-205
                loss = sum(loss_val) / len(loss_val)
            else:
                loss_val = [criterion(output, target)]
                loss = loss_val[0]
            with hip_loss.hand_over_to_network():                            # this loop never looks at d(loss)/d(probs): the network's node forms it
                loss.backward()                                              # train.py:210
            all_reduce_gradients(self.model)                                 # nn.DataParallel's reduce_add (main.py:61); no-op on one rank
            optimizer.step()
            optimizer.zero_grad()
            if scheduler is not None:
                scheduler.step()                                             # per iteration (train.py:222-223)
            for m in train_metrics:
                m.update(output, target)
            if not isinstance(self.tb_writer, _NullWriter):
                # train.py:226-227 logs lv.item() per step -- a host sync per step; here the values wait on the device and go out in batches
                for i, lv in enumerate(loss_val):
                    self._pending_scalars.append(("loss/loss-%d" % i, lv.detach(), global_step))
                for i, group in enumerate(optimizer.param_groups):
                    self.tb_writer.add_scalar("misc/lr-%d" % i, group["lr"], global_step)
                if len(self._pending_scalars) >= self.log_every * max(1, len(loss_val)):
                    self._flush_scalars()
            global_step += 1
        self._flush_scalars()
        for m in train_metrics:
            self._reduce_metric(m)
            results[m.name].append(m.get())
            _log_metric(self.tb_writer, m, "train/", epoch)
        self.state.optimizer_state = optimizer.state_dict()
        return global_step

    def _flush_scalars(self):
        if not self._pending_scalars:
            return
        vals = torch.stack([v.reshape(()).to(torch.float64) for _, v, _ in self._pending_scalars]).cpu().tolist()     # one device -> host copy
        for (tag, _, step), v in zip(self._pending_scalars, vals):
            self.tb_writer.add_scalar(tag, v, step)
        self._pending_scalars = []

    # ------------------------------------------------------------------ inference (train.py:129-176)
    def _freeze_for_eval(self):
        """Evaluation runs many forwards on constant weights: the executor packs them once (dropped again by model.train() and by
        load_state_dict, model.UNet.freeze_params)."""
        net = self.model.module if hasattr(self.model, "module") else self.model
        if hasattr(net, "freeze_params") and self.state.cuda:
            net.freeze_params(True)

    def predict(self, batch):
        self.model.eval()
        if self.state.cuda:
            self.model.cuda()
        self._freeze_for_eval()
        with torch.no_grad():
            assert isinstance(batch[0], list)
            return self.model(self._to_device(batch[0]))

    def _auto_batch_tiles(self, tile_shape, nvol, cap=8):
        """Tiles per forward: as many as fit in a third of the free device memory (the inference arena keeps every activation of a
        forward: ~5.7 GiB per 192^3 tile in bf16x3, 7.6 GiB in f32), at most `cap`."""
        net = self.model.module if hasattr(self.model, "module") else self.model
        if not (torch.cuda.is_available() and hasattr(net, "_get_engine")):
            return 1
        from . import _lib as L
        eng = net._get_engine()
        per = L.load().ru_unet_workspace_bytes(eng.h, int(nvol), int(tile_shape[0]), int(tile_shape[1]), int(tile_shape[2]), 0)
        if per == 0:
            return 1
        free, _total = torch.cuda.mem_get_info()
        return int(max(1, min(cap, (free // 3) // per)))

    def predict_tiled(self, batch, output_shape, tile_shape=None, center_shape=None, border=None, batch_tiles=None):
        """train.py:145-176: per centre block, run the model on the zero-padded tile and paste the centre back.
        The volume goes to the device once, tiles are cut there and `batch_tiles` of them share one forward (every op of
        the network is per-sample, so batching tiles does not change a single value); the pasted result stays on the
        device until the end (the reference does .cuda()/.cpu() per tile).  `batch_tiles=None`: sized from the free device
        memory (at most 8)."""
        tile_shape = tuple(tile_shape or self.tile_shape)
        center_shape = tuple(center_shape or self.center_shape)
        border = tuple(border or self.border)
        inp = batch[0][0]
        if self.state.cuda:
            self.model.cuda()
            inp = inp.cuda(non_blocking=True)
        self.model.eval()
        self._freeze_for_eval()
        if batch_tiles is None:
            batch_tiles = self._auto_batch_tiles(tile_shape, int(inp.shape[0]))
            rank, world = world_info()
            if world > 1:                      # sized from each rank's free memory: every replica takes the smallest, so all run the same kernels
                import torch.distributed as dist
                t = torch.tensor([int(batch_tiles)], dtype=torch.int64, device=inp.device if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                batch_tiles = int(t.item())
        inp = inp.contiguous().float()
        output = torch.zeros(output_shape, dtype=torch.float32, device=inp.device)
        grid = tiling.grid_for(inp.shape[2:], center_shape)
        positions = [(i, j, k) for i in range(grid[0]) for j in range(grid[1]) for k in range(grid[2])]
        nvol = int(inp.shape[0])
        with torch.no_grad():
            for s0 in range(0, len(positions), max(1, int(batch_tiles))):
                chunk = positions[s0:s0 + max(1, int(batch_tiles))]
                los = [tiling.get_indices(pos, center_shape, border)[0] for pos in chunk]
                tiles = tiling.copy_tiles(inp, tile_shape, los)                  # one gather launch straight into the batch tensor
                out = self.model([tiles])[0]
                tiling.copy_back_tiles(output, out, center_shape, los, border)   # one scatter launch
        return [output.cpu()]

    def _evaluate_and_save(self, loader, split_into_tiles, val_metrics, track_metric, results, epoch, comparator):
        for m in val_metrics:
            m.reset()
        self.model.eval()
        for batch in loader:
            assert isinstance(batch[0], list) and isinstance(batch[1], list)
            if split_into_tiles:
                output = self.predict_tiled(batch, tuple(batch[1][0].shape))
                target = list(batch[1])
            else:
                output = self.predict(batch)
                target = self._to_device(batch[1])
            for m in val_metrics:
                m.update(target, output)                                     # train.py:304: (ground, predict)
        val = 0.0
        for m in val_metrics:
            if m.name == track_metric:
                val = m.get()
            _log_metric(self.tb_writer, m, "val/", epoch)
            results[m.name].append(m.get())
        better = bool(comparator(val, self.state.best_val))                  # train.py:315-318
        rank, world = world_info()
        if world > 1:
            # every rank validates the whole set on its own replica; a mask voxel on the threshold could make the ranks disagree, and
            # `_save` contains a barrier: rank 0 decides for everybody (and its metric value is the one recorded)
            import torch.distributed as dist
            box = [better, val]
            dist.broadcast_object_list(box, src=0)
            better, val = bool(box[0]), box[1]
        if better:
            self.state.best_val = val
            self._save(suffix="best_model")
            print("model saved")

    # ------------------------------------------------------------------ checkpoints (train.py:320-339)
    def _ckpt(self, suffix):
        return os.path.join(self.model_path, self.name + suffix + ".pth")    # no separator, like the reference

    def _save(self, suffix):
        rank, world = world_info()
        if rank == 0:                          # replicas are identical: one writer
            torch.save({"state": self.state, "model": self.model}, self._ckpt(suffix))
        if world > 1:
            import torch.distributed as dist
            dist.barrier()                     # nobody reads (load_latest on resume) a file that is still being written

    def _load(self, suffix):
        from .compat import install_aliases
        install_aliases()                      # reference pickles name the modules `model` and `train`
        print("loading model %s" % suffix)
        s = torch.load(self._ckpt(suffix), map_location=torch.device("cpu"), weights_only=False)
        self.state = s["state"]
        if self.model is None:
            self.model = s["model"]                    # train.py:329-330: the pickled module, wrapper included
            if isinstance(self.model, torch.nn.DataParallel) and len(self.model.device_ids or []) > 1:
                # saved by the reference under --gpus > 1 (main.py:61): the several-device wrapper would scatter over GPUs this process
                # does not own -- data parallelism is one process per GPU here -- so the module itself is kept (a single-device wrapper
                # stays as pickled)
                print("checkpoint holds nn.DataParallel over %d devices: using its .module (one process per GPU)" % len(self.model.device_ids))
                self.model = self.model.module
        else:
            src = s["model"].state_dict()
            want = self.model.state_dict().keys()
            # a DataParallel-saved checkpoint carries the `module.` prefix (export_onnx_group_norm.py:28-32); either side may be wrapped
            want_pref, src_pref = any(k.startswith("module.") for k in want), any(k.startswith("module.") for k in src)
            if src_pref and not want_pref:
                src = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in src.items()}
            elif want_pref and not src_pref:
                src = {"module." + k: v for k, v in src.items()}
            self.model.load_state_dict(src)

    def load_latest(self):
        self._load("last_model")

    def load_best(self):
        self._load("best_model")
