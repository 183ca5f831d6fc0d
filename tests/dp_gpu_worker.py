#!/usr/bin/env python3
"""One data-parallel rank on cuda:0 (child process of tests/test_parallel_gpu.py; also runs stand-alone).

    RANK=r WORLD_SIZE=W MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/dp_gpu_worker.py <mode> <outdir> [backend]

W ranks share ONE GPU (the pool's boxes have one), so the transport is gloo unless W == 1 (then `nccl` = RCCL runs too); what is under
test is the product path: HipBackend / model.UNet + loss modules through libresunet_hip.so, sharded over the ranks.  With W = 1 the
same script produces the single-process global-batch result the sharded runs are compared with.

modes:  step     parallel.DataParallelStep (what bench.py times): 2 steps of the shipped configuration in bf16x3
        trainer  train.Trainer.train with the reference's loop (train.py:59-127,178-241): 1 epoch of 2 global batches, torch Adam(amsgrad)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from oracle import resunet_oracle as O       # seeded inputs / weights only (data, not arithmetic)

CFG = O.DEFAULT_CFG
DHW = (32, 32, 32)
GLOBAL_BATCH = 2
SEED = 41


def main():
    mode, out_dir = sys.argv[1], sys.argv[2]
    backend = sys.argv[3] if len(sys.argv) > 3 else "gloo"
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0)
    group = None
    if backend == "gloo+nccl":
        # what bench.py's ranks do (parallel.init_process_groups_with_fallback): gloo default group, RCCL data group on top, health check
        from brats2019_amd import parallel as P0
        os.environ.setdefault("LOCAL_RANK", "0")
        _r, _l, _w, group, info = P0.init_process_groups_with_fallback("nccl", allow_single=True, device_index=0)
        assert info == {"backend": "nccl", "fallback_reason": None}, info
        assert group is not None and dist.get_backend(group) == "nccl" and dist.get_backend() == "gloo"
    elif world > 1 or backend == "nccl":
        dist.init_process_group("gloo" if backend == "rccl-direct" else backend, rank=rank, world_size=world)
    from brats2019_amd import parallel as P
    x = torch.from_numpy(O.make_input(GLOBAL_BATCH, *DHW, seed=SEED))
    g = torch.from_numpy(O.make_target(GLOBAL_BATCH, *DHW, seed=SEED))
    params = O.make_params(SEED, **CFG)
    res = {}
    if mode == "step":
        be = P.HipBackend(cfg=CFG)
        assert be.engine.precision == "bf16x3"
        flat = be.new_flat()
        for k, v in be.engine.layout.views(flat).items():
            v.copy_(torch.from_numpy(params[k]))
        comm = P.RcclComm(rank, world) if backend == "rccl-direct" else None      # ru_comm_* / ru_allreduce instead of torch.distributed
        st = P.DataParallelStep(be, flat, lr=1e-3, step_size=1, gamma=0.5, comm=comm, process_group=group)
        sl = P.DataParallelStep.shard(GLOBAL_BATCH, rank, world)
        xs, gs = x[sl].cuda(), g[sl].cuda()
        l1, d1, b1 = st.loss_and_grads(xs, gs)
        res.update(loss=float(l1), dice=float(d1), bce=float(b1), grads=st.grads.cpu().numpy())
        st2 = P.DataParallelStep(be, flat, lr=1e-3, step_size=1, gamma=0.5, comm=comm, process_group=group)
        la = float(st2.step(xs, gs)[0])
        lb = float(st2.step(xs, gs)[0])
        res.update(l_step1=la, l_step2=lb, weights=flat.cpu().numpy())
        if comm is not None:
            probe32 = torch.arange(1000, dtype=torch.float32, device="cuda") * 0.5
            probe64 = torch.arange(7, dtype=torch.float64, device="cuda") + 0.25
            comm.all_reduce(probe32); comm.all_reduce(probe64)
            res.update(probe32=probe32.cpu().numpy(), probe64=probe64.cpu().numpy(), comm_world=comm.world)
            comm.close()
    elif mode == "trainer":
        from brats2019_amd import model as M, loss as L, train as TR, metrics as MT
        net = M.UNet(**CFG)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        tr = TR.Trainer(name="dp", models_root=os.path.join(out_dir, "models_w%d" % world), model=net, rewrite=True, connect_tb=False)
        losses = []

        class Rec:
            def add_scalar(self, tag, val, step):
                if tag.startswith("loss/"):
                    losses.append((tag, float(val), int(step)))
        tr.tb_writer = Rec()
        x2 = torch.from_numpy(O.make_input(GLOBAL_BATCH, *DHW, seed=SEED + 1))
        g2 = torch.from_numpy(O.make_target(GLOBAL_BATCH, *DHW, seed=SEED + 1))
        loader = [([x], [g]), ([x2], [g2])]                       # the loader yields GLOBAL batches, like under nn.DataParallel
        ev = [([x2[:1]], [g2[:1]])]
        tr.train(criterion=[L.Dice_loss_joint(index=0, priority=1), L.BCE_Loss(index=0, bg_weight=1e-2)],      # main.py:126-128
                 optimizer=torch.optim.Adam, optimizer_params=dict(lr=1e-3, weight_decay=1e-6, amsgrad=True),   # main.py:133-137 (lr raised)
                 scheduler=torch.optim.lr_scheduler.StepLR, scheduler_params=dict(step_size=1, gamma=0.5),
                 training_data_loader=loader, evaluation_data_loader=ev, split_into_tiles=False, pretrained_weights=None,
                 train_metrics=[MT.Dice(name="Dice")], val_metrics=[MT.Dice(name="Dice")], track_metric="Dice", epoches=1,
                 default_val=np.zeros(3), comparator=lambda a, b: np.min(a) + np.mean(a) > np.min(b) + np.mean(b),      # main.py:155
                 eval_cpu=False, continue_form_pretraining=False)
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu().numpy()
        res.update(weights=flat, losses=np.asarray([v for _t, v, _s in losses]), train_dice=np.asarray(tr.state.train_metric["Dice"][0], np.float64),
                   val_dice=np.asarray(tr.state.val_metric["Dice"][0], np.float64), global_step=tr.state.global_step,
                   ckpt=int(os.path.exists(os.path.join(tr.model_path, "dplast_model.pth"))))
    else:
        raise SystemExit("unknown mode " + mode)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "%s_w%d_r%d.npz" % (mode, world, rank)), **res)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
