"""Training driver with the CLI, outputs and loop semantics of the reference's
`linear_program_experiment.py` for the sparse bipartite methods ('gs-topk', 'soft-topk').

    python linear_program_experiment.py --cfg linear_program_netlib.yaml

reference linear_program_experiment.py:17-46 (config, seed, dataset, criterion), :115-157 (loop:
graph -> forward -> BCEWithLogitsLoss -> backward -> Adam step -> top-m F1 / correct count ->
print / train_log.json), :176-177 (torch.save(state_dict) to linear_program_<data>_<method>.pt).

What differs, by design: the graph of an instance is built once and stays in HBM (the reference
rebuilds it in Python every step, :124); forward/BCE/backward/Adam/metrics run in libmllp_hip.so;
`batch_size` (optional yaml key) > 1 groups instances into block-diagonal batches with ONE Adam step
per batch; with WORLD_SIZE > 1 (torchrun) each batch is sharded over the ranks and the flat gradient is
all-reduced over RCCL.  `batch_size: 1` on one GPU reproduces the reference's update sequence.
"""
import json
import os
import sys
import time

import numpy as np
import torch

from .config import load_config
from .data import LPInstance, get_netlib_dataset
from .model import GNNModel, set_seed

SPARSE_METHODS = ("gs-topk", "soft-topk")


def _dist_setup():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        if not dist.is_initialized():
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        return dist.get_rank(), world
    return 0, 1


def plan_batches(instances, batch_size, rank, world):
    """Groups of `batch_size` consecutive instances (0 = the whole dataset); with world > 1 every group is sharded
    over the ranks by nnz (trainer.shard_instances).  Returns [(global ids owned by this rank, global group size)]."""
    from .trainer import shard_instances
    bs = int(batch_size)
    if bs <= 0:
        bs = len(instances)
    plan = []
    for i in range(0, len(instances), bs):
        grp = list(range(i, min(i + bs, len(instances))))
        mine = grp if world == 1 else [grp[j] for j in shard_instances([instances[k].nnz for k in grp], world)[rank]]
        plan.append((mine, len(grp)))
    return plan


def run_epochs(cfg, instances, train_dict, trainer, batches, rank, world, device, start_epoch=0, out=print,
               save_all=None):
    """The epoch loop of reference linear_program_experiment.py:120-157 over prebuilt batches, for any trainer
    with `step(batch) -> (loss, logits)`, `step_empty()` and `metrics_of(batch) -> [n, 2]` (correct_num, f1).

    `batches` = [(global instance ids of THIS rank, batch or None, global instance count of the group)].
    With world > 1 the per-instance metrics of every rank are summed into one dense [n_instances, 2] tensor
    (each instance is owned by exactly one rank), so rank 0 prints and logs the complete epoch; the other
    ranks print nothing and write no files."""
    dist = None
    if world > 1:
        import torch.distributed as dist
    log_every = max(int(cfg.get_default("log_every")), 1)
    save_every = int(cfg.get_default("save_every"))
    for epoch in range(start_epoch, cfg.train_iter):                                # reference :120
        obj_sum = torch.zeros(1, device=device)
        logging = epoch % log_every == 0
        table = torch.zeros(len(instances), 2, device=device) if logging else None
        for mine, b, gcount in batches:
            if b is None:      # a rank without instances in this batch still joins the all-reduce
                trainer.step_empty()
                continue
            trainer.global_instances = gcount
            loss, _ = trainer.step(b)                                               # reference :124-144
            obj_sum += loss.to(device) * gcount     # loss is the batch mean over gcount instances (this rank's share)
            if logging:
                table[torch.as_tensor(mine, device=device)] = trainer.metrics_of(b).to(device)
        if dist is not None:
            dist.all_reduce(obj_sum)
            if logging:
                dist.all_reduce(table)
        obj = float(obj_sum[0]) / len(instances)
        if logging:
            met = table.cpu().numpy()
            for gi, inst in enumerate(instances):      # groups are consecutive id ranges: this is the dataset order
                correct_num, f1 = float(met[gi, 0]), float(met[gi, 1])              # reference :146-153
                if rank == 0:
                    out("%8d, %8d, %8d, %5f" % (correct_num, inst.m, inst.n, f1))
                train_dict[inst.name].append(correct_num)
        train_dict["obj"].append(obj)                  # plain float: the reference's numpy.float32 breaks json.dump
        if rank == 0:
            with open("train_log.json", "w") as json_file:                          # reference :155-156
                json.dump(train_dict, json_file)
            out(f"epoch {epoch}, obj={obj}")                                        # reference :157
        if save_all is not None and save_every and (epoch + 1) % save_every == 0:
            save_all(epoch)
    return train_dict


def train_method(cfg, method_name, train_dataset, train_dict, out=print):
    from .graph import LPBatch
    from .trainer import LPTrainer
    rank, world = _dist_setup()
    model_path = f"linear_program_{cfg.train_data_type}_{method_name}.pt"
    ckpt_path = f"linear_program_{cfg.train_data_type}_{method_name}.ckpt"
    if rank == 0:
        out(f"Training the model weights for {method_name}...")
    if str(cfg.get_default("dtype")).lower() not in ("f32", "fp32", "float32"):
        raise NotImplementedError(
            f"dtype: {cfg.get_default('dtype')!r}: the training step computes and stores fp32 (the reference's arithmetic, "
            "north_star's 1e-5 parity bar); the opt-in bf16 feature image exists for the plain SpMM only "
            "(LPBatch.spmm_bf16 / mllp_spmm_csr_bf16), where it measured slower than fp32 (DESIGN.md section 6)")
    device = torch.device(cfg.get_default("device"))
    if device.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError("this build runs the learned-LP path on MI355X through HIP only (device: 'cuda'); "
                           "there is no CPU fallback")
    model = GNNModel().to(device)                     # reference :117
    if world > 1:                                     # same initial weights on every rank
        import torch.distributed as dist
        flat0 = model.flat_parameters().detach().clone()
        dist.broadcast(flat0, src=0)
        model.load_flat(flat0)
    instances = [LPInstance.from_reference_tuple(t) for t in train_dataset]
    batches = [(mine, LPBatch.from_instances([instances[i] for i in mine]) if mine else None, gcount)
               for mine, gcount in plan_batches(instances, cfg.get_default("batch_size"), rank, world)]
    trainer = LPTrainer(model.flat_parameters().detach(), lr=cfg.train_lr,
                        use_hip_graph=cfg.get_default("use_hip_graph"), with_metrics=True,
                        tiled_copies=cfg.get_default("tiled_copies"))
    start_epoch = 0
    if cfg.get_default("resume") and os.path.exists(ckpt_path):
        start_epoch = load_checkpoint(ckpt_path, trainer, train_dict, device)
        if rank == 0:
            out(f"resumed from {ckpt_path} at epoch {start_epoch}")

    def save_all(epoch):
        if rank != 0:
            return
        model.load_flat(trainer.params)
        torch.save(model.state_dict(), model_path)                                   # reference :176
        save_checkpoint(ckpt_path, trainer, train_dict, epoch)

    run_epochs(cfg, instances, train_dict, trainer, batches, rank, world, device, start_epoch, out, save_all)
    save_all(cfg.train_iter - 1)
    if rank == 0:
        out(f"Model saved to {model_path}.")
    return model_path


def save_checkpoint(path, trainer, train_dict, epoch):
    """Everything a restart needs: weights, Adam moments + step counter, the epoch, the log so far."""
    torch.save({"params": trainer.params, "opt": trainer.opt.state_dict(), "epoch": epoch,
                "train_dict": json.dumps(train_dict)}, path)


def load_checkpoint(path, trainer, train_dict, device):
    ck = torch.load(path, map_location=device, weights_only=True)
    trainer.params.copy_(ck["params"])
    trainer.opt.load_state_dict(ck["opt"])
    train_dict.update({k: list(v) for k, v in json.loads(ck["train_dict"]).items()})
    return int(ck["epoch"]) + 1


def train_angle(cfg, method_name, train_dataset, train_dict, out=print):
    """reference linear_program_experiment.py:81-114: AngleModel(feat_dim=256) on the complete angle graph of the
    dense dataset's instance, BCEWithLogits + Adam, top-k metrics, train_log.json, state_dict .pt."""
    from .angle import AngleModel, AngleStepper, get_netlib_dataloader
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        # one instance, one dense graph: nothing to shard, and N ranks would all write the same files
        raise RuntimeError("the angleNet method trains a single instance: run it with one rank")
    out(f"Training the model weights for {method_name}...")
    device = torch.device(cfg.get_default("device"))
    if device.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError("this build runs the learned-LP path on MI355X through HIP only (device: 'cuda'); "
                           "there is no CPU fallback")
    train_loader = get_netlib_dataloader(train_dataset, device)                     # reference :33
    model = AngleModel(feat_dim=int(cfg.get_default("angle_feat_dim"))).to(device)  # reference :83
    # reference :41, :87-96: BCEWithLogitsLoss + torch.optim.Adam around autograd -- here the same arithmetic on flat
    # parameters without autograd (AngleStepper: C ABI forward / backward, the library's Adam kernel; tests/test_angle.py
    # checks it step for step against the nn.Module + torch.optim loop), 2-3 x faster per step on small instances
    stepper = AngleStepper(model, lr=cfg.train_lr)
    for epoch in range(cfg.train_iter):
        obj_sum = 0.0
        for graph in train_loader:
            name, basis_num, var_num = graph.name, graph.basis_num, graph.var_num
            basis_opt = torch.tensor(np.asarray(graph.basis_opt), dtype=torch.float, device=device)
            obj, latent_vars = stepper.step(graph, basis_opt)
            obj_sum += float(obj.detach())
            pred_indices = torch.topk(latent_vars, k=basis_num)[-1].cpu().detach().numpy()
            pred = np.zeros(var_num)
            pred[pred_indices] = 1
            truth = basis_opt.cpu().numpy()
            correct_num = float(pred @ truth)
            tp = correct_num
            f1 = 2.0 * tp / max(pred.sum() + truth.sum(), 1.0)                      # sklearn f1_score on {0,1} vectors
            out("%8d, %8d, %8d, %5f" % (correct_num, basis_num, var_num, f1))
            train_dict[name].append(correct_num)
        train_dict["obj"].append(obj_sum / len(train_dataset))
        with open("train_log.json", "w") as json_file:
            json.dump(train_dict, json_file)
        out(f"epoch {epoch}, obj={obj_sum / len(train_dataset)}")
    model_path = f"linear_program_{cfg.train_data_type}_{method_name}.pt"
    model.load_flat(stepper.params)
    torch.save(model.state_dict(), model_path)                                      # reference :176
    out(f"Model saved to {model_path}.")
    return train_dict


def main(argv=None):
    cfg = load_config(argv)                                                         # reference :17
    set_seed()                                                                      # reference :19
    if cfg.train_data_type == "netlib":                                             # reference :28-35
        names = cfg.get_default("instances")
        if cfg.methods[0] == "invariant":
            raise NotImplementedError(
                "method 'invariant' is the reference's InvariantModel research path (reference "
                f"linear_program_methods.py:136-185), not built: use 'angleNet' or one of {SPARSE_METHODS}")
        if cfg.methods[0] == "angleNet":                                            # reference :31-33
            from .data import get_netlib_dataset_dense
            train_dataset, train_dict = get_netlib_dataset_dense(normalize=True, names=names)
        else:
            train_dataset, train_dict = get_netlib_dataset(normalize=True, names=names)
    else:
        raise ValueError(f"Unknown training dataset {cfg.train_data_type}!")
    for method_name in cfg.methods:                                                 # reference :45
        if method_name in SPARSE_METHODS:
            train_method(cfg, method_name, train_dataset, train_dict)
        elif method_name == "angleNet":
            train_angle(cfg, method_name, train_dataset, train_dict)
        else:
            raise NotImplementedError(f"method {method_name!r} is outside the sparse bipartite hot path of this "
                                      f"build (supported: {SPARSE_METHODS})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
