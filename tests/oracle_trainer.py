"""Test infrastructure: a CPU trainer with LPTrainer's interface (step / step_empty / metrics_of / params / opt)
whose arithmetic is the oracle's literal restatement.  It lets the experiment driver's epoch loop
(mllp_amd.experiment.run_epochs: sharding, empty ranks, metric gather, log, checkpoint) run on the CPU, with gloo
for world_size 2.  Never imported by the product."""
import numpy as np
import torch

from mllp_amd.trainer import FlatAdam, allreduce_sum_
from oracle import pyg_restatement as o1


class CpuBatch:
    def __init__(self, instances):
        self.instances = list(instances)
        self.n_inst = len(self.instances)


class OracleTrainer:
    def __init__(self, params_flat, lr=1e-3):
        self.params = params_flat.detach().clone().double()
        self.opt = FlatAdam(self.params, lr=lr, backend="torch")
        self.global_instances = None
        self._metrics = {}

    def step(self, batch):
        sd = o1.unflatten_state(self.params)
        loss, logits, grads = o1.batch_loss_and_grads(sd, batch.instances, torch.float64,
                                                      global_count=self.global_instances or batch.n_inst)
        self._metrics[id(batch)] = torch.tensor(
            [o1.topk_metrics(z.numpy(), i.m, i.basis) for z, i in zip(logits, batch.instances)], dtype=torch.float32)
        allreduce_sum_(grads)
        self.opt.step(grads)
        return loss.reshape(1).float(), logits

    def step_empty(self):
        zero = torch.zeros_like(self.params)
        allreduce_sum_(zero)
        self.opt.step(zero)

    def metrics_of(self, batch):
        return self._metrics[id(batch)]
