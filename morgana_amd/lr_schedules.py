"""Mirror of ``morgana.lr_schedules`` (host-side scalars).  Reference: morgana/lr_schedules.py:1-146."""
from functools import partial

from torch.optim import lr_scheduler

EPOCH_LR_SCHEDULES = ['constant', 'lambda', 'step', 'multi_step', 'exponential', 'cosine_annealing',
                      'cosine_annealing_warm_restarts']
BATCH_LR_SCHEDULES = ['cyclic', 'noam', 'cyclic_noam']


class DummyLR(lr_scheduler._LRScheduler):
    """Constant learning rate (lr_schedules.py:33-39)."""

    def __init__(self, optimizer):
        super(DummyLR, self).__init__(optimizer)

    def get_lr(self):
        return self.base_lrs


class NoamLR(lr_scheduler._LRScheduler):
    """scale = warmup^0.5 * min(step^-0.5, step * warmup^-1.5), step = max(1, last_epoch) (lr_schedules.py:45-89)."""

    def __init__(self, optimizer, warmup_steps=4000):
        self.warmup_steps = warmup_steps
        super(NoamLR, self).__init__(optimizer)

    def scale(self, step):
        return self.warmup_steps ** 0.5 * min(step ** (-0.5), step * self.warmup_steps ** (-1.5))

    def get_lr(self):
        scale = self.scale(max(1, self.last_epoch))
        return [base_lr * scale for base_lr in self.base_lrs]


class CyclicNoamLR(NoamLR):
    """Noam pattern repeating every ``cycle_steps`` batches (lr_schedules.py:95-142)."""

    def __init__(self, optimizer, warmup_steps=4000, cycle_trigger=0.2, cycle_steps=None):
        self.warmup_steps = warmup_steps
        if cycle_steps is None:
            self.cycle_steps = int((cycle_trigger / self.warmup_steps ** 0.5) ** -2)
        else:
            self.cycle_steps = cycle_steps
        super(CyclicNoamLR, self).__init__(optimizer, warmup_steps=warmup_steps)

    def get_lr(self):
        scale = self.scale(max(1, self.last_epoch % self.cycle_steps))
        return [base_lr * scale for base_lr in self.base_lrs]


SUPPORTED = {
    'lambda': lr_scheduler.LambdaLR, 'step': lr_scheduler.StepLR, 'multi_step': lr_scheduler.MultiStepLR,
    'exponential': lr_scheduler.ExponentialLR, 'cosine_annealing': lr_scheduler.CosineAnnealingLR,
    'plateau': lr_scheduler.ReduceLROnPlateau, 'cyclic': lr_scheduler.CyclicLR,
    'cosine_annealing_warm_restarts': lr_scheduler.CosineAnnealingWarmRestarts,
    'constant': DummyLR, 'noam': NoamLR, 'cyclic_noam': CyclicNoamLR,
}


def init_lr_schedule(lr_name, **kwargs):
    """Partially initialise the schedule; the optimiser completes it (lr_schedules.py:28-30)."""
    return partial(SUPPORTED[lr_name], **kwargs)
