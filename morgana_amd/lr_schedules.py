"""Learning-rate schedules of the training loop: host-side scalars, nothing here touches the device.

Same names and constructor arguments as ``morgana.lr_schedules`` (DummyLR / NoamLR / CyclicNoamLR, ``init_lr_schedule``,
reference lr_schedules.py:28-145) so experiment configs carry over, but built the other way round: every schedule of the
family is ONE closed-form multiplier, ``noam_scale(step, warmup, cycle)``, handed to torch's ``LambdaLR``.  The values are
pinned to the reference's by tests/golden/g9_ema_lr.npz (tests/test_host_logic.py).
"""
import functools
import math

from torch.optim import lr_scheduler

# which schedules ``train_epoch`` advances per batch and which ``run_train`` advances per epoch (experiment_builder.py:477, :559)
BATCH_LR_SCHEDULES = ('cyclic', 'noam', 'cyclic_noam')
EPOCH_LR_SCHEDULES = ('constant', 'lambda', 'step', 'multi_step', 'exponential', 'cosine_annealing',
                      'cosine_annealing_warm_restarts')


def noam_scale(step, warmup, cycle=None):
    """Multiplier on the base learning rate after ``step`` scheduler steps.

    Linear warm-up to 1 at ``step == warmup``, inverse-square-root decay after it (Vaswani et al. 2017 with the peak
    normalised to 1): sqrt(warmup) * min(1/sqrt(s), s / warmup^1.5).  ``cycle`` restarts the pattern every ``cycle`` steps;
    step 0 - and with it the first step of every cycle - counts as step 1.
    """
    s = step if cycle is None else step % cycle
    s = s if s > 1 else 1
    rise = s * warmup ** -1.5
    fall = s ** -0.5
    return math.sqrt(warmup) * (rise if rise < fall else fall)


def cycle_length(warmup, trigger):
    """Steps until the decaying branch sqrt(warmup / s) has fallen to ``trigger``: s = warmup / trigger^2 (truncated)."""
    return int((trigger / math.sqrt(warmup)) ** -2)


class _ScaledLR(lr_scheduler.LambdaLR):
    """LambdaLR whose multiplier is a picklable partial of a module-level function (LambdaLR.state_dict skips lambdas)."""

    def __init__(self, optimizer, multiplier):
        self.multiplier = multiplier
        super(_ScaledLR, self).__init__(optimizer, multiplier)


def _unit(step):
    return 1.0


class DummyLR(_ScaledLR):
    """The 'constant' schedule: the base learning rate at every step."""

    def __init__(self, optimizer):
        super(DummyLR, self).__init__(optimizer, _unit)


class NoamLR(_ScaledLR):
    def __init__(self, optimizer, warmup_steps=4000):
        self.warmup_steps = warmup_steps
        super(NoamLR, self).__init__(optimizer, functools.partial(noam_scale, warmup=warmup_steps))


class CyclicNoamLR(_ScaledLR):
    """Noam restarted every ``cycle_steps`` batches; by default when the decay has reached ``cycle_trigger`` of the peak."""

    def __init__(self, optimizer, warmup_steps=4000, cycle_trigger=0.2, cycle_steps=None):
        self.warmup_steps = warmup_steps
        self.cycle_steps = cycle_length(warmup_steps, cycle_trigger) if cycle_steps is None else cycle_steps
        super(CyclicNoamLR, self).__init__(optimizer, functools.partial(noam_scale, warmup=warmup_steps, cycle=self.cycle_steps))


_TORCH_SCHEDULES = dict(lambda_='LambdaLR', step='StepLR', multi_step='MultiStepLR', exponential='ExponentialLR',
                        cosine_annealing='CosineAnnealingLR', plateau='ReduceLROnPlateau', cyclic='CyclicLR',
                        cosine_annealing_warm_restarts='CosineAnnealingWarmRestarts')


def schedule_class(lr_name):
    """Schedule class for a ``--lr_schedule_name`` value: the three above, or torch's own for the other names."""
    own = {'constant': DummyLR, 'noam': NoamLR, 'cyclic_noam': CyclicNoamLR}
    if lr_name in own:
        return own[lr_name]
    key = 'lambda_' if lr_name == 'lambda' else lr_name
    if key not in _TORCH_SCHEDULES:
        raise KeyError('unknown lr schedule %r (known: %s)' % (lr_name, sorted(list(own) + [k.rstrip('_') for k in _TORCH_SCHEDULES])))
    return getattr(lr_scheduler, _TORCH_SCHEDULES[key])


def init_lr_schedule(lr_name, **kwargs):
    """Schedule with everything bound but the optimiser, which ``run_train`` supplies once it exists."""
    return functools.partial(schedule_class(lr_name), **kwargs)
