"""Host-side scalar schedules of the training driver (FusionDynMM/train.py:120-128, 189, 195-197;
src/utils.py:203-214).  Pure Python — nothing here touches the device."""
import math


class ExpDecayTemp:
    """start_t * b**epoch with b = exp(log(end_t/start_t)/time_len); b = 1 when time_len == 0.
    Not clamped: the reference keeps decaying after `time_len` epochs."""

    def __init__(self, start_t, end_t, time_len):
        self.start_t, self.end_t, self.time_len = start_t, end_t, time_len
        self.b = 1 if time_len == 0 else math.exp(1 / time_len * math.log(end_t / start_t))

    def get_t(self, epoch):
        return self.start_t * self.b ** epoch


def one_cycle_lr(epoch, total_steps, max_lr, div_factor=25.0, pct_start=0.1, final_div_factor=1e4):
    """torch.optim.lr_scheduler.OneCycleLR (cos anneal, two phases) evaluated at step `epoch` — the
    reference steps it once per EPOCH with total_steps = epochs (train.py:119-128, 267)."""
    initial_lr = max_lr / div_factor
    min_lr = initial_lr / final_div_factor
    up_end = float(pct_start * total_steps) - 1.0
    down_end = float(total_steps) - 1.0

    def cos(start, end, pct):
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1.0)

    if epoch <= up_end or up_end <= 0 and epoch <= 0:
        pct = epoch / up_end if up_end > 0 else 1.0
        return cos(initial_lr, max_lr, pct)
    pct = (epoch - up_end) / (down_end - up_end)
    return cos(max_lr, min_lr, min(pct, 1.0))


def one_cycle_momentum(epoch, total_steps, base_momentum=0.85, max_momentum=0.95, pct_start=0.1):
    """Momentum (SGD) / beta1 (Adam) that the reference's OneCycleLR writes into the optimizer at the same
    step: it is built with the default cycle_momentum=True (train.py:120-128), which OVERWRITES the optimizer's
    momentum every scheduler step with the mirror image of the lr curve — max_momentum at the start, down to
    base_momentum at peak lr, back up to max_momentum (so --momentum is effectively ignored by the reference)."""
    up_end = float(pct_start * total_steps) - 1.0
    down_end = float(total_steps) - 1.0

    def cos(start, end, pct):
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1.0)

    if epoch <= up_end or up_end <= 0 and epoch <= 0:
        pct = epoch / up_end if up_end > 0 else 1.0
        return cos(max_momentum, base_momentum, pct)
    pct = (epoch - up_end) / (down_end - up_end)
    return cos(base_momentum, max_momentum, min(pct, 1.0))


def scaled_lr(lr, batch_size):
    """train.py:46-49: the CLI learning rate refers to batch 8 (use the GLOBAL batch under DP)."""
    return lr if batch_size == 8 else lr * batch_size / 8
