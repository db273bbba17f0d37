"""Running mean / variance normalisation of observations (reference
pfrl/nn/empirical_normalization.py:6-107).

Statistics are updated batch-wise with the parallel-variance rule: for a batch of n
values with mean m_x and (biased) variance v_x and the new total count c,

    r = n / c;  d = m_x - mean;  mean += r d;  var += r (v_x - var + d (m_x - mean_new))

``forward(x, update=True)`` learns from x (until ``until`` values have been seen) and
returns ``clip((x - mean) / sqrt(var + eps), +-clip_threshold)``.  All buffers live on
the module's device; nothing is read back to the host except the ``until`` check.

With ``torch.distributed`` initialised (env-sharded data parallelism) the batch
that is folded in is the UNION of the ranks' batches: one all-reduce of (n, sum x, sum x^2) in
float64, then the same update rule -- every rank keeps the statistics a single process would
have after seeing the concatenated batch.  ``sync_across_ranks=False`` keeps them rank-local.
"""
import numpy as np
import torch
from torch import nn


class EmpiricalNormalization(nn.Module):
    def __init__(self, shape, batch_axis=0, eps=1e-2, dtype=np.float32, until=None,
                 clip_threshold=None):
        super().__init__()
        dtype = np.dtype(dtype)
        self.batch_axis = batch_axis
        self.eps = dtype.type(eps)
        self.until = until
        self.clip_threshold = clip_threshold
        expand = lambda a: torch.tensor(np.expand_dims(a, batch_axis))
        self.register_buffer("_mean", expand(np.zeros(shape, dtype=dtype)))
        self.register_buffer("_var", expand(np.ones(shape, dtype=dtype)))
        self.register_buffer("count", torch.tensor(0))
        self._cached_std_inverse = None
        self.sync_across_ranks = True

    @property
    def mean(self):
        return torch.squeeze(self._mean, self.batch_axis).clone()

    @property
    def std(self):
        return torch.sqrt(torch.squeeze(self._var, self.batch_axis)).clone()

    @property
    def _std_inverse(self):
        if self._cached_std_inverse is None:
            self._cached_std_inverse = (self._var + self.eps) ** -0.5
        return self._cached_std_inverse

    def experience(self, x):
        """Fold the batch ``x`` into the running statistics."""
        if self.until is not None and self.count >= self.until:
            return
        n = x.shape[self.batch_axis]
        from pfrl_amd import distributed

        if self.sync_across_ranks and distributed.world_size() > 1:
            n, mean_x, var_x = self._union_batch_stats(x, n)   # collective: every rank calls it
        elif n == 0:
            return
        else:
            var_x, mean_x = torch.var_mean(x, dim=self.batch_axis, keepdim=True, unbiased=False)
        self.count += n
        rate = n / self.count.float()
        delta = mean_x - self._mean
        self._mean += rate * delta
        self._var += rate * (var_x - self._var + delta * (mean_x - self._mean))
        self._cached_std_inverse = None

    def _union_batch_stats(self, x, n):
        """(count, mean, biased variance) of all ranks' batches together, as tensors."""
        x64 = x.double()
        s1 = x64.sum(dim=self.batch_axis, keepdim=True)
        s2 = (x64 * x64).sum(dim=self.batch_axis, keepdim=True)
        acc = torch.cat([s1.reshape(-1), s2.reshape(-1),
                         torch.full((1,), float(n), dtype=torch.float64, device=x.device)])
        from pfrl_amd import distributed

        distributed.control_all_reduce_sum(acc)
        d = s1.numel()
        total = acc[-1].clamp(min=1.0)
        mean = (acc[:d] / total).view_as(s1)
        var = (acc[d:2 * d] / total).view_as(s1) - mean * mean
        return acc[-1].to(self.count.dtype), mean.to(x.dtype), var.clamp(min=0.0).to(x.dtype)

    def forward(self, x, update=True):
        if update:
            self.experience(x)
        y = (x - self._mean) * self._std_inverse
        if self.clip_threshold is not None:
            y = torch.clamp(y, -self.clip_threshold, self.clip_threshold)
        return y

    def inverse(self, y):
        return y * torch.sqrt(self._var + self.eps) + self._mean
