"""Linear layer whose weight gradient is a batched split-K GEMM.

The PPO update runs minibatches of 10^5..10^6 rows through 256-wide layers, so the weight
gradient dW = dY^T X is a skinny-output GEMM with a huge reduction dimension (K = batch).
hipBLASLt's pick for that shape reaches 42 TF/s on MI355X (806 us at 262144 x 256 x 256);
reshaping the batch into S independent slices turns it into a batched GEMM that fills all
256 CUs (275 us, 125 TF/s — `tools/micro_linear.py`), followed by a 16 MB sum over slices.
Same parameters, same state_dict keys as nn.Linear; small batches take the stock path.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

_MIN_ROWS = 16384


class _SplitKLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        B, N = dy.shape
        K = x.shape[1]
        S = 256 if N == 1 else 128
        while B % S:
            S //= 2
        dx = dy.mm(weight) if ctx.needs_input_grad[0] else None
        dw = torch.bmm(dy.view(S, B // S, N).transpose(1, 2), x.view(S, B // S, K)).sum(0)
        return dx, dw, dy.sum(0)


class Linear(nn.Linear):
    def forward(self, x):
        if (x.dim() == 2 and x.shape[0] >= _MIN_ROWS and x.is_contiguous() and torch.is_grad_enabled()
                and self.weight.requires_grad and self.bias is not None):
            return _SplitKLinear.apply(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)
