from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from dsgn import _C


class _SigmoidFocalLoss(Function):
    @staticmethod
    def forward(ctx, logits, targets, gamma, alpha):
        ctx.save_for_backward(logits, targets)
        ctx.num_classes, ctx.gamma, ctx.alpha = logits.shape[1], gamma, alpha
        return _C.sigmoid_focalloss_forward(logits, targets, ctx.num_classes, gamma, alpha)

    @staticmethod
    @once_differentiable
    def backward(ctx, d_loss):
        logits, targets = ctx.saved_tensors
        d_logits = _C.sigmoid_focalloss_backward(logits, targets, d_loss.contiguous(), ctx.num_classes, ctx.gamma, ctx.alpha)
        return d_logits, None, None, None


sigmoid_focal_loss_cuda = _SigmoidFocalLoss.apply


class SigmoidFocalLoss(nn.Module):
    def __init__(self, gamma, alpha):
        super().__init__()
        self.gamma, self.alpha = gamma, alpha

    def forward(self, logits, targets):
        return sigmoid_focal_loss_cuda(logits, targets, self.gamma, self.alpha).sum()
