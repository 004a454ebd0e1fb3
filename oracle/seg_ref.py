"""TEST INFRASTRUCTURE — CPU restatement of the reference's drivable-area segmentation loss (models/seg_loss.py).
Only tests/, __graft_entry__.smoke() and bench.py's CPU-baseline leg may import this.
Pinned by tests/golden/seg_loss.npz and seg_net_*.npz (tools/gen_golden_seg.py, real reference)."""
import torch


class _PassThroughSigmoid(torch.autograd.Function):
    """seg_loss.py:15-32: forward 1/(1+exp(-x)); backward hands the incoming gradient through UNCHANGED
    (no sigma' factor)."""

    @staticmethod
    def forward(ctx, x):
        return 1.0 / (1.0 + torch.exp(-x))

    @staticmethod
    def backward(ctx, g):
        return g.clone()


def seg_loss(head_nchw, seg_maps_nhwc):
    """seg_loss.py:51-76 (training branch).  head [N,C,h,w] raw logits, seg_maps [N,h,w,C].
    -> (loss * 0.05 as a tensor with grad, mean sigmoid where truth >= 0.5, mean sigmoid where truth < 0.5) — the two
    means are python floats; the mean of an empty selection is NaN like torch.mean's."""
    truth = seg_maps_nhwc.clone().permute(0, 3, 1, 2)
    out = _PassThroughSigmoid.apply(head_nchw)
    obj = torch.masked_select(out, truth >= 0.5)
    no_obj = torch.masked_select(out, truth < 0.5)
    weights = torch.ones_like(head_nchw)
    sq = (out - truth) ** 2                                                      # :41-47 weighted_mse_loss
    loss = torch.sum(sq * weights / torch.sum(weights))
    return loss * 0.05, torch.mean(obj).item(), torch.mean(no_obj).item()


def seg_eval(head_nchw):
    """seg_loss.py:77-80 (eval branch): sigmoid of image 0 only, as a numpy array [C,h,w]."""
    return (1.0 / (1.0 + torch.exp(-head_nchw)))[0].detach().cpu().numpy()
