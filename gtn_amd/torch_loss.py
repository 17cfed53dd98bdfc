"""PyTorch entry point of the hot path.

`ctc_loss` is the device-resident counterpart of the reference's
bindings/python/examples/pytorch_loss.py:19-102: the emissions tensor never leaves
the GPU (no inputs.cpu(), no per-sample weights_to_numpy), the batch runs through
the batched graph functions, and the emission gradients come back as one tensor.
"""
import torch

import gtn_amd as gtn


def ctc_target_graph(target, blank=0):
    """the target acceptor of benchmarks/ctc.cpp:40-58 / pytorch_loss.py's criterion"""
    L = 2 * len(target) + 1
    g = gtn.Graph(False)
    for l in range(L):
        idx = (l - 1) // 2
        g.add_node(l == 0, l == L - 1 or l == L - 2)
        label = target[idx] if l % 2 else blank
        g.add_arc(l, l, label)
        if l > 0:
            g.add_arc(l - 1, l, label)
        if l % 2 and l > 1 and label != target[idx - 1]:
            g.add_arc(l - 2, l, label)
    g.arc_sort()
    return g


class _CTCLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_probs, targets, blank, reduction):
        assert log_probs.is_cuda and log_probs.dtype == torch.float32 and log_probs.dim() == 3
        B, T, C = log_probs.shape
        x = log_probs.contiguous()
        stream = torch.cuda.current_stream(x.device)
        gtn.set_stream(stream.cuda_stream if stream.cuda_stream else None)
        if not stream.cuda_stream:
            torch.cuda.current_stream(x.device).synchronize()  # engine runs on its own stream
        ems = gtn.linear_graph_n(B, T, C, x, calc_grad=log_probs.requires_grad)
        tgs = [ctc_target_graph(list(t), blank) for t in targets]
        losses = gtn.subtract(gtn.forward_score(ems), gtn.forward_score(gtn.intersect(tgs, ems)))
        out = torch.empty(B, dtype=torch.float32, device=x.device)
        gtn.items_to_device(losses, out)
        if not stream.cuda_stream:
            gtn.synchronize()
        ctx.graphs = (losses, ems)
        ctx.shape = (B, T, C)
        ctx.reduction = reduction
        if reduction == "mean":
            return out.mean()
        if reduction == "sum":
            return out.sum()
        return out

    @staticmethod
    def backward(ctx, grad_out):
        losses, ems = ctx.graphs
        B, T, C = ctx.shape
        gtn.backward(losses)
        grad = torch.empty(B, T, C, dtype=torch.float32, device=grad_out.device)
        gtn.grads_to_device(ems, grad, [b * T * C for b in range(B)])
        gtn.synchronize()
        if ctx.reduction == "mean":
            scale = (grad_out / B).reshape(1, 1, 1)
        elif ctx.reduction == "sum":
            scale = grad_out.reshape(1, 1, 1)
        else:
            scale = grad_out.reshape(B, 1, 1)
        return grad * scale, None, None, None


def ctc_loss(log_probs, targets, blank=0, reduction="none"):
    """log_probs: float32 CUDA tensor [B, T, C] (any scores; the loss carries its own
    normaliser forwardScore(emissions), as in benchmarks/ctc.cpp:150-158).
    targets: sequence of B label sequences.  Returns per-utterance losses (or their
    mean / sum); differentiable w.r.t. log_probs."""
    return _CTCLoss.apply(log_probs, targets, blank, reduction)
