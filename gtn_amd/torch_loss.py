"""PyTorch entry points of the hot path: `ctc_loss` and `asg_loss`.

`ctc_loss` is the device-resident counterpart of the reference's
bindings/python/examples/pytorch_loss.py:19-102: the emissions tensor never leaves
the GPU (no inputs.cpu(), no per-sample weights_to_numpy), the batch runs through
the batched graph functions, and the emission gradients come back as one tensor.
"""
import ctypes as C
import os

import numpy as np
import torch

import gtn_amd as gtn

_NATIVE = None


def _native():
    """libgtn_criteria.so (gtn_amd/criteria/): the whole batch step in one native call --
    target graphs on host threads, batched graph functions, loss and gradient on the device"""
    global _NATIVE
    if _NATIVE is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libgtn_criteria.so")
        if os.path.exists(path) and not os.environ.get("GTN_AMD_PYTHON_CRITERIA"):
            lib = C.CDLL(path)
            lib.gtn_ctc_loss_n.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p]
            lib.gtn_ctc_loss_n.restype = C.c_int
            lib.gtn_asg_loss_n.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            lib.gtn_asg_loss_n.restype = C.c_int
            lib.gtn_criteria_last_error.restype = C.c_char_p
            _NATIVE = lib
        else:
            _NATIVE = False
    return _NATIVE


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
        if len(targets) != B:
            raise ValueError(f"ctc_loss: {len(targets)} target sequences for a batch of {B}")
        x = log_probs.contiguous()
        stream = torch.cuda.current_stream(x.device)
        gtn.set_stream(stream.cuda_stream if stream.cuda_stream else None)
        if not stream.cuda_stream:
            torch.cuda.current_stream(x.device).synchronize()  # engine runs on its own stream
        lib = _native()
        if lib:
            flat, lens = _flat_targets(targets)
            out = torch.empty(B, dtype=torch.float32, device=x.device)
            grad = torch.empty(B, T, C, dtype=torch.float32, device=x.device) if log_probs.requires_grad else None
            rc = lib.gtn_ctc_loss_n(x.data_ptr(), flat.ctypes.data, lens.ctypes.data, B, T, C, int(blank),
                                    out.data_ptr(), grad.data_ptr() if grad is not None else None)
            if rc != 0:
                raise RuntimeError(lib.gtn_criteria_last_error().decode())
            if not stream.cuda_stream:
                gtn.synchronize()
            ctx.graphs = None
            ctx.grad = grad
            ctx.shape = (B, T, C)
            ctx.reduction = reduction
            return out.mean() if reduction == "mean" else (out.sum() if reduction == "sum" else out)
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
        B, T, C = ctx.shape
        if ctx.graphs is None:
            grad = ctx.grad  # computed with the forward pass by the native criterion
        else:
            losses, ems = ctx.graphs
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


def _flat_targets(targets):
    flat = np.ascontiguousarray(np.concatenate([np.asarray(t, np.int32).reshape(-1) for t in targets])
                                if len(targets) else np.zeros(0, np.int32), dtype=np.int32)
    return flat, np.ascontiguousarray([len(t) for t in targets], dtype=np.int32)


class _ASGLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emissions, transitions, start, targets, reduction):
        assert emissions.is_cuda and emissions.dtype == torch.float32 and emissions.dim() == 3
        B, T, N = emissions.shape
        assert transitions.shape == (N, N) and start.shape == (N,)
        if len(targets) != B:
            raise ValueError(f"asg_loss: {len(targets)} target sequences for a batch of {B}")
        lib = _native()
        if not lib:
            raise RuntimeError("asg_loss needs gtn_amd/lib/libgtn_criteria.so (run __graft_entry__.build())")
        x = emissions.contiguous()
        # arc order of gtn::criteria::asgTransitions: N start arcs, then arc N + i*N + j = j -> i
        w = torch.cat([start.reshape(-1), transitions.reshape(-1)]).to(torch.float32).contiguous()
        stream = torch.cuda.current_stream(x.device)
        gtn.set_stream(stream.cuda_stream if stream.cuda_stream else None)
        if not stream.cuda_stream:
            stream.synchronize()
        flat, lens = _flat_targets(targets)
        out = torch.empty(B, dtype=torch.float32, device=x.device)
        gem = torch.empty(B, T, N, dtype=torch.float32, device=x.device) if emissions.requires_grad else None
        need_tr = transitions.requires_grad or start.requires_grad
        gtr = torch.empty(N + N * N, dtype=torch.float32, device=x.device) if need_tr else None
        rc = lib.gtn_asg_loss_n(x.data_ptr(), flat.ctypes.data, lens.ctypes.data, B, T, N, w.data_ptr(),
                                out.data_ptr(), gem.data_ptr() if gem is not None else None,
                                gtr.data_ptr() if gtr is not None else None)
        if rc != 0:
            raise RuntimeError(lib.gtn_criteria_last_error().decode())
        if not stream.cuda_stream:
            gtn.synchronize()
        ctx.grads = (gem, gtr)
        ctx.shape = (B, T, N)
        ctx.reduction = reduction
        ctx.per_utterance = reduction == "none"
        return out.mean() if reduction == "mean" else (out.sum() if reduction == "sum" else out)

    @staticmethod
    def backward(ctx, grad_out):
        B, T, N = ctx.shape
        gem, gtr = ctx.grads
        if ctx.per_utterance:
            # the transition gradient was summed over the batch with unit seeds (as
            # criterion_test.cpp:289-305 accumulates it); per-utterance seeds must be uniform
            scale_em = grad_out.reshape(B, 1, 1)
            scale_tr = grad_out.reshape(-1)[0] if gtr is not None else None
            if gtr is not None and not bool((grad_out == grad_out.reshape(-1)[0]).all()):
                raise RuntimeError("asg_loss(reduction='none'): transition gradients need a uniform upstream "
                                   "gradient; use reduction='sum' or 'mean'")
        else:
            scale_em = (grad_out / B if ctx.reduction == "mean" else grad_out).reshape(1, 1, 1)
            scale_tr = scale_em.reshape(())
        g_em = gem * scale_em if gem is not None else None
        g_tr = g_st = None
        if gtr is not None:
            g_st = gtr[:N] * scale_tr
            g_tr = gtr[N:].reshape(N, N) * scale_tr
        return g_em, g_tr, g_st, None, None


def asg_loss(emissions, transitions, targets, start=None, reduction="none"):
    """The ASG criterion of examples/asg.cpp:30-68 / criterion_test.cpp:182-306 for a batch.
    emissions: float32 CUDA [B, T, N]; transitions: [N, N] with transitions[i, j] the score of
    label j followed by label i; start: [N] scores of the first label (zeros when omitted);
    targets: B label sequences.  Differentiable w.r.t. emissions, transitions and start; the
    full-connect term runs on the symbolic composition (nothing of size T*N*N is stored)."""
    if start is None:
        start = torch.zeros(emissions.shape[-1], dtype=torch.float32, device=emissions.device)
    return _ASGLoss.apply(emissions, transitions, start, targets, reduction)
