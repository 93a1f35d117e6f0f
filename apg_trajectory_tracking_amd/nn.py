"""`torch.nn.Linear` with the weight gradient computed by the library
(apg_linear_wgrad) - for the policy layers that run in PyTorch (any policy on
the row-layout path, the per-step fallbacks, architectures the fused kernels
are not built for).

Autograd's own backward of `F.linear` evaluates dW = dY^T X through rocBLAS,
which has no good kernel for this tall-skinny reduction (B = 65 536 rows,
M, N <= 256: 190-210 us per layer on an MI355X, the weight gradients of a
three-layer MLP policy cost 0.6 ms next to an 8.8 us rollout).  Here the
forward and dL/dx stay with rocBLAS (well served), dW and db are one split-K
pass over dY and X on the fp32 matrix instruction (exact fp32).

`Linear` is a drop-in subclass: same parameters, same state_dict keys, same
results; on CPU tensors (oracle-side tests, checkpoint conversion) it IS
torch.nn.Linear.  The package's own model classes use it, and the trainers
give an arbitrary user policy the same treatment: `use_apg_linear(net)`
(TrainBase.init_optimizer, opt out with `trainer.swap_linear = False`) turns
every plain `torch.nn.Linear` leaf into this class in place - same Parameter
objects, same state_dict keys."""
import torch
import torch.nn as nn

from . import _capi


class _LinearWgrad(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        # once_differentiable: the raw-pointer kernel below is invisible to
        # autograd, so double backward / create_graph=True raises instead of
        # silently returning wrong higher-order gradients
        x, weight = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad
        need_b = need_b and ctx.has_bias
        gx = (grad_out.matmul(weight.to(grad_out.dtype)).to(x.dtype)
              if need_x else None)
        gw = gb = None
        if need_w or need_b:
            M, N = weight.shape
            dy = grad_out.reshape(-1, M)
            xx = x.reshape(-1, N)
            if (dy.dtype != torch.float32 or xx.dtype != torch.float32
                    or not dy.is_cuda or dy.numel() * 4 >= _MAX_OPERAND_BYTES
                    or xx.numel() * 4 >= _MAX_OPERAND_BYTES):
                # what the kernel is not built for (another dtype, an operand of
                # 4 GiB or more: 32-bit byte offsets): autograd's own formulas
                # (mixed dtypes - autocast hands a half grad_out to a float x:
                # the product in the wider of the two, as autograd's linear does)
                wide = torch.promote_types(dy.dtype, xx.dtype)
                gw = (dy.to(wide).t().matmul(xx.to(wide)).to(weight.dtype)
                      if need_w else None)
                gb = dy.sum(0).to(weight.dtype) if need_b else None
                return gx, gw, gb
            dy, xx = dy.contiguous(), xx.contiguous()
            lib = _capi.lib()
            gw = torch.empty_like(weight, memory_format=torch.contiguous_format)
            gb = torch.empty(M, dtype=torch.float32, device=dy.device) if need_b else None
            ws = torch.empty(max(1, lib.apg_linear_wgrad_workspace_floats(M, N)),
                             dtype=torch.float32, device=dy.device)
            _capi.check(lib.apg_linear_wgrad(
                _capi.ptr(dy), _capi.ptr(xx), dy.shape[0], M, N, _capi.ptr(gw),
                _capi.ptr(gb), _capi.ptr(ws), _capi.stream_of(dy)), "apg_linear_wgrad")
            if not need_w:
                gw = None
        return gx, gw, gb


_MAX_OPERAND_BYTES = 1 << 32


def linear(x, weight, bias=None):
    """F.linear with the library's weight gradient on the GPU (fp32)."""
    if (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
            and torch.is_grad_enabled() and (weight.requires_grad or (
                bias is not None and bias.requires_grad))):
        return _LinearWgrad.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


class Linear(nn.Linear):
    """torch.nn.Linear whose dW / db come from apg_linear_wgrad on the GPU."""

    def forward(self, x):
        return linear(x, self.weight, self.bias)


def use_apg_linear(module):
    """Turn every plain `torch.nn.Linear` in `module` (exact type: subclasses
    keep their own forward) into `Linear`, IN PLACE: the class of the layer
    object is switched, so parameters, buffers, hooks, state_dict keys and
    every reference an optimizer holds stay what they were.  Returns the
    number of layers switched."""
    n = 0
    for m in module.modules():
        if type(m) is nn.Linear:
            m.__class__ = Linear
            n += 1
    return n
