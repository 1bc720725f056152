"""mmdet3d/ops/spconv/functional.py:20-83 — autograd Functions (same signatures)."""
from torch.autograd import Function

from . import ops


class SparseConvFunction(Function):

    @staticmethod
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        ctx.save_for_backward(indice_pair_num, features, filters)
        ctx.indice_pairs = indice_pairs  # keeps the native tables attached to it alive
        ctx.weight = filters             # the Parameter itself (its gradient may be delivered at the end of the pass)
        return ops.indice_conv(features, filters, indice_pairs, indice_pair_num,
                               num_activate_out, False, False)

    @staticmethod
    def backward(ctx, grad_output):
        indice_pair_num, features, filters = ctx.saved_tensors
        input_bp, filters_bp = ops.indice_conv_backward(
            features, filters, grad_output, ctx.indice_pairs, indice_pair_num, False, False,
            need_input_grad=ctx.needs_input_grad[0], defer_weight=ctx.weight if ctx.needs_input_grad[1] else None)
        return input_bp, filters_bp, None, None, None


class SubMConvFunction(Function):

    @staticmethod
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        ctx.save_for_backward(indice_pair_num, features, filters)
        ctx.indice_pairs = indice_pairs
        ctx.weight = filters
        return ops.indice_conv(features, filters, indice_pairs, indice_pair_num,
                               num_activate_out, False, True)

    @staticmethod
    def backward(ctx, grad_output):
        indice_pair_num, features, filters = ctx.saved_tensors
        input_bp, filters_bp = ops.indice_conv_backward(
            features, filters, grad_output, ctx.indice_pairs, indice_pair_num, False, True,
            need_input_grad=ctx.needs_input_grad[0], defer_weight=ctx.weight if ctx.needs_input_grad[1] else None)
        return input_bp, filters_bp, None, None, None


indice_conv = SparseConvFunction.apply
indice_subm_conv = SubMConvFunction.apply
