"""Stand-in for the dense-convolution kernels on HOST tensors, for CPU-only tests of the host logic around
them (module graphs, state dicts, losses): torch's own convolution.  Test infrastructure — installed by
tests/conftest.py through detmatch_amd.dense_conv.HOST_TENSOR_HOOK; CUDA tensors always take the HIP kernels."""
import torch.nn.functional as F


def conv2d(x, weight, bias, stride, padding, relu, w_scale, residual):
    w = weight if w_scale is None else weight * w_scale.view(-1, 1, 1, 1)
    y = F.conv2d(x, w, bias, stride, padding)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def conv_transpose2d(x, weight, k):
    return F.conv_transpose2d(x, weight, None, stride=k)
