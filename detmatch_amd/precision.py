"""Mixed precision of the DetMatch step — the counterpart of the reference's fp16 configs
(mmdet3d/apis/ssl_train.py:100-105: `cfg.fp16` -> mmcv Fp16OptimizerHook / autocast; BASELINE configs[4]).

`set_mixed(True)` switches the GEMM-shaped kernels of BOTH branches to 16-bit multiplicands with fp32
accumulation while every tensor in HBM, the master weights, the optimizer state and all reductions stay fp32:
  * dense convolutions (csrc/conv2d.hip): v_mfma_f32_32x32x16_bf16, operands rounded on their way into LDS;
  * sparse convolutions (csrc/spconv16.hip, storage mode DM_SP16_F32ROWS): v_mfma_f32_16x16x32_bf16 for the
    forward and input-gradient gather-GEMMs of every layer with >= 16 input channels.
bfloat16 keeps fp32's exponent range, so no loss scaling is needed: `loss_scale` of the reference's fp16 config
is accepted and ignored.  The default — and the headline benchmark — is exact fp32 everywhere.
(Half-precision STORAGE of sparse features, the reference's `indice_conv_half`, is a property of the tensors
handed to spconv.ops.indice_conv, not of this switch.)"""
import contextlib

_STATE = {'mixed': False}


def set_mixed(on):
    from . import dense_conv
    dense_conv.set_math('bf16' if on else 'fp32')      # 'fp32' = the default fp32 flavour of dense_conv
    _STATE['mixed'] = bool(on)


def mixed():
    return _STATE['mixed']


def sparse_bf16():
    """fp32 sparse features take the bf16-multiplicand gather-GEMM."""
    return _STATE['mixed']


def fp32_flavour():
    """'fp32_mfma' (the matrix pipe's own fp32 instructions) or 'fp32_split' (fp32-class arithmetic from six bf16
    products of three-way split operands — dense_conv.set_math): which kernels serve EXACT-class fp32, for the
    dense and the sparse convolutions alike (dense_conv.FP32_DEFAULT, environment DM_FP32_CONV)."""
    from . import dense_conv
    return dense_conv.FP32_DEFAULT


@contextlib.contextmanager
def mixed_precision(on=True):
    prev = mixed()
    set_mixed(on)
    try:
        yield
    finally:
        set_mixed(prev)
