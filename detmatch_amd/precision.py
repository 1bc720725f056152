"""Mixed precision of the DetMatch step — the counterpart of the reference's fp16 configs
(mmdet3d/apis/ssl_train.py:100-105: `cfg.fp16` -> mmcv Fp16OptimizerHook / autocast; BASELINE configs[4]).

`set_mixed(True)` switches the GEMM-shaped kernels of BOTH branches to 16-bit multiplicands with fp32
accumulation while every tensor in HBM, the master weights, the optimizer state and all reductions stay fp32:
  * dense convolutions (csrc/conv2d.hip): v_mfma_f32_32x32x16_bf16, operands rounded on their way into LDS;
  * sparse convolutions (csrc/spconv16.hip, storage mode DM_SP16_F32ROWS): v_mfma_f32_16x16x32_bf16 for the
    forward and input-gradient gather-GEMMs of every layer with >= 16 input channels.
bfloat16 keeps fp32's exponent range, so no loss scaling is needed: `loss_scale` of the reference's fp16 config
is accepted and ignored.  The default — and the headline benchmark — is fp32-CLASS arithmetic: the sparse
gather-GEMMs on the matrix pipe's own fp32 instruction (v_mfma_f32_16x16x4_f32), the dense convolutions on the
three-way bf16 split (six products on v_mfma_f32_32x32x16_bf16, fp32 accumulate: within 1.25x of the fp32
instruction's error against float64, finite inputs; see dense_conv.set_math and DESIGN §6.0) — bench.py prints
the flavour as config.conv_math.
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
    dense convolutions (dense_conv.FP32_DEFAULT, environment DM_FP32_CONV, default fp32_split) and the sparse
    ones (environment DM_FP32_SPCONV, default fp32_mfma: the sparse gather-GEMM is paced by its table -> row
    fetches, the split's extra weight traffic and VALU work cancel what the matrix pipe saves — measured 28 vs
    29 us on the 64 -> 64 layer, slower on the 16-channel ones)."""
    return SPARSE_FP32


SPARSE_FP32 = __import__('os').environ.get('DM_FP32_SPCONV', 'fp32_mfma')


@contextlib.contextmanager
def mixed_precision(on=True):
    prev = mixed()
    set_mixed(on)
    try:
        yield
    finally:
        set_mixed(prev)
