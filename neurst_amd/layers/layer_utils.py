"""neurst/layers/layer_utils.py restated for the HIP path.

The reference materialises additive attention bias tensors (padding*FLOAT_MIN [B,T]; FLOAT_MIN*(1-tril) [1,1,L,L]).
Here the key-padding bias keeps its [B,T] form (it is read by the attention kernel), while the causal bias is
generated inside the kernel from the (query, key) indices -- same values, no L x L tensor.
"""
from neurst_amd.kernels import FLOAT_MIN


def input_padding_to_bias(input_padding):
    """layer_utils.py:19-32: bias = padding * FLOAT_MIN, shape [batch, max_length], float32.  A padding tensor built on the
    device from lengths (model_utils.input_length_to_padding) already carries it."""
    bias = getattr(input_padding, "_nst_bias", None)
    if bias is not None and bias.shape == input_padding.shape:
        return bias
    return (input_padding.float() * FLOAT_MIN).contiguous()


def lower_triangle_attention_bias(length, device=None):
    """layer_utils.py:35-53 (dense form, for tests/debugging only; the kernels take causal=True instead)."""
    import torch
    tril = torch.tril(torch.ones(length, length, device=device))
    return (FLOAT_MIN * (1.0 - tril)).reshape(1, 1, length, length)
