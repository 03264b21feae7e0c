"""neurst/models/model_utils.py: length -> padding masks and text-length deduction (device tensors)."""
import torch

from neurst_amd.utils.compat import PaddingMode


def _on_device(lengths, dtype):
    return torch.is_tensor(lengths) and lengths.is_cuda and dtype == torch.float32


def input_length_to_nonpadding(lengths, max_len, dtype=torch.float32):
    """model_utils.py:44-59: sequence_mask -> 1.0 for non-padding (device lengths: one nst_seq_mask launch)."""
    if _on_device(lengths, dtype):
        from neurst_amd import kernels as K
        return K.seq_mask(lengths, max_len, 1.0, 0.0)
    ar = torch.arange(int(max_len), device=lengths.device)[None, :]
    return (ar < lengths.long()[:, None]).to(dtype)


def input_length_to_padding(lengths, max_len, dtype=torch.float32, halvings=0, stride=2):
    """model_utils.py:62-75: 1.0 for padding.  halvings / stride: lengths after that many stride-`stride` convolutions with
    SAME padding (speech_transformer.py:179-189), folded into the same launch on the device.  The device tensor also carries the
    attention bias padding * FLOAT_MIN (layer_utils.input_padding_to_bias returns it without further launches)."""
    if _on_device(lengths, dtype):
        from neurst_amd import kernels as K
        pad = K.seq_mask(lengths, max_len, 0.0, 1.0, halvings, stride)
        pad._nst_bias = K.seq_mask(lengths, max_len, 0.0, K.FLOAT_MIN, halvings, stride)
        return pad
    for _ in range(int(halvings)):
        lengths = (lengths + stride - 1) // stride
    return 1.0 - input_length_to_nonpadding(lengths, max_len, dtype)


def deduce_text_length(data_tensor, pad_id, padding_mode):
    """model_utils.py:23-41."""
    ne = (data_tensor != pad_id).to(torch.int32)
    if padding_mode == PaddingMode.DEFAULT:
        return ne.sum(dim=1)
    if padding_mode == PaddingMode.EOS_AS_PADDING:
        return torch.argmin(ne, dim=-1) + 1
    raise NotImplementedError
