"""neurst/models/model_utils.py: length -> padding masks and text-length deduction (device tensors)."""
import torch

from neurst_amd.utils.compat import PaddingMode


def input_length_to_nonpadding(lengths, max_len, dtype=torch.float32):
    """model_utils.py:44-59: sequence_mask -> 1.0 for non-padding."""
    ar = torch.arange(int(max_len), device=lengths.device)[None, :]
    return (ar < lengths.long()[:, None]).to(dtype)


def input_length_to_padding(lengths, max_len, dtype=torch.float32):
    """model_utils.py:62-75: 1.0 for padding."""
    return 1.0 - input_length_to_nonpadding(lengths, max_len, dtype)


def deduce_text_length(data_tensor, pad_id, padding_mode):
    """model_utils.py:23-41."""
    ne = (data_tensor != pad_id).to(torch.int32)
    if padding_mode == PaddingMode.DEFAULT:
        return ne.sum(dim=1)
    if padding_mode == PaddingMode.EOS_AS_PADDING:
        return torch.argmin(ne, dim=-1) + 1
    raise NotImplementedError
