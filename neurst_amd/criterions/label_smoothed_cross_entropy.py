"""LabelSmoothedCrossEntropy (neurst/criterions/label_smoothed_cross_entropy.py:27-157) on the HIP path.

One fused kernel pass computes, per target position, the smoothed cross entropy minus the normalising constant
times the length mask (no one_hot / softmax tensors are materialised); a second fused pass writes
d(reduce_loss)/d(logits) = (softmax - soft_target) * weight / sum(n_tokens), with 1/sum(n_tokens) read from a device
scalar so the training step never synchronises with the host.
"""
import collections

import torch

from neurst_amd import kernels as K
from neurst_amd.criterions.criterion import Criterion, register_criterion
from neurst_amd.models.model_utils import input_length_to_nonpadding
from neurst_amd.utils.flags_core import Flag

MetricWrapper = collections.namedtuple("MetricWrapper", "flag greater_is_better")


@register_criterion
class LabelSmoothedCrossEntropy(Criterion):
    def __init__(self, args=None):
        super().__init__()
        self._label_smoothing = float((args or {}).get("label_smoothing", 0.) or 0.)
        self._saved = None
        self._n_samples = {}

    @staticmethod
    def class_or_method_args():
        return [Flag("label_smoothing", dtype=Flag.TYPE.FLOAT, default=0., help="The label smoothing constant.")]

    def _weights(self, model_inp, labels):
        padding = model_inp.get("trg_padding", None)
        if padding is None:
            padding = model_inp.get("padding", None)
        length = model_inp.get("trg_length", None)
        if length is None:
            length = model_inp.get("length", None)
        if padding is None:
            weights = input_length_to_nonpadding(length, labels.shape[1], torch.float32)
        else:
            weights = (1 - padding).float()
        if model_inp.get("mask", None) is not None:
            weights = weights * model_inp["mask"].float()
        return weights

    def __call__(self, model_inp, model_out):
        """-> (nll_sum [B], n_samples [1], n_tokens [B]), float32 device tensors."""
        logits = model_out["logits"] if isinstance(model_out, dict) else model_out
        if not torch.is_tensor(logits):
            raise ValueError("Not supported type of model_out: {}".format(type(model_out)))
        labels = model_inp["trg"].long().contiguous()
        B, L, V = logits.shape
        weights = self._weights(model_inp, labels).contiguous()
        l2 = logits.reshape(B * L, V)
        xent, lse = K.ls_xent_fwd(l2, labels.view(-1), weights.view(-1), self._label_smoothing)
        # the four reductions (per-sample sums, the batch loss, 1 / tokens for the backward kernel) in one launch
        nll_sum, n_tokens, loss, inv_tokens = K.xent_reduce(xent, weights)
        key = (B, str(logits.device))
        if key not in self._n_samples:
            self._n_samples[key] = torch.full((1,), float(B), dtype=torch.float32, device=logits.device)
        self._saved = (l2, labels.view(-1), weights.view(-1), lse, inv_tokens, (B, L, V))
        self._loss = loss
        return nll_sum, self._n_samples[key], n_tokens

    def reduce_loss(self, model_inp, model_out):
        """sum(nll_sum) / sum(n_tokens)  (label_smoothed_cross_entropy.py:46-53), a device scalar."""
        self(model_inp, model_out)
        return self._loss[0]

    def backward(self, loss_scale=1.0, loss_scale_dev=None):
        """d(reduce_loss * loss_scale)/d(logits) for the logits of the last __call__; [B, L, V] in the logits dtype.
        loss_scale_dev: an additional factor held in a device scalar (the dynamic loss scale)."""
        l2, labels, weights, lse, inv, (B, L, V) = self._saved
        self._saved = None
        if loss_scale_dev is not None:
            inv = inv * loss_scale_dev.reshape(1)
        inv = inv.contiguous()
        # the gradient keeps the logits' row stride (rows padded to whole 128-byte lines, text_modalities.py); its padding columns
        # are allocated too and ZERO, so that the consumer may run its products over a vocabulary rounded up to 8 columns
        stride = l2.stride(0)
        if stride > V and l2.stride(1) == 1:
            buf = torch.empty(l2.shape[0], stride, dtype=l2.dtype, device=l2.device)
            buf[:, V:].zero_()
            out = buf[:, :V]
        else:
            out = torch.empty_strided(l2.shape, l2.stride(), dtype=l2.dtype, device=l2.device)
        g = K.ls_xent_bwd(l2, labels, weights, lse, self._label_smoothing, float(loss_scale), out=out, gscale_dev=inv).view(B, L, V)
        if stride > V and l2.stride(1) == 1:
            g._nst_zero_padded = stride      # columns [V, stride) of every row exist and are zero
        return g

    def reduce_metrics(self, eval_res_list):
        nll_sum = nll_samples = nll_tokens = 0.
        for _nll_sum, _nll_samples, _nll_tokens in eval_res_list:
            nll_sum += float(torch.as_tensor(_nll_sum).sum())
            nll_samples += float(torch.as_tensor(_nll_samples).sum())
            nll_tokens += float(torch.as_tensor(_nll_tokens).sum())
        return {"NLL": nll_sum / nll_samples, "PPL": 2. ** (nll_sum / nll_tokens)}

    def reduce_sample_metrics(self, eval_res):
        nll_sum, _, nll_tokens = eval_res
        return [{"nll": float(n), "ppl": 2. ** (float(n) / float(t)), "nll_per_token": float(n) / float(t)}
                for n, t in zip(nll_sum.tolist(), nll_tokens.tolist())]

    def as_metric(self):
        return MetricWrapper(flag="NLL", greater_is_better=False)
