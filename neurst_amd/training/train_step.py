"""One optimizer step of the hot loop (GradAccumKerasModel.train_step / fit loop,
neurst/training/gradaccum_keras_model.py:162-260, 437-477; hvd optimizer hook hvd_utils.py:85-91):

    forward -> label-smoothed CE -> backward (bucketed RCCL all-reduce overlapped on a side stream)
            -> [update_cycle-1 more micro batches accumulated] -> average [-> clip by value / per-tensor norm] -> fused Adam

No host<->device synchronisation happens inside a step; the loss is returned as a device scalar.
"""
import torch


class TrainStep(object):
    def __init__(self, model, criterion, optimizer, reducer=None, update_cycle=1, clip_value=None, clip_norm=None):
        self.model, self.criterion, self.optimizer, self.reducer = model, criterion, optimizer, reducer
        # gradaccum_keras_model.py:228-233: clip_value takes precedence over clip_norm; both act on the averaged gradients
        self.clip_value = clip_value if clip_value else None
        self.clip_norm = None if self.clip_value else (clip_norm if clip_norm else None)
        self.update_cycle = max(1, int(update_cycle))
        if reducer is not None:
            model.grad_ready_hook = self._hook
            ws = getattr(model.rt, "wgrad_stream", None)
            if ws is not None and ws not in reducer.extra_streams:
                reducer.extra_streams.append(ws)
        self._last_micro = True

    def _hook(self, prefixes):
        if self._last_micro and self.reducer is not None:
            self.reducer.component_ready(prefixes)

    def __call__(self, batches):
        """batches: one model-input dict, or a list of `update_cycle` dicts (gradient accumulation: the mean of
        the micro-batch gradients, GradientAccumulator semantics gradaccum_keras_model.py:62-109)."""
        if isinstance(batches, dict):
            batches = [batches]
        n = len(batches)
        loss_sum = None
        for i, inputs in enumerate(batches):
            self._last_micro = (i == n - 1)
            logits = self.model(inputs, is_training=True)
            loss = self.criterion.reduce_loss(inputs, logits)
            dlogits = self.criterion.backward(loss_scale=1.0 / n)
            del logits
            self.model.backward(dlogits, accumulate=(i > 0))
            loss_sum = loss if loss_sum is None else loss_sum + loss
        scale = self.reducer.finish() if self.reducer is not None else 1.0
        if self.clip_value or self.clip_norm:
            from neurst_amd import kernels as K
            table, nentries, seg_first, nseg = self.model.store.clip_tables()
            K.grad_clip(self.model.store.grad, table, nentries, seg_first, nseg, pre_scale=scale,
                        clip_value=self.clip_value, clip_norm=self.clip_norm)
            scale = 1.0   # the average is already applied
        self.optimizer.apply_gradients(grad_scale=scale)
        self.model.rt.step += 1
        return loss_sum / n
