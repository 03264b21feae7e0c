"""Keras-semantics Adam over the flat parameter buffer (the optimizer the reference's hparams sets select,
neurst/models/speech_transformer.py:265-279; neurst/optimizers/__init__.py registers tf.keras Adam).

    lr_t = lr * sqrt(1 - beta_2^t) / (1 - beta_1^t);  m, v EMAs;  p -= lr_t * m / (sqrt(v) + epsilon)

One fused kernel launch updates fp32 master weights, both moments and the bf16 compute shadow for all parameters;
the data-parallel 1/world_size average is folded in as `grad_scale` (hvd.Average, neurst/training/hvd_utils.py:46-50).
"""
import math

import torch

from neurst_amd import kernels as K
from neurst_amd.optimizers.registries import register_optimizer


@register_optimizer("Adam")
class Adam(object):
    def __init__(self, store=None, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, **unused):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon
        self.iterations = 0
        self.store = None
        if store is not None:
            self.bind(store)

    def bind(self, store):
        self.store = store
        self.m = torch.zeros_like(store.master)
        self.v = torch.zeros_like(store.master)
        return self

    def current_lr(self):
        lr = self.learning_rate
        return float(lr(self.iterations)) if callable(lr) else float(lr)

    def step_size(self):
        """lr_t of the NEXT update: lr * sqrt(1 - beta_2^t) / (1 - beta_1^t)."""
        t = self.iterations + 1
        return self.current_lr() * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)

    def apply_gradients(self, grad_scale=1.0, lr_t_dev=None, loss_scale_state=None):
        """lr_t_dev: 1-element float32 device tensor holding step_size() -- the kernel reads it when it runs, so the launch
        can sit in a captured graph (the caller refreshes the scalar and calls advance() per replay)."""
        st = self.store
        K.adam_update(st.master, self.m, self.v, st.grad, st.shadow, lr_t_dev if lr_t_dev is not None else self.step_size(),
                      self.beta_1, self.beta_2, self.epsilon, grad_scale, loss_scale_state=loss_scale_state)
        st.refresh_transposed()   # the fused feed-forward reads transposed copies of its two kernels
        if lr_t_dev is None:
            self.advance()

    def advance(self):
        self.iterations += 1

    def state(self):
        """Resume state (checkpoints.py keeps it next to the weights): step count and both moment buffers."""
        return {"step": self.iterations, "m": self.m, "v": self.v}

    def load_state(self, st):
        if st["m"].numel() != self.m.numel():
            raise ValueError("optimizer state does not match this parameter layout")
        self.iterations = int(st["step"])
        self.m.copy_(st["m"].to(self.m.device))
        self.v.copy_(st["v"].to(self.v.device))

    def get_config(self):
        return {"beta_1": self.beta_1, "beta_2": self.beta_2, "epsilon": self.epsilon}
