"""Checkpoint managers over TensorFlow-format tensor bundles (neurst/utils/checkpoints.py).

  * NameBasedCheckpointManager (:148-183): `ckpt-<step>.index/.data-00000-of-00001` holding every model variable under the
    reference's names (`<ModelName>/<variable path>`, object-based keys, see tensor_bundle.py), the text file `checkpoint`
    in TensorFlow's CheckpointState layout (:127-141), at most `max_to_keep` checkpoints on disk;
  * restore_checkpoint_if_possible (:340-361): latest checkpoint of a directory (or an explicit prefix), variables
    matched by name AFTER replacing the checkpoint's top scope by the model's -- a `SpeechTransformer/...` checkpoint
    loads into a model called anything; optional regex filter; returns the path, or None when nothing is there;
  * optimizer slots and the step are saved next to the weights under `_optimizer/...` keys (the reference keeps them in
    its Keras optimizer; they are ignored by a weights-only restore).
"""
import glob
import logging
import os
import re
import time

import numpy as np
import torch

from neurst_amd.utils import tensor_bundle as tb


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: the `model_checkpoint_path` of directory/checkpoint, if its index file exists."""
    meta = os.path.join(directory, "checkpoint")
    if not os.path.isfile(meta):
        return None
    for line in open(meta):
        m = re.match(r'\s*model_checkpoint_path:\s*"(.*)"', line)
        if m:
            path = m.group(1)
            if not os.path.isabs(path):
                path = os.path.join(directory, path)
            return path if os.path.exists(path + ".index") else None
    return None


def list_variables(prefix):
    """[(variable name, shape)] of a checkpoint (tf.train.list_variables with the object-based key decoration removed)."""
    out = []
    for key, val in tb.read_bundle(prefix).items():
        if key == tb.OBJECT_GRAPH_KEY or key.startswith("_optimizer"):
            continue
        out.append((tb.variable_name(key), list(np.shape(val))))
    return out


def checkpoint_scope_name(prefix):
    """checkpoints.py:315-337: the top scope shared by the checkpoint's variables."""
    scopes = set(n.split("/")[0] for n, _ in list_variables(prefix) if "/" in n)
    if len(scopes) > 1:
        logging.warning("more than one scope names(%s) extracted from %s", scopes, prefix)
    return scopes.pop() if scopes else None


def remove_checkpoint_by_prefix(dirname, prefix):
    for f in glob.glob(os.path.join(dirname, prefix) + ".data-?????-of-?????") + [os.path.join(dirname, prefix) + ".index"]:
        try:
            os.remove(f)
        except OSError:
            pass


def _model_scope(model):
    return getattr(model, "name", None) or model.__class__.__name__


def restore_checkpoint_if_possible(model, model_dir, var_name_pattern=None, optimizer=None):
    """Loads the latest checkpoint under `model_dir` (or the bundle prefix `model_dir`) into model.store.
    Returns the checkpoint path, or None if there is no readable checkpoint or no variable matched."""
    if not model_dir:
        return None
    path = latest_checkpoint(model_dir) if os.path.isdir(model_dir) else None
    if path is None:
        path = model_dir
        if not os.path.exists(path + ".index"):
            return None
    try:
        bundle = tb.read_bundle(path)
    except (IOError, ValueError) as e:
        logging.warning("fail to read checkpoint %s: %s", path, e)
        return None
    by_name = {tb.variable_name(k): v for k, v in bundle.items() if k != tb.OBJECT_GRAPH_KEY and not k.startswith("_optimizer")}
    scopes = set(n.split("/")[0] for n in by_name if "/" in n)
    ckpt_scope = scopes.pop() if len(scopes) == 1 else None
    restored, unrestored, sd = [], [], {}
    for name, p in model.store.params.items():
        if var_name_pattern is not None and re.search(var_name_pattern, name) is None:
            continue
        cand = [f"{ckpt_scope}/{name}"] if ckpt_scope else []
        cand += [f"{_model_scope(model)}/{name}", name]
        hit = next((c for c in cand if c in by_name and tuple(np.shape(by_name[c])) == tuple(p.shape)), None)
        if hit is None:
            unrestored.append(name)
        else:
            sd[name] = torch.from_numpy(np.asarray(by_name[hit], dtype=np.float32))
            restored.append(name)
    if not restored:
        logging.info("No variables matched with checkpoint: %s", path)
        return None
    model.store.load_state_dict(sd, strict=False)
    if not unrestored:
        logging.info("All variables matched with checkpoint: %s", path)
    else:
        for n in unrestored:
            logging.info("Unrestored %s", n)
    if optimizer is not None and "_optimizer/step" in bundle:
        optimizer.load_state({"step": int(bundle["_optimizer/step"]), "m": torch.from_numpy(bundle["_optimizer/m"]),
                              "v": torch.from_numpy(bundle["_optimizer/v"])})
    return path


class NameBasedCheckpointManager(object):
    def __init__(self, model, directory, max_to_keep=8, checkpoint_name="ckpt", optimizer=None):
        self._model, self._directory = model, directory
        self._max_to_keep, self._checkpoint_name, self._optimizer = max_to_keep, checkpoint_name, optimizer
        self._all_model_checkpoints = []  # (prefix, timestamp)
        os.makedirs(directory, exist_ok=True)

    @property
    def directory(self):
        return self._directory

    def _update_checkpoint_meta(self):
        while len(self._all_model_checkpoints) > self._max_to_keep:
            prefix, _ = self._all_model_checkpoints.pop(0)
            remove_checkpoint_by_prefix(self._directory, prefix)
        text = 'model_checkpoint_path: "{}"\n'.format(self._all_model_checkpoints[-1][0])
        for path, _ in self._all_model_checkpoints:
            text += 'all_model_checkpoint_paths: "{}"\n'.format(path)
        for _, ts in self._all_model_checkpoints:
            text += "all_model_checkpoint_timestamps: {}\n".format(str(ts))
        tmp = os.path.join(self._directory, "checkpoint.incomplete")
        with open(tmp, "w") as fp:
            fp.write(text)
        os.replace(tmp, os.path.join(self._directory, "checkpoint"))

    def save(self, checkpoint_number):
        return self._write("{}-{}".format(self._checkpoint_name, checkpoint_number), time.time())

    def _write(self, prefix, tag, state_dict=None):
        """Writes the bundle `prefix` (weights from `state_dict` or the live model) and appends (prefix, tag) to the kept list."""
        scope = _model_scope(self._model)
        names = [f"{scope}/{n}" for n in self._model.store.params]
        tensors = {tb.checkpoint_key(f"{scope}/{n}"): np.asarray(v, dtype=np.float32)
                   for n, v in (state_dict or {k: t.numpy() for k, t in self._model.store.state_dict().items()}).items()}
        tensors[tb.OBJECT_GRAPH_KEY] = [tb.object_graph_proto(names)]
        if self._optimizer is not None:
            st = self._optimizer.state()
            tensors["_optimizer/step"] = np.asarray(st["step"], dtype=np.int64)
            tensors["_optimizer/m"] = st["m"].detach().cpu().numpy().astype(np.float32)
            tensors["_optimizer/v"] = st["v"].detach().cpu().numpy().astype(np.float32)
        path = os.path.join(self._directory, prefix)
        tb.write_bundle(path, tensors)
        self._all_model_checkpoints.append((prefix, tag))
        self._after_append()
        self._update_checkpoint_meta()
        return path

    def _after_append(self):
        pass

    def restore(self, restore_path=None):
        return restore_checkpoint_if_possible(self._model, restore_path or self._directory, optimizer=self._optimizer)


class KeepBestCheckpointSaver(NameBasedCheckpointManager):
    """checkpoints.py:186-237: keeps the `max_to_keep` checkpoints with the best validation metric.  A checkpoint is written
    whenever fewer than max_to_keep are kept or the new value is at least as good as the WORST kept one; the kept list is
    ordered worst -> best (so rotation drops the worst and `model_checkpoint_path` names the best) and the value is part of
    the name: ckpt-<step>-<value %.2f>."""

    def __init__(self, model, directory, metric, max_to_keep=8, checkpoint_name="ckpt"):
        super().__init__(model, directory, max_to_keep=max_to_keep, checkpoint_name=checkpoint_name)
        self._metric = metric

    def _accepts(self, value):
        kept = self._all_model_checkpoints
        return len(kept) < self._max_to_keep or self._metric.greater_or_eq(value, kept[0][1])

    def _after_append(self):
        import functools
        ge = self._metric.greater_or_eq

        def worse_first(x, y):
            if x[1] == y[1]:
                return 0
            return -1 if ge(y[1], x[1]) else 1
        self._all_model_checkpoints.sort(key=functools.cmp_to_key(worse_first))

    def _state(self):
        return None

    def save(self, checkpoint_number, metric_value):
        """-> path if a checkpoint was written, else None."""
        value = float(self._metric.get_value(metric_value))
        if not self._accepts(value):
            return None
        return self._write("{}-{}-{}".format(self._checkpoint_name, checkpoint_number, "%.2f" % value), value, self._state())


class AverageCheckpointSaver(KeepBestCheckpointSaver):
    """checkpoints.py:239-312: at every validation the current weights join a window of the latest `max_to_keep` evaluated
    weight sets; under the same keep-best rule the AVERAGE of the window is written."""

    def __init__(self, model, directory, metric, max_to_keep=8, checkpoint_name="ckpt"):
        super().__init__(model, directory, metric, max_to_keep=max_to_keep, checkpoint_name=checkpoint_name)
        self._window = []

    def save(self, checkpoint_number, metric_value):
        self._window.append({k: v.numpy().astype(np.float64) for k, v in self._model.store.state_dict().items()})
        if len(self._window) > self._max_to_keep:
            self._window.pop(0)
        return super().save(checkpoint_number, metric_value)

    def _state(self):
        return {k: (sum(w[k] for w in self._window) / len(self._window)).astype(np.float32) for k in self._window[0]}
