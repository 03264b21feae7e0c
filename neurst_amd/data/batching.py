"""Length-bucketed batching of single examples, as the reference's tf.data pipeline for SpeechToText does it
(neurst/tasks/speech2text.py:236-384, neurst/data/dataset_utils.py:328-339, 435-465).

All functions are plain-Python restatements over iterators of dicts of numpy arrays:
  * `create_audio_bucket_boundaries` / `minimal_multiple` / `adjust_batch_size`: the boundary and batch-size arithmetic;
  * `clean_by_length`: the filter of `clean_dataset_by_length`;
  * `shuffle_buffer`: Dataset.shuffle's buffer algorithm (numpy generator instead of TF's);
  * `group_by_window_padded_batch`: tf.data.experimental.group_by_window with a padded_batch(drop_remainder=True) reducer:
    an example goes to the window of its bucket key, a window is emitted the moment it holds `window_size(key)`
    examples, windows still open at the end of the stream are dropped.
"""
import math

import numpy as np


def minimal_multiple(val, factor):
    """neurst/training/training_utils.py:48-51."""
    if val % factor == 0:
        return val
    return int((val // factor + 1) * factor)


def create_audio_bucket_boundaries(maxlen, minlen=128):
    """neurst/tasks/speech2text.py:38-57: bucket upper bounds growing by an increasing stride, last one = maxlen + 1."""
    if minlen is None:
        minlen = 128
    bounds = [minlen]
    base = minlen
    base_incr = int(2 ** ((math.log2(minlen) + 1) // 2))
    base_incr_mult = 1
    times = len(str(int(minlen)))
    while True:
        for _ in range(times):
            bounds.append(bounds[-1] + base)
            if bounds[-1] > maxlen:
                break
        base += base_incr * base_incr_mult
        base_incr_mult += 1
        if bounds[-1] > maxlen:
            break
    bounds[-1] = maxlen + 1
    return bounds


def adjust_batch_size(batch_size=None, batch_size_per_gpu=None, num_replicas_in_sync=1):
    """dataset_utils.py:435-452 without bucket boundaries: the global batch, a multiple of the replica count;
    `batch_size_per_gpu` takes precedence."""
    if batch_size is None and batch_size_per_gpu is None:
        raise ValueError("At least one of the `batch_size` and `batch_size_per_gpu` should be provided.")
    if batch_size_per_gpu is not None:
        batch_size = int(batch_size_per_gpu * num_replicas_in_sync)
    return int(batch_size // num_replicas_in_sync * num_replicas_in_sync)


_MIN_BUCKET_BOUNDARY = 8
_BUCKET_BOUNDARY_SCALE = 1.1


def create_batch_bucket_boundaries(max_length, min_boundary=_MIN_BUCKET_BOUNDARY, boundary_scale=_BUCKET_BOUNDARY_SCALE):
    """dataset_utils.py:125-147: token-length bucket bounds 8, 9, ... growing by 10 % (at least 1), last = max_length + 1."""
    bounds = []
    x = min_boundary
    while x < max_length:
        bounds.append(x)
        x = max(x + 1, int(x * boundary_scale))
    if bounds[-1] < max_length + 1:
        bounds = bounds + [max_length + 1]
    return bounds


def associated_bucket_boundaries(a, b):
    """dataset_utils.py:150-178: two boundary lists resampled to the same (shorter) length."""
    l1, l2 = len(a), len(b)
    if l1 == l2:
        return a, b
    if l1 > l2:
        s1, s2 = l1 * 1. / l2, 1
    else:
        s1, s2 = 1, l2 * 1. / l1
    n1, n2 = [], []
    i = 1
    while i < min(l1, l2) + 1:
        n1.append(a[int(math.ceil(i * s1)) - 1])
        n2.append(b[int(math.ceil(i * s2)) - 1])
        i += 1
    return n1, n2


def text_bucket_plan(max_src_len, max_trg_len, batch_size, batch_size_per_gpu, batch_by_tokens=True, num_replicas_in_sync=1):
    """The training bucket table of Seq2Seq.create_and_batch_tfds (tasks/seq2seq.py:247-264): associated source / target
    bounds and, per bucket, batch_size // max(bound) sentences (a multiple of the replica count) when batching by tokens,
    else a fixed sentence count.  Returns dict(src_bounds, trg_bounds, batch_sizes)."""
    if max_src_len is None:
        raise RuntimeError("Must provide `max_src_len` for training.")
    if max_trg_len is None:
        raise RuntimeError("Must provide `max_trg_len` for training.")
    sb, tb = associated_bucket_boundaries(create_batch_bucket_boundaries(max_src_len), create_batch_bucket_boundaries(max_trg_len))
    if batch_size is None and batch_size_per_gpu is None:
        raise ValueError("At least one of the `batch_size` and `batch_size_per_gpu` should be provided.")
    if batch_size_per_gpu is not None:
        batch_size = int(batch_size_per_gpu * num_replicas_in_sync)
    if batch_by_tokens:
        sizes = [int(batch_size // max(s, t) // num_replicas_in_sync * num_replicas_in_sync) for s, t in zip(sb, tb)]
    else:
        sizes = [int(batch_size // num_replicas_in_sync * num_replicas_in_sync)] * len(sb)
    return {"src_bounds": list(sb), "trg_bounds": list(tb), "batch_sizes": sizes}


def speech_bucket_plan(max_src_len, max_trg_len, min_src_bucket_boundary, batch_size, batch_size_per_gpu,
                       num_replicas_in_sync=1, disable_batch_efficiency=False, frame_transcript_ratio=None):
    """The training bucket table of speech2text.py:293-336.

    Returns dict(audio_bounds, batch_sizes, trans_bounds) -- `trans_bounds` is None (transcripts padded to the longest of
    the batch) or, with `experimental_frame_transcript_ratio`, one [this, next] pair of fixed transcript lengths per
    audio bucket."""
    if max_src_len is None:
        raise RuntimeError("`max_src_len` for SpeechToText task must be provided.")
    if max_trg_len is None:
        raise RuntimeError("`max_trg_len` for SpeechToText task must be provided.")
    max_trg_len = minimal_multiple(max_trg_len, 8)
    bounds = create_audio_bucket_boundaries(max_src_len, min_src_bucket_boundary)
    bounds[-1] = minimal_multiple(bounds[-1], 8)
    global_batch = adjust_batch_size(batch_size, batch_size_per_gpu, num_replicas_in_sync)
    per_gpu = global_batch // num_replicas_in_sync
    assert per_gpu > max_src_len, (f"batch size per gpu({per_gpu} must be greater than `max_src_len`={max_src_len}")
    if disable_batch_efficiency:
        sizes = [int(per_gpu // b * num_replicas_in_sync) for b in bounds]
    else:
        sizes = [int(minimal_multiple(per_gpu // b, 8) * num_replicas_in_sync) for b in bounds]
    trans = None
    if frame_transcript_ratio is not None:
        t = [int(b / (frame_transcript_ratio + i * (max_src_len / max_trg_len - frame_transcript_ratio) / len(bounds)))
             for i, b in enumerate(bounds)]
        t = [minimal_multiple(min(x, max_trg_len), 8) for x in t]
        trans = [[t[i], t[min(i + 1, len(t) - 1)]] for i in range(len(t))]
    return {"audio_bounds": bounds, "batch_sizes": sizes, "trans_bounds": trans}


def clean_by_length(examples, data_max_lengths):
    """dataset_utils.py:328-339: keep an example iff, for every listed key, size <= max (max None / -1 = unbounded) and,
    for bounded keys, size > 1."""
    for ex in examples:
        ok = True
        for k, length in data_max_lengths.items():
            size = int(np.size(ex[k]))
            if not (length == -1 or length is None or size <= length):
                ok = False
            if not (length == -1 or size > 1):
                ok = False
        if ok:
            yield ex


def shuffle_buffer(examples, buffer_size, rng):
    """Dataset.shuffle(buffer_size): fill a buffer, then emit a uniformly drawn slot and refill it from the stream."""
    if not buffer_size or buffer_size <= 1:
        yield from examples
        return
    buf = []
    for ex in examples:
        if len(buf) < buffer_size:
            buf.append(ex)
            continue
        i = int(rng.randint(0, buffer_size))
        out, buf[i] = buf[i], ex
        yield out
    while buf:
        i = int(rng.randint(0, len(buf)))
        buf[i], buf[-1] = buf[-1], buf[i]
        yield buf.pop()


def pad_batch(window, padded_lengths, padding_values):
    """Dataset.padded_batch of one window: 1-D fields are right padded to `padded_lengths[key]` (None = longest of the
    window), scalars are stacked."""
    out = {}
    for k in window[0]:
        vals = [np.asarray(ex[k]) for ex in window]
        if vals[0].ndim == 0:
            out[k] = np.stack(vals)
            continue
        target = padded_lengths.get(k, None)
        if target is None:
            target = max(v.shape[0] for v in vals)
        arr = np.full((len(vals), int(target)), padding_values[k], dtype=vals[0].dtype)
        for i, v in enumerate(vals):
            if v.shape[0] > target:
                raise ValueError(f"{k}: element of length {v.shape[0]} does not fit the padded length {target}")
            arr[i, :v.shape[0]] = v
        out[k] = arr
    return out


def group_by_window_padded_batch(examples, key_fn, window_size_fn, padded_lengths_fn, padding_values):
    windows = {}
    for ex in examples:
        key = key_fn(ex)
        if key is None:  # no bucket fits (the reference's tf.where(...)[0] would fail): cleaned out upstream
            continue
        w = windows.setdefault(key, [])
        w.append(ex)
        if len(w) == window_size_fn(key):
            yield pad_batch(w, padded_lengths_fn(key), padding_values)
            windows[key] = []
    # partial windows: padded_batch(drop_remainder=True) emits nothing for them
