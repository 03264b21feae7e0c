"""Beam search over a step function with a reorderable cache (neurst/layers/search/beam_search.py:24-551,
neurst/layers/layer_utils.py stack_beam_size / one_entry_bias).

The search itself is a handful of [batch * beam, vocab] tensor operations per step; they run as torch ops on the device
the logits live on (top-k, gather, log-softmax are plumbing here -- the model step inside `symbols_to_logits_fn` is the
HIP path).  Semantics kept from the reference:
  * every sample keeps exactly `beam_size` hypotheses; at step 0 only the first beam of a sample is expanded;
  * a finished hypothesis (last symbol EOS) can only continue with EOS at no cost, its length stops growing;
  * candidates are ranked by  accumulated log-prob * length_penalty(length),  ((5 + len) / 6) ** -alpha, or 1 / len for
    alpha < 0 / None;  UNK is masked unless `enable_unk`; EOS is masked while step < minimum_decode_length - 1;
  * the loop stops when every hypothesis has finished or after min(encoder_len + extra_decode_length,
    maximum_decode_length) steps (at least minimum_decode_length);
  * returns (hypotheses [batch * top_k, maximum_decode_length] padded with EOS, scores [batch * top_k]).
"""
import torch

from neurst_amd.utils.flags_core import Flag
from neurst_amd.utils.registry import setup_registry

FLOAT_MIN = -1.e9  # neurst/utils/compat.py FLOAT_MIN


class SequenceSearch(object):
    REGISTRY_NAME = "search_method"

    @staticmethod
    def class_or_method_args():
        return []


build_search_layer, register_search_layer = setup_registry(SequenceSearch.REGISTRY_NAME, base_class=SequenceSearch, backend="pt")


def stack_beam_size(x, beam_size):
    """layer_utils.stack_beam_size: repeats every batch entry `beam_size` times along dim 0 (b0,b0,..,b1,b1,..)."""
    if isinstance(x, dict):
        return {k: stack_beam_size(v, beam_size) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(stack_beam_size(v, beam_size) for v in x)
    if x is None or not torch.is_tensor(x):
        return x
    return x.repeat_interleave(beam_size, dim=0)


def length_penalty_term(lengths, alpha, dtype=torch.float32):
    """beam_search.py:24-41."""
    lengths = lengths.to(dtype)
    if alpha is None or alpha < 0.0:
        return 1.0 / lengths.clamp(min=1.0)
    return ((5.0 + lengths) / 6.0) ** (-alpha)


def sequence_beam_search(symbols_to_logits_fn, generation_initializer, top_k=1, beam_size=4, length_penalty=0.6,
                         extra_decode_length=50, maximum_decode_length=256, minimum_decode_length=0, enable_unk=False,
                         reorder_cache_fn=None, ensemble_weights=None):
    """generation_initializer: {"decoder_input": int tensor [batch], "decoder_internal_cache": cache ALREADY stacked to
    batch * beam rows by the caller (the model owns its layout), "encoder_inputs_maxlen": int | None, "eos_id", "unk_id"}.
    symbols_to_logits_fn(ids [batch*beam], cache, time) -> logits [batch*beam, vocab], or a list of them (an ENSEMBLE:
    the step distribution is sum_i ensemble_weights[i] * softmax(logits_i), beam_search.py:104-116).
    reorder_cache_fn(cache, beam_ids) -> cache re-ordered like tf.gather(cache, beam_ids)."""
    ids = generation_initializer["decoder_input"]
    cache = generation_initializer["decoder_internal_cache"]
    enc_len = generation_initializer.get("encoder_inputs_maxlen", None)
    eos_id = generation_initializer["eos_id"]
    unk_id = None if enable_unk else generation_initializer.get("unk_id", None)
    batch, dev = ids.shape[0], ids.device
    bb = batch * beam_size
    input_ids = stack_beam_size(ids, beam_size).long()
    finished = torch.zeros(bb, dtype=torch.bool, device=dev)
    log_probs = torch.zeros(bb, dtype=torch.float32, device=dev)
    lengths = torch.zeros(bb, dtype=torch.int64, device=dev)
    predicted = torch.zeros(bb, 0, dtype=torch.int64, device=dev)
    max_steps = maximum_decode_length if enc_len is None else min(int(enc_len) + extra_decode_length, maximum_decode_length)
    max_steps = max(max_steps, minimum_decode_length)
    beam_base = (torch.arange(batch, device=dev) * beam_size).repeat_interleave(beam_size)
    time = 0
    # the all-finished test costs a device sync: it is made every 4th step -- extra steps on finished beams only append
    # EOS at zero cost (log-prob, length and ranking unchanged), the returned hypotheses are the same
    while time < max_steps and not (time % 4 == 0 and time > 0 and bool(finished.all())):
        logits = symbols_to_logits_fn(input_ids, cache, time)
        if isinstance(logits, (list, tuple)) and len(logits) > 1:
            w = torch.as_tensor(ensemble_weights, dtype=torch.float32, device=dev)
            assert w.numel() == len(logits), "one ensemble weight per sub-model"
            step_lp = torch.log(sum(wi * torch.softmax(lg.float(), dim=-1) for wi, lg in zip(w, logits)))
        else:
            logits = logits[0] if isinstance(logits, (list, tuple)) else logits
            step_lp = torch.log_softmax(logits.float(), dim=-1)
        vocab = step_lp.shape[-1]
        # finished beams: only EOS, at no cost (beam_search.py:117-130)
        fin = finished.float()[:, None]
        fin_bias = torch.full((vocab,), FLOAT_MIN, dtype=torch.float32, device=dev)
        fin_bias[eos_id] = 0.0
        step_lp = step_lp * (1.0 - fin) + fin_bias[None, :] * fin
        if unk_id is not None:
            step_lp[:, unk_id] += FLOAT_MIN
        if time < minimum_decode_length - 1:
            step_lp[:, eos_id] += FLOAT_MIN
        # _sample_next_word (:144-215)
        total = step_lp + log_probs[:, None]
        next_len = lengths + 1 - finished.long()
        scores = (total * length_penalty_term(next_len, length_penalty)[:, None]).view(batch, beam_size * vocab)
        if time == 0:
            scores = scores[:, :vocab]
        _, sample = torch.topk(scores, k=beam_size, dim=-1)
        sample = sample.reshape(-1)
        word_ids = sample % vocab
        beam_ids = sample // vocab + beam_base
        lengths = next_len.index_select(0, beam_ids)
        log_probs = total.reshape(-1).index_select(0, beam_base * vocab + sample)
        predicted = torch.cat([predicted.index_select(0, beam_ids), word_ids[:, None]], dim=1)
        if reorder_cache_fn is not None:
            cache = reorder_cache_fn(cache, beam_ids)
        finished = word_ids == eos_id
        input_ids = word_ids
        time += 1
    # _extract_beam_results (:218-251)
    final = (log_probs * length_penalty_term(lengths, length_penalty)).view(batch, beam_size)
    top_scores, top_idx = torch.topk(final, k=top_k, dim=-1)
    rows = (top_idx + (torch.arange(batch, device=dev) * beam_size)[:, None]).reshape(-1)
    hyp = predicted.index_select(0, rows)
    if hyp.shape[1] < maximum_decode_length:
        hyp = torch.nn.functional.pad(hyp, (0, maximum_decode_length - hyp.shape[1]), value=eos_id)
    return hyp, top_scores.reshape(-1)


@register_search_layer(["beam_search", "BeamSearch"])
class BeamSearch(SequenceSearch):
    """SequenceSearch "beam_search" (beam_search.py:443-551): binds the search hyper-parameters, drives a model."""

    def __init__(self, args=None, beam_size=4, length_penalty=0.6, top_k=1, maximum_decode_length=None, minimum_decode_length=0,
                 extra_decode_length=50, enable_unk=False, use_graphs=False):
        a = dict(beam_size=beam_size, length_penalty=length_penalty, top_k=top_k, maximum_decode_length=maximum_decode_length,
                 minimum_decode_length=minimum_decode_length, extra_decode_length=extra_decode_length, enable_unk=enable_unk,
                 use_graphs=use_graphs)
        a.update({k: v for k, v in (args or {}).items() if v is not None and k in a})
        if (args or {}).get("padded_decode", None):
            pass  # the cache is preallocated to the maximum length anyway: nothing to switch
        self.use_graphs = bool(a["use_graphs"])
        self.beam_size, self.length_penalty, self.top_k = a["beam_size"], a["length_penalty"], a["top_k"]
        self.maximum_decode_length, self.minimum_decode_length = a["maximum_decode_length"], a["minimum_decode_length"] or 0
        self.extra_decode_length, self.enable_unk = a["extra_decode_length"], bool(a["enable_unk"])
        assert self.top_k <= self.beam_size

    @staticmethod
    def class_or_method_args():
        return [
            Flag("beam_size", dtype=Flag.TYPE.INTEGER, default=4, help="The beam width of beam search inference."),
            Flag("length_penalty", dtype=Flag.TYPE.FLOAT, default=0.6, help="The length penalty of beam search inference."),
            Flag("top_k", dtype=Flag.TYPE.INTEGER, default=1, help="The number of reserved predictions with top scores."),
            Flag("maximum_decode_length", dtype=Flag.TYPE.INTEGER, default=None, help="The maximum decoding length."),
            Flag("minimum_decode_length", dtype=Flag.TYPE.INTEGER, default=0, help="The minimum decoding length."),
            Flag("extra_decode_length", dtype=Flag.TYPE.INTEGER, default=50,
                 help="The extra decoding length versus the (encoded) source length."),
            Flag("padded_decode", dtype=Flag.TYPE.BOOLEAN, default=None, help="Accepted for compatibility (static cache always)."),
            Flag("enable_unk", dtype=Flag.TYPE.BOOLEAN, default=None, help="Whether the search may generate UNK."),
            Flag("use_graphs", dtype=Flag.TYPE.BOOLEAN, default=None, help="Replay the decoding step as a captured HIP graph."),
        ]

    def __call__(self, model, inputs, ensemble_weights=None):
        """model: one model, or a list of models decoded as an ENSEMBLE (sequence_generator.py builds
        EncoderDecoderEnsembleModel from several model_dirs; `ensemble_weights`: "average" / None, or one weight per model)."""
        max_len = self.maximum_decode_length or 256
        kw = dict(top_k=self.top_k, beam_size=self.beam_size, length_penalty=self.length_penalty,
                  extra_decode_length=self.extra_decode_length, maximum_decode_length=max_len,
                  minimum_decode_length=self.minimum_decode_length, enable_unk=self.enable_unk)
        if not isinstance(model, (list, tuple)):
            fn, init, reorder = model.get_symbols_to_logits_fn(inputs, beam_size=self.beam_size, decode_padded_length=max_len,
                                                               use_graphs=self.use_graphs)
            return sequence_beam_search(fn, init, reorder_cache_fn=reorder, **kw)
        parts = [m.get_symbols_to_logits_fn(inputs, beam_size=self.beam_size, decode_padded_length=max_len) for m in model]
        if ensemble_weights is None or ensemble_weights == "average":
            ensemble_weights = [1.0 / len(parts)] * len(parts)
        init = dict(parts[0][1], decoder_internal_cache=[p[1]["decoder_internal_cache"] for p in parts])

        def fn(ids, caches, time):
            return [p[0](ids, c, time) for p, c in zip(parts, caches)]

        def reorder(caches, beam_ids):
            return [p[2](c, beam_ids) for p, c in zip(parts, caches)]
        return sequence_beam_search(fn, init, reorder_cache_fn=reorder, ensemble_weights=ensemble_weights, **kw)
