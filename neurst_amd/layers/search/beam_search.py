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

FLOAT_MIN = -1.e9  # neurst/utils/compat.py FLOAT_MIN


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
                         reorder_cache_fn=None):
    """generation_initializer: {"decoder_input": int tensor [batch], "decoder_internal_cache": cache ALREADY stacked to
    batch * beam rows by the caller (the model owns its layout), "encoder_inputs_maxlen": int | None, "eos_id", "unk_id"}.
    symbols_to_logits_fn(ids [batch*beam], cache, time) -> logits [batch*beam, vocab].
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
        vocab = logits.shape[-1]
        step_lp = torch.log_softmax(logits.float(), dim=-1)
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


class BeamSearch(object):
    """SequenceSearch "beam_search" (beam_search.py:443-551): binds the search hyper-parameters, drives a model."""

    def __init__(self, beam_size=4, length_penalty=0.6, top_k=1, maximum_decode_length=None, minimum_decode_length=0,
                 extra_decode_length=50, enable_unk=False, use_graphs=False):
        self.use_graphs = use_graphs
        self.beam_size, self.length_penalty, self.top_k = beam_size, length_penalty, top_k
        self.maximum_decode_length, self.minimum_decode_length = maximum_decode_length, minimum_decode_length
        self.extra_decode_length, self.enable_unk = extra_decode_length, enable_unk
        assert top_k <= beam_size

    def __call__(self, model, inputs):
        max_len = self.maximum_decode_length or 256
        fn, init, reorder = model.get_symbols_to_logits_fn(inputs, beam_size=self.beam_size, decode_padded_length=max_len,
                                                           use_graphs=self.use_graphs)
        return sequence_beam_search(fn, init, top_k=self.top_k, beam_size=self.beam_size, length_penalty=self.length_penalty,
                                    extra_decode_length=self.extra_decode_length, maximum_decode_length=max_len,
                                    minimum_decode_length=self.minimum_decode_length, enable_unk=self.enable_unk,
                                    reorder_cache_fn=reorder)
