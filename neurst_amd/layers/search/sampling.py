"""Top-k / nucleus sampling search (neurst/layers/search/sampling.py:38-354): at every step the logits (UNK masked, EOS masked
before `minimum_decode_length`) are filtered and ONE next symbol per row is drawn from the renormalised distribution;
`sample_num` independent continuations per input ride in the batch like beams (they never interact, so no cache
re-ordering).  The filters follow the reference:

  top_k_logits (:67-75)   everything below the k-th largest logit of a row -> FLOAT_MIN (k = 0: no filter);
  top_p_logits (:78-92)   in descending order, the threshold is the first logit at which the cumulative probability
                          reaches p; every logit >= that threshold stays (the smallest prefix whose mass reaches p, plus
                          ties), the rest -> FLOAT_MIN.

Host-side control flow on device tensors, like the beam search; the draw uses a torch.Generator seeded per call
(the reference draws from TF's global generator -- parity is distributional).
"""
import torch

from neurst_amd.kernels import FLOAT_MIN
from neurst_amd.layers.search.beam_search import SequenceSearch, register_search_layer, stack_beam_size
from neurst_amd.utils.flags_core import Flag


def top_k_logits(logits, k):
    if k == 0:
        return logits
    kth = torch.topk(logits, k=k, dim=-1).values[:, -1:]
    return torch.where(logits < kth, torch.full_like(logits, FLOAT_MIN), logits)


def top_p_logits(logits, p):
    srt = torch.sort(logits, dim=-1, descending=True).values
    cum = torch.cumsum(torch.softmax(srt, dim=-1), dim=-1)
    masked = torch.where(cum < p, logits.min().expand_as(srt), srt)
    threshold = masked.max(dim=-1, keepdim=True).values
    return torch.where(logits < threshold, torch.full_like(logits, FLOAT_MIN), logits)


def sequence_sampling_search(symbols_to_logits_fn, generation_initializer, sample_next_word_fn, sample_num,
                             extra_decode_length=50, maximum_decode_length=256, minimum_decode_length=0, generator=None,
                             sync_every=4):
    """sampling.py:95-283 -> hypotheses [batch * sample_num, maximum_decode_length] (EOS padded)."""
    input_ids = stack_beam_size(generation_initializer["decoder_input"], sample_num)
    cache = generation_initializer["decoder_internal_cache"]
    eos_id, unk_id = generation_initializer["eos_id"], generation_initializer["unk_id"]
    steps = max(min(generation_initializer["encoder_inputs_maxlen"] + extra_decode_length, maximum_decode_length),
                minimum_decode_length)
    finished = torch.zeros_like(input_ids, dtype=torch.bool)
    out = []
    for time in range(steps):
        logits = symbols_to_logits_fn(input_ids, cache, time).float()
        if unk_id is not None:
            logits[:, unk_id] = FLOAT_MIN
        if time < minimum_decode_length - 1:
            logits[:, eos_id] = FLOAT_MIN
        probs = torch.softmax(sample_next_word_fn(logits), dim=-1)
        ids = torch.multinomial(probs, num_samples=1, generator=generator)[:, 0]
        out.append(ids)
        finished = finished | (ids == eos_id)
        input_ids = ids
        if (time + 1) % sync_every == 0 and bool(finished.all()):   # the reference tests the flags every step (one sync each)
            break
    hyp = torch.stack(out, dim=1)
    # symbols drawn after a row's first EOS are not part of its hypothesis: the reference keeps sampling for finished rows
    # too and leaves the cut to the post-processing (decode() stops at the first EOS); pad the tail with EOS
    if hyp.shape[1] < maximum_decode_length:
        hyp = torch.nn.functional.pad(hyp, (0, maximum_decode_length - hyp.shape[1]), value=eos_id)
    return hyp


@register_search_layer(["TopSampling", "top_sampling", "sampling"])
class Sampling(SequenceSearch):
    def __init__(self, args=None, sample_num=1, top_k=0, top_p=1.0, maximum_decode_length=None, minimum_decode_length=0,
                 extra_decode_length=50, seed=None):
        a = dict(sample_num=sample_num, top_k=top_k, top_p=top_p, maximum_decode_length=maximum_decode_length,
                 minimum_decode_length=minimum_decode_length, extra_decode_length=extra_decode_length, seed=seed)
        a.update({k: v for k, v in (args or {}).items() if v is not None and k in a})
        self.sample_num, self.top_k_filter, self.top_p = a["sample_num"], a["top_k"], a["top_p"]
        self.maximum_decode_length, self.minimum_decode_length = a["maximum_decode_length"], a["minimum_decode_length"] or 0
        self.extra_decode_length, self.seed = a["extra_decode_length"], a["seed"]
        if self.top_p < 1 and self.top_k_filter > 0:
            raise NotImplementedError("Not implemented search logic when top_k > 0 and top_p < 1.")
        self.top_k = self.sample_num   # rows per input in the returned hypotheses (what callers slice by)
        self._calls = 0

    @staticmethod
    def class_or_method_args():
        return [
            Flag("sample_num", dtype=Flag.TYPE.INTEGER, default=1, help="The number of copies for each input item."),
            Flag("top_k", dtype=Flag.TYPE.INTEGER, default=0, help="The number of token in each step for top_k sampling."),
            Flag("top_p", dtype=Flag.TYPE.FLOAT, default=1.0, help="The threshold for cumulated probability."),
            Flag("maximum_decode_length", dtype=Flag.TYPE.INTEGER, default=None, help="The maximum decoding length of sampling."),
            Flag("minimum_decode_length", dtype=Flag.TYPE.INTEGER, default=0, help="The minimum decoding length of sampling."),
            Flag("extra_decode_length", dtype=Flag.TYPE.INTEGER, default=50,
                 help="The extra decoding length versus the (encoded) source length."),
            Flag("padded_decode", dtype=Flag.TYPE.BOOLEAN, default=None, help="Accepted for compatibility (static cache always)."),
            Flag("seed", dtype=Flag.TYPE.INTEGER, default=None, help="Seed of the sampler (default: torch's global generator)."),
        ]

    def __call__(self, model, inputs):
        max_len = self.maximum_decode_length or 256
        if self.minimum_decode_length >= max_len:
            raise ValueError("`minimum_decode_length` must be less than maximum decode length.")
        fn, init, _ = model.get_symbols_to_logits_fn(inputs, beam_size=self.sample_num, decode_padded_length=max_len)
        gen = None
        if self.seed is not None:
            gen = torch.Generator(device=init["decoder_input"].device).manual_seed(int(self.seed) + self._calls)
            self._calls += 1
        filt = (lambda lg: top_p_logits(lg, self.top_p)) if self.top_p < 1 else (lambda lg: top_k_logits(lg, self.top_k_filter))
        hyp = sequence_sampling_search(fn, init, filt, self.sample_num, extra_decode_length=self.extra_decode_length,
                                       maximum_decode_length=max_len, minimum_decode_length=self.minimum_decode_length,
                                       generator=gen)
        return hyp, torch.zeros(hyp.shape[0], device=hyp.device)
