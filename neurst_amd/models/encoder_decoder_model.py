"""EncoderDecoderModel (neurst/models/encoder_decoder_model.py:27-279): modalities -> encoder -> decoder ->
tied logits, training path, with an explicit backward pass."""
import torch

from neurst_amd.layers.common_layers import Dense, PositionEmbeddingWrapper
from neurst_amd.layers.decoders import Decoder, build_decoder
from neurst_amd.layers.encoders import Encoder, build_encoder
from neurst_amd.layers.modalities.text_modalities import WordEmbeddingSharedWeights
from neurst_amd.models.model import BaseModel, register_model
from neurst_amd.models.model_utils import input_length_to_padding
from neurst_amd.runtime import Runtime
from neurst_amd.utils.flags_core import Flag, ModuleFlag


def _timing_name(timing):
    if isinstance(timing, dict):
        timing = timing.get("timing")
    return timing


class _DecodeSession(object):
    """Static buffers + one captured HIP graph per decoding time index for one (batch * beam, memory length, dtype,
    maximum length) shape.  begin() copies a new batch's encoder memory into the static buffers and resets the cache
    lengths; step(ids, cache, t) replays graph t (capturing it the first time: an eager warm-up run, then the capture of
    the same step -- writing position t twice is idempotent)."""

    def __init__(self, decoder, eager_step, memory, padding, max_len):
        self.decoder, self.eager_step, self.max_len = decoder, eager_step, max_len
        self.memory, self.padding = memory.clone(), padding.clone()
        self.cache = decoder.create_decoding_internal_cache(self.memory, self.padding, is_inference=True, decode_padded_length=max_len)
        self.graphs, self.pool = {}, None

    def _set_len(self, n):
        for st in self.cache["decoding_states"].values():
            st["self_attention"]["len"] = n

    def begin(self, memory, padding):
        from neurst_amd.layers import layer_utils
        self.memory.copy_(memory)
        self.cache["memory_bias"].copy_(layer_utils.input_padding_to_bias(padding))
        self._set_len(0)
        return self.cache, self.step

    def step(self, ids, cache, time):
        assert cache is self.cache
        entry = self.graphs.get(time)
        if entry is None:
            static_ids = ids.clone()
            self._set_len(time)
            self.eager_step(static_ids, cache, time)          # warm-up: allocations, lazy state (projected memory at t = 0)
            self._set_len(time)
            if time == 0:  # the projection of the memory belongs INSIDE graph 0: every new batch must recompute it
                for st in cache["decoding_states"].values():
                    st["encdec_attention"].pop("kv", None)
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, pool=self.pool):
                static_logits = self.eager_step(static_ids, cache, time)
            if self.pool is None:
                self.pool = graph.pool()
            entry = self.graphs[time] = (graph, static_ids, static_logits)
        graph, static_ids, static_logits = entry
        static_ids.copy_(ids)
        graph.replay()
        self._set_len(time + 1)
        return static_logits


# where the grouped weight-gradient launch goes: "end" = behind the whole backward pass; "encoder" = behind the encoder stack
# (decoder + encoder products in one launch, in front of the front end's backward).  One launch per stack, and the groups that
# cannot fill the chip on the weight-gradient stream, were measured and lost (profiles/r04_history/c3..c5_ab_step.log: end 14.46,
# per stack 15.18, small groups on the side stream 14.75 ms per step -- a group that cannot fill the chip, or fills it 1.06 times,
# wastes more than the launch saves) and are gone.  "encoder" vs "end" on ONE GPU: 14.07 vs 13.91 and 13.79 vs 13.79 ms in two
# sessions (behind the encoder the group shares the chip with the weight-gradient stream's leftovers).
_WGRAD_GROUP_AT = None      # tests pin it; None: "encoder" whenever there is an exchange to overlap, else "end"


# The DECODER stack's group may leave earlier, on the weight-gradient stream right behind the decoder's backward, next to the
# encoder's backward on the compute stream -- when its units are SHORT (the reduction of a weight gradient runs over the rows:
# the speech models' decoder has L = 75 target positions against T/4 = 225 encoder frames per utterance).  Short units are
# picked up by the CUs the encoder's 225-workgroup launches leave idle and are gone before a launch that wants the whole chip
# comes along, and the launch at the end shrinks from 372 units (one long + one short round: 600 K steps) to the encoder's 240
# (one round of 450): speech_transformer_s 12.11 -> 12.02 ms, speech_transformer_m 21.30 -> 20.98 (three / two alternations on
# one box, profiles/r06_history/c22_group_dec.log, c23_*, c25_*: graph replay 12.23 -> 12.14 ms, eager launches unchanged).  With units as long as the encoder's the same move LOSES (text
# models, decoder rows = encoder rows: transformer_base 13.69 -> 13.90 ms, transformer_big 29.5 -> 29.6), as does the whole
# group on that stream (12.38 vs 12.22, c20_group_side.log).  Rule: decoder rows <= half of the encoder's.
_WGRAD_DECODER_SIDE = None   # tests pin True / False; None: the rule


def _decoder_group_on_side(ddec, dmemory):
    if _WGRAD_DECODER_SIDE is not None:
        return bool(_WGRAD_DECODER_SIDE)
    rows_dec = ddec.numel() // ddec.shape[-1]
    rows_enc = dmemory.numel() // dmemory.shape[-1]
    return 2 * rows_dec <= rows_enc


def _wgrad_group_at():
    import torch.distributed as dist
    exchanging = dist.is_available() and dist.is_initialized()
    if _WGRAD_GROUP_AT is not None:
        if _WGRAD_GROUP_AT == "end" and exchanging and dist.get_world_size() > 1:
            import warnings
            warnings.warn("grouped weight gradients at the END of the backward pass with more than one rank: every reducer report "
                          "waits for that launch, so no gradient exchange overlaps compute")
        return _WGRAD_GROUP_AT
    return "encoder" if exchanging else "end"


@register_model(["seq2seq", "sequence_to_sequence", "SequenceToSequence"])
class EncoderDecoderModel(BaseModel):
    def __init__(self, args, src_meta, trg_meta, src_modality, trg_modality, encoder, decoder, name=None, rt=None,
                 gen=None):
        super().__init__(args, name=name or "SequenceToSequence")
        self._src_meta, self._trg_meta = src_meta, trg_meta
        self._src_modality, self._trg_modality = src_modality, trg_modality
        self._encoder, self._decoder = encoder, decoder
        self.rt = rt
        # encoder_decoder_model.py:63-67: without weight tying the logits come from a separate Keras Dense
        # `softmax_linear` (kernel [d, V] glorot-uniform, bias [V]); created after the decoder like Keras builds it
        self._output_linear_layer = None
        if not args["modality.share_embedding_and_softmax_weights"]:
            self._output_linear_layer = Dense(rt, "softmax_linear", trg_modality.embedding_dim, trg_meta["vocab_size"],
                                              gen if gen is not None else torch.Generator().manual_seed(4242))
        self._logits_in = []

    @staticmethod
    def class_or_method_args():
        return [
            ModuleFlag(Encoder.REGISTRY_NAME, default=None, help="The encoder."),
            ModuleFlag(Decoder.REGISTRY_NAME, default=None, help="The decoder."),
            Flag("modality.share_source_target_embedding", dtype=Flag.TYPE.BOOLEAN, default=False,
                 help="Whether to share source and target embedding table."),
            Flag("modality.share_embedding_and_softmax_weights", dtype=Flag.TYPE.BOOLEAN, default=False,
                 help="Whether to share the target embedding table and softmax weights."),
            Flag("modality.dim", dtype=Flag.TYPE.INTEGER, default=None,
                 help="The default embedding dimension for both source and target side."),
            Flag("modality.source.dim", dtype=Flag.TYPE.INTEGER, default=None,
                 help="The source-side embedding dimension, or `modality.dim` if not provided."),
            Flag("modality.target.dim", dtype=Flag.TYPE.INTEGER, default=None,
                 help="The target-side embedding dimension, or `modality.dim` if not provided."),
            Flag("modality.timing", dtype=Flag.TYPE.STRING, default=None,
                 help="The arbitrary parameters for positional encoding of both source and target side."),
            Flag("modality.source.timing", dtype=Flag.TYPE.STRING, default=None,
                 help="The arbitrary parameters for source-side positional encoding."),
            Flag("modality.target.timing", dtype=Flag.TYPE.STRING, default=None,
                 help="The arbitrary parameters for target-side positional encoding."),
        ]

    # ------------------------------------------------------------------ construction helpers
    @classmethod
    def build_modality(cls, rt, gen, vocab_size, emb_dim, name, timing=None, share_embedding_and_softmax_weights=False):
        """encoder_decoder_model.py:118-145."""
        modality = WordEmbeddingSharedWeights(rt, name, emb_dim, vocab_size, gen,
                                              share_softmax_weights=share_embedding_and_softmax_weights)
        timing = _timing_name(timing)
        if timing:
            modality = PositionEmbeddingWrapper(rt, name + "_posenc_wrapper", modality, timing=timing)
        return modality

    @staticmethod
    def _runtime(kwargs):
        rt = kwargs.pop("runtime", None)
        if rt is None:
            rt = Runtime(device=kwargs.pop("device", "cuda:0"), dtype=kwargs.pop("dtype", "float32"),
                         seed=kwargs.pop("seed", 1234))
        gen = torch.Generator().manual_seed(kwargs.pop("init_seed", 42))
        return rt, gen

    def finalize(self):
        """Allocates the flat parameter buffers on the device (the reference's eager fake forward creates the
        Keras variables at this point, speech_transformer.py:172-176)."""
        self.rt.store.finalize(self.rt.device, self.rt.dtype)
        return self

    # ------------------------------------------------------------------ forward
    def output_logits_layer(self, features, is_training=True):
        """encoder_decoder_model.py:180-185."""
        if self._output_linear_layer is None:
            return self._trg_modality.forward(features, mode="linear", is_training=is_training)
        x2 = features.reshape(-1, features.shape[-1])
        if is_training:
            self._logits_in.append(x2)
        return self._output_linear_layer.forward(x2).view(*features.shape[:-1], self._output_linear_layer.out_dim)

    def _output_logits_backward(self, dlogits):
        """d(decoder output) from d(logits); the projection's own gradients go to the flat buffer."""
        if self._output_linear_layer is None:
            return self._trg_modality.backward(dlogits, mode="linear")
        lin = self._output_linear_layer
        x2 = self._logits_in.pop()
        dl = dlogits.reshape(-1, lin.out_dim)
        lin.backward_params(x2, dl)
        return lin.backward_input(dl).view(*dlogits.shape[:-1], lin.in_dim)

    def _src_padding(self, inputs, embedded_inputs):
        src_padding = inputs.get("src_padding", None)
        if src_padding is None:
            src_padding = input_length_to_padding(inputs["src_length"], embedded_inputs.shape[1])
        return src_padding

    def forward(self, inputs, is_training=True):
        """inputs: dict(src, src_length|src_padding, trg_input) of device tensors -> logits [B, L, V]
        (encoder_decoder_model.py:211-279)."""
        with self.rt.bound():
            embedded_inputs = self._src_modality.forward(inputs["src"], is_training=is_training)
            src_padding = self._src_padding(inputs, embedded_inputs)
            encoder_outputs = self._encoder.forward(embedded_inputs, src_padding, is_training=is_training)
            cache = self._decoder.create_decoding_internal_cache(encoder_outputs, src_padding, is_inference=False)
            dec_in = self._trg_modality.forward(inputs["trg_input"], is_training=is_training)
            decoder_output = self._decoder.forward(dec_in, cache, is_training=is_training,
                                                   decode_lagging=self.decode_lagging(is_training, None))
            return self.output_logits_layer(decoder_output, is_training=is_training)

    def decode_lagging(self, is_training, time):
        """The wait-k lagging of this call (None = full attention); WaitkTransformer overrides."""
        return None

    __call__ = forward

    # ------------------------------------------------------------------ inference
    def get_symbols_to_logits_fn(self, inputs, beam_size=1, decode_padded_length=256, use_graphs=False):
        """encoder_decoder_model.py:187-260 for inference: runs the encoder once, builds the decoder's incremental cache for
        batch * beam rows and returns (symbols_to_logits_fn, generation_initializer, reorder_cache_fn) for
        neurst_amd.layers.search.sequence_beam_search.

        use_graphs: a decoding step is ~100 small launches and host-launch bound; with use_graphs the step of every time
        index is captured ONCE into a HIP graph over static buffers (ids in, logits out, the K/V caches and the encoder
        memory in place) and replayed for every later batch of the same shape (see _DecodeSession)."""
        from neurst_amd.layers.search.beam_search import stack_beam_size
        embedded_inputs = self._src_modality.forward(inputs["src"], is_training=False)
        src_padding = self._src_padding(inputs, embedded_inputs)
        encoder_outputs = self._encoder.forward(embedded_inputs, src_padding, is_training=False)
        memory = stack_beam_size(encoder_outputs, beam_size).contiguous()
        padding = stack_beam_size(src_padding, beam_size).contiguous()

        def eager_step(ids, cache, time):
            dec_in = self._trg_modality.forward(ids, is_training=False, time=time)
            hidden = self._decoder.decode_step(dec_in, cache, decode_lagging=self.decode_lagging(False, time))
            return self.output_logits_layer(hidden, is_training=False)

        if use_graphs:
            key = (tuple(memory.shape), memory.dtype, decode_padded_length)
            sessions = self.__dict__.setdefault("_decode_sessions", {})
            if key not in sessions:
                sessions[key] = _DecodeSession(self._decoder, eager_step, memory, padding, decode_padded_length)
            cache, step_fn = sessions[key].begin(memory, padding)
        else:
            cache = self._decoder.create_decoding_internal_cache(memory, padding, is_inference=True,
                                                                 decode_padded_length=decode_padded_length)
            step_fn = eager_step
        batch = encoder_outputs.shape[0]
        first = inputs.get("trg_input", None)
        if first is None:
            first = torch.full((batch,), self._trg_meta["bos_id"], dtype=torch.int64, device=encoder_outputs.device)
        init = {"decoder_input": first.reshape(batch), "decoder_internal_cache": cache,
                "encoder_inputs_maxlen": int(encoder_outputs.shape[1]), "eos_id": self._trg_meta["eos_id"],
                "unk_id": self._trg_meta.get("unk_id", None)}
        return step_fn, init, self._decoder.reorder_cache

    def backward(self, dlogits, accumulate=False):
        """Back-propagates d(loss)/d(logits) through the whole model; parameter gradients land in the flat
        gradient buffer (rt.store.grad).  `accumulate`: add to existing gradients (update_cycle micro steps)."""
        stale = self.rt.drop_pending_wgrads()      # left behind by a backward pass that aborted (see Runtime.drop_pending_wgrads)
        if stale:
            import warnings
            warnings.warn(f"{stale} weight-gradient products / reducer reports of an aborted backward pass were dropped")
        try:
            self._backward(dlogits, accumulate)
        except BaseException:
            self.rt.drop_pending_wgrads()
            raise

    def _backward(self, dlogits, accumulate):
        with self.rt.bound():
            self.rt.store.begin_backward(accumulate)
            user_hook = self.grad_ready_hook or (lambda prefixes: None)

            def hook(prefixes):   # a report promises that everything writing these gradients is QUEUED: incl. deferred reduces
                def report():
                    self.rt.wgrad_boundary()
                    user_hook(prefixes)
                # the layer's deferred second stages (LayerNorm parameter gradients, split-K reduces) start now either way ...
                self.rt.flush_wgrads()
                self.rt.sublayer_boundary()
                # ... the REPORT waits while weight gradients of the layer sit in the group: it follows their launch
                self.rt.report_or_defer(report)
            # A report means "everything that writes these gradients has been QUEUED" (on the current stream or on the
            # weight-gradient stream); the reducer orders its side stream behind both.  The compute stream itself only
            # joins the weight-gradient stream once, at the end -- a join per component would stall the dgrad chain.
            shared = self._src_modality is self._trg_modality
            ddec = self._output_logits_backward(dlogits)
            self.rt.sublayer_boundary(force=True)   # the logits weight gradient (vocabulary x d) starts with the decoder's backward
            ddec_in, dmemory = self._decoder.backward(ddec, layer_done=hook)
            # softmax_linear (untied logits) is registered right after the decoder: one contiguous slice with it
            hook([self._decoder.name + "/"] + (["softmax_linear/"] if self._output_linear_layer is not None else []))
            self._trg_modality.backward(ddec_in, mode="embedding")
            if not shared:
                hook([self._modality_scope(self._trg_modality) + "/"])
            at = _wgrad_group_at()
            if _decoder_group_on_side(ddec, dmemory):
                self.rt.launch_wgrad_group(side=True)
            denc_in = self._encoder.backward(dmemory, layer_done=hook)
            hook([self._encoder.name + "/"])
            if at == "encoder":
                # decoder + encoder stacks in ONE full-chip launch here, in front of the front end's backward -- their gradients
                # (94 % of the parameters) are then reported to the data-parallel reducer ~2 ms before the step ends
                self.rt.launch_wgrad_group()
            self._src_modality.backward(denc_in, mode="embedding")
            hook([self._modality_scope(self._src_modality) + "/"])
            self.rt.launch_wgrad_group()   # whatever is still waiting ("end": everything in one launch)
            self.rt.join_wgrad_stream()

    grad_ready_hook = None  # callable(list of variable-name prefixes): the data-parallel reducer plugs in here

    @staticmethod
    def _modality_scope(modality):
        inner = getattr(modality, "embedding_layer", modality)
        return inner.name

    @property
    def store(self):
        return self.rt.store
