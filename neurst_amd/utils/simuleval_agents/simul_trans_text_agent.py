"""SimulTransTextAgent (neurst/utils/simuleval_agents/simul_trans_text_agent.py:45-245): the wait-k read / write policy
over a WaitkTransformer, streaming on the HIP path.

  policy   (:190-214)  READ while subword units of the current word are pending, or while fewer than `wait_k` more source
                       WORDS than target words have been seen (wait-k counts segments, not subwords); WRITE otherwise.
  predict  (:216-245)  encodes the source units that arrived since the last prediction (WaitkTransformer.
                       incremental_encode: monotonic encoder over cached keys / values, appended to the decoder memory),
                       decodes one target position against everything read (incremental_decode) and returns the argmax
                       of the models' averaged probabilities (log-mean-exp over an ensemble).
  segment_to_units / units_to_segment (:87-170)  word <-> subword-unit conversion through the task's text pipelines;
                       target words are flushed when the next unit starts with the SentencePiece word marker.

SimulEval itself is not installed in this image.  When it is importable the class derives from its TextAgent and uses its
actions / states; otherwise the stand-ins at the bottom of this file provide the few attributes the agent touches and
`run_agent_on_sentence` plays the client loop for one sentence, so the agent (and its latency) can be exercised locally.
"""
import logging

import torch

from neurst_amd.utils.simuleval_agents import register_agent

BOW_PREFIX = "▁"

try:  # pragma: no cover - SimulEval is not part of this image
    from simuleval import DEFAULT_EOS, READ_ACTION, WRITE_ACTION
    from simuleval.agents import TextAgent
    from simuleval.states import ListEntry, QueueEntry, TextStates
    HAVE_SIMULEVAL = True
except ImportError:
    HAVE_SIMULEVAL = False
    DEFAULT_EOS, READ_ACTION, WRITE_ACTION = "</s>", "read", "write"

    class ListEntry(object):
        def __init__(self):
            self.value = []

        def append(self, x):
            self.value.append(x)

        def __len__(self):
            return len(self.value)

        def __getitem__(self, i):
            return self.value[i]

        def __iter__(self):
            return iter(list(self.value))

    class QueueEntry(ListEntry):
        def push(self, x):
            self.value.append(x)

        def pop(self):
            return self.value.pop(0) if self.value else None

        def empty(self):
            return len(self.value) == 0

    class _Bag(object):
        pass

    class TextStates(object):
        """Stand-in for simuleval.states.TextStates: word-level source client, unit queues, read / write status."""

        def __init__(self, args, client, sentence_id, agent):
            self.args, self.client, self.sentence_id, self.agent = args, client, sentence_id, agent
            self.units, self.segments, self.unit_queue = _Bag(), _Bag(), _Bag()
            self.status = {"read": True, "write": True}
            self.delays = []           # source words read when each target word was emitted (for the latency)
            self.hypothesis = []

        source = property(lambda self: self.units.source)
        target = property(lambda self: self.units.target)

        def finish_read(self):
            return not self.status["read"]

        def finish_hypo(self):
            return not self.status["write"]

        def update_source(self):
            if self.unit_queue.source.empty():
                word = self.client.next_word()
                if word is None:
                    self.status["read"] = False
                    return
                self.segments.source.append(word)
                for u in self.agent.segment_to_units(word, self):
                    self.unit_queue.source.push(u)
            if not self.unit_queue.source.empty():
                self.units.source.append(self.unit_queue.source.pop())

        def update_target(self, unit):
            self.units.target.append(unit)
            self.unit_queue.target.push(unit)
            segment = self.agent.units_to_segment(self.unit_queue.target, self)
            if segment is None:
                return
            for s in (segment if isinstance(segment, list) else [segment]):
                if s == DEFAULT_EOS:
                    self.status["write"] = False
                else:
                    self.segments.target.append(s)
                    self.hypothesis.append(s)
                    self.delays.append(len(self.segments.source))

    class TextAgent(object):
        def __init__(self, args):
            self.args = args


class _WordClient(object):
    def __init__(self, words):
        self._it = iter(words)

    def next_word(self):
        return next(self._it, None)


def build_task_and_model(model_dir, wait_k, **model_kwargs):
    """simul_trans_text_agent.py:33-42: task + (ensemble of) models from model_dir(s), wait_k overriding the stored one."""
    from neurst_amd.tasks import build_task
    from neurst_amd.utils.checkpoints import restore_checkpoint_if_possible
    from neurst_amd.utils.configurable import ModelConfigs
    dirs = [d for d in (model_dir.split(",") if isinstance(model_dir, str) else list(model_dir)) if d]
    cfgs = ModelConfigs.load(dirs[0])
    cfgs["task.class"] = "WaitkTranslation"
    cfgs.setdefault("task.params", {})["wait_k"] = wait_k
    task = build_task(cfgs)
    models = []
    for md in dirs:
        models.append(task.build_model(ModelConfigs.load(md), **model_kwargs))
        restore_checkpoint_if_possible(models[-1], md)
    return task, models


@register_agent
class SimulTransTextAgent(TextAgent):
    def __init__(self, args, task=None, models=None):
        super().__init__(args)
        self.wait_k = args.wait_k
        if task is None:
            task, models = build_task_and_model(args.model_dir, self.wait_k, device=getattr(args, "device", None) or "cuda:0",
                                                dtype=getattr(args, "dtype", None) or "float32")
        self.task, self.models = task, models
        self.force_segment = getattr(args, "force_segment", False)
        self.max_len = getattr(args, "max_len", 200)
        self.src_pipeline, self.trg_pipeline = task._src_data_pipeline, task._trg_data_pipeline
        self.words = ['lbs.', 'Dr.', 'Prof.', 'Mr.', 'Mrs.', 'Ms.']

    @staticmethod
    def add_args(parser):
        parser.add_argument('--model-dir', type=str, required=True, dest="model_dir", help='Path to the MT model(s).')
        parser.add_argument("-k", "--wait-k", type=int, dest="wait_k", default=3)
        parser.add_argument("--force-segment", default=False, action="store_true", dest="force_segment")
        parser.add_argument("--max-len", type=int, default=200, dest="max_len", help="Max length of translation")
        parser.add_argument("--device", type=str, default="cuda:0", help="The GPU this agent's model lives on.")
        parser.add_argument("--dtype", type=str, default="float32", help="Compute dtype: float32 or bfloat16.")

    # ------------------------------------------------------------------ states
    def build_states(self, args, client, sentence_id):
        states = TextStates(args, client, sentence_id, self)
        self.initialize_states(states)
        return states

    def initialize_states(self, states):
        states.units.source, states.units.target = ListEntry(), ListEntry()
        states.segments.source, states.segments.target = ListEntry(), ListEntry()
        states.unit_queue.source, states.unit_queue.target = QueueEntry(), QueueEntry()
        states.encoder_cache = [{} for _ in self.models]
        states.decoder_cache = [{} for _ in self.models]
        states.segment = False
        states.encoding_time, states.decoding_time = 0, 0

    # ------------------------------------------------------------------ words <-> units
    def _sentence_end(self, segment):
        for q in ('"', ''):
            n = len(q)
            if segment.endswith('.' + q) or segment.endswith('?' + q) or segment.endswith('!' + q):
                if segment.endswith('...' + q):
                    return True
                core = segment[:-n] if n else segment
                if len(segment) > 1 + n and not segment[-2 - n].isupper() and core not in self.words:
                    return True
        return False

    def segment_to_units(self, segment, states):
        """One source word -> its subword ids (without the EOS the pipeline appends); with force_segment a word that
        ends a sentence also closes the current sub-sentence with EOS (:87-108)."""
        if self.force_segment and self._sentence_end(segment):
            states.segment = True
        units = list(self.src_pipeline.encode(segment))[:-1]
        if self.force_segment and states.segment:
            units.append(self.src_pipeline.meta["eos_id"])
        return units

    def units_to_segment(self, units, states):
        """Target unit queue -> finished word(s), DEFAULT_EOS, or None while the word may still grow (:110-170)."""
        eos = self.trg_pipeline.meta["eos_id"]
        if eos == units[0] or len(states.segments.target) > self.max_len:
            units.pop()
            if self.force_segment and states.status["read"]:
                states.segment = False
                self.initialize_states(states)
                return None
            return DEFAULT_EOS
        tokens = self.trg_pipeline.tokens
        if self.trg_pipeline.meta.get("language", None) == "ja":   # character by character
            token = tokens[units.pop()]
            if token == BOW_PREFIX:
                return None
            return token[1:] if token[0] == BOW_PREFIX else token
        segment = []
        for index in units:
            token = tokens[index]
            if token.startswith(BOW_PREFIX):
                if len(segment) == 0:
                    segment.append(token.replace(BOW_PREFIX, ""))
                else:   # the next word starts: flush the finished one
                    for _ in range(len(segment)):
                        units.pop()
                    out = ["".join(segment)]
                    if eos == units[0]:
                        out.append(DEFAULT_EOS)
                    return out
            else:
                segment.append(token.replace(BOW_PREFIX, ""))
        if (len(units) > 0 and eos == units[-1]) or len(states.units.target) > self.max_len:
            text = "".join(tokens[u] for u in list(units)[:-1]).replace(BOW_PREFIX, "")
            if self.force_segment and states.status["read"]:
                states.segment = False
                self.initialize_states(states)
                return [text]
            return [text, DEFAULT_EOS]
        return None

    # ------------------------------------------------------------------ policy / prediction
    def policy(self, states):
        if self.force_segment and not states.status["read"]:
            return WRITE_ACTION
        if self.force_segment and states.segment:
            return READ_ACTION if not states.unit_queue.source.empty() else WRITE_ACTION
        eos = self.src_pipeline.meta["eos_id"]
        if not states.status["read"] and (len(states.units.source) == 0 or states.units.source[-1] != eos):
            states.units.source.append(eos)   # finished reading: close the source
        if not states.unit_queue.source.empty() and states.status["read"]:
            return READ_ACTION
        if len(states.segments.source) - len(states.segments.target) < self.wait_k and not states.finish_read():
            return READ_ACTION   # wait-k at the WORD level
        return WRITE_ACTION

    def predict(self, states):
        eos = self.trg_pipeline.meta["eos_id"]
        if self.force_segment and not states.status["read"]:
            return eos
        if len(states.units.target) > self.max_len:
            return eos
        src_indices = list(states.source.value[states.encoding_time:])
        trg_input = self.trg_pipeline.meta["bos_id"] if self.task._target_begin_of_sentence == "bos" else eos
        if len(states.target.value) > 0:
            trg_input = states.target.value[-1]
        log_probs = []
        for i, model in enumerate(self.models):
            if len(src_indices) > 0:
                states.encoder_cache[i], states.decoder_cache[i] = model.incremental_encode(
                    {"src": [src_indices], "src_length": [len(src_indices)]}, states.encoder_cache[i],
                    states.decoder_cache[i], time=states.encoding_time,
                    max_source_length=getattr(self.args, "max_source_units", 1024), decode_padded_length=self.max_len + 8)
            logits, states.decoder_cache[i] = model.incremental_decode([trg_input], states.decoder_cache[i],
                                                                       time=states.decoding_time)
            log_probs.append(torch.log_softmax(logits[0].float(), dim=-1))
        states.encoding_time = len(states.source.value)
        states.decoding_time += 1
        total = torch.logsumexp(torch.stack(log_probs, 0), dim=0)   # - log(n_models): constant under argmax
        return int(torch.argmax(total))


def average_lagging(delays, source_length, target_length=None):
    """Average Lagging (Ma et al. 2019, STACL eq. 11-13) of one sentence from the number of source words read when each
    target word was written: AL = 1/tau * sum_{t<=tau} (g(t) - (t-1)/gamma), gamma = |y|/|x|, tau = first t with
    g(t) = |x|."""
    if not delays or source_length == 0:
        return 0.0
    gamma = (target_length or len(delays)) / float(source_length)
    tau = next((t for t, g in enumerate(delays, 1) if g >= source_length), len(delays))
    return sum(delays[t - 1] - (t - 1) / gamma for t in range(1, tau + 1)) / tau


def run_agent_on_sentence(agent, source_words, sentence_id=0, max_actions=10000):
    """Local stand-in for SimulEval's client loop on one sentence (text input, word granularity): alternates the
    agent's policy with source reads / target writes until the agent emits DEFAULT_EOS.  Returns a dict with the
    hypothesis words, the delays and the Average Lagging."""
    if HAVE_SIMULEVAL:  # pragma: no cover
        logging.warning("simuleval is installed: use its own CLI for evaluation; this loop is the local stand-in")
    states = agent.build_states(agent.args, _WordClient(source_words), sentence_id)
    actions = []
    for _ in range(max_actions):
        if states.finish_hypo():
            break
        action = agent.policy(states)
        actions.append(action)
        if action == READ_ACTION:
            states.update_source()
        else:
            states.update_target(agent.predict(states))
    return {"hypothesis": list(states.hypothesis), "delays": list(states.delays), "actions": actions,
            "average_lagging": average_lagging(states.delays, len(source_words)),
            "source_units": list(states.units.source.value), "target_units": list(states.units.target.value)}
