"""HIP-graph replay of the training step (neurst_amd/training/train_step.py, graph mode) against the eager step: same
losses and weights for the same seeds, new dropout masks at every replay (the step counter is read from device memory by
the kernels), Adam's step size refreshed per replay, and the capture cut into segments where a data-parallel reducer
issues its buckets."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(dtype, dropout, use_graph, reducer_factory=None, lr=1e-2, seed=11, **step_kwargs):
    from neurst_amd.criterions import build_criterion
    from neurst_amd.models import build_model
    from neurst_amd.optimizers.adam import Adam
    from neurst_amd.training.train_step import TrainStep
    from neurst_amd.utils.hparams_sets import get_hyper_parameters
    p = dict(get_hyper_parameters("speech_transformer_toy")["model.params"])
    p.update({"modality.dim": 64, "modality.source.channels": 32, "encoder.num_layers": 2, "decoder.num_layers": 2,
              "encoder.hidden_size": 64, "decoder.hidden_size": 64, "encoder.num_attention_heads": 2,
              "decoder.num_attention_heads": 2, "encoder.filter_size": 128, "decoder.filter_size": 128})
    for k in list(p):
        if k.endswith("dropout_rate"):
            p[k] = dropout
    V = 50
    model = build_model({"model.class": "SpeechTransformer", "model.params": p},
                        {"audio_feature_dim": 16, "audio_feature_channels": 1},
                        {"vocab_size": V, "eos_id": V - 1, "bos_id": V - 2, "unk_id": V - 3}, device=DEV, dtype=dtype,
                        init_seed=3, seed=seed)
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    opt = Adam(model.store, learning_rate=(lambda it: lr * (1.0 + 0.1 * it)), beta_1=0.9, beta_2=0.98, epsilon=1e-9)
    red = reducer_factory(model.store) if reducer_factory else None
    return model, TrainStep(model, crit, opt, red, use_graph=use_graph, **step_kwargs), opt


def _solid(v, rel=1e-8):
    """Adam divides by sqrt(v): where the true gradient is zero (key biases: softmax is shift invariant) the update is
    lr * sign(rounding noise), and the float atomics of the embedding gradient make that noise run-dependent.  Those
    elements are excluded from weight comparisons."""
    return (v > rel * v.max()).float()


def _batch(seed, B=3, T=70, F=16, L=9, V=50):
    g = torch.Generator().manual_seed(seed)
    src = torch.randn(B, T, F, 1, generator=g)
    trg = torch.randint(0, V - 3, (B, L), generator=g)
    trg[:, -1] = V - 1
    b = {"src": src, "src_length": torch.tensor([T, T - 9, T - 20]), "trg": trg, "trg_length": torch.tensor([L, L, L]),
         "trg_input": torch.cat([torch.full((B, 1), V - 2), trg[:, :-1]], 1)}
    return {k: v.to(DEV) for k, v in b.items()}


@pytest.mark.parametrize("dtype,dropout", [("float32", 0.0), ("float32", 0.1), ("bfloat16", 0.1)])
def test_graph_replay_equals_eager_steps(dtype, dropout):
    batches = [_batch(100 + i) for i in range(6)]
    res = {}
    for use_graph in (False, True):
        model, step, opt = _setup(dtype, dropout, use_graph)
        losses = [float(step(b)) for b in batches]
        torch.cuda.synchronize()
        res[use_graph] = (losses, model.store.master.clone(), opt.iterations, model.rt.step, step.replays, opt.v.clone())
    (le, we, ie, se, _, ve), (lg, wg, ig, sg, replays, _) = res[False], res[True]
    assert replays == len(batches) - 1 and ie == ig == len(batches) and se == sg == len(batches)
    tol = 1e-5 if dtype == "float32" else 2e-3     # embedding gradients use float atomics: last-bit differences
    for a, b in zip(le, lg):
        assert abs(a - b) <= tol * max(1.0, abs(a)), (le, lg)
    if dtype == "float32":
        assert float(((we - wg).abs() * _solid(ve)).max()) <= 2e-5
    else:
        # bf16: a last-bit difference of an fp32 master weight (atomics order) can round its bf16 shadow the other way, after
        # which the two runs are two bf16-noise realisations of the same trajectory -- compare where the gradient is solid,
        # with a bound of a couple of Adam steps of pure noise (observed once in ~10 suite runs above the old 1e-3 / 1e-8)
        # (the fraction of weights that moved apart is the criterion; a single weight whose gradient sits at the mask threshold
        # can take one Adam step of the opposite sign in one of the runs: lr = 1e-2 .. 1.6e-2 here, seen at 3.6e-3 on an MI355X)
        d = (we - wg).abs() * _solid(ve, 1e-4)
        assert float(d.max()) <= 2e-2 and float((d > 1e-3).float().mean()) < 1e-3


@pytest.mark.parametrize("clip", [{"clip_value": 2e-3}, {"clip_norm": 5e-2}])
def test_loss_scale_with_clipping_eager_and_replayed(clip):
    """Dynamic loss scale + clipping in one step (gradaccum_keras_model.py:224-233: aggregate -> unscale -> clip -> apply): the
    scaled step lands on the weights of the unscaled clipped step, eagerly and as a graph replay (the un-scaling factor is
    read from the device-resident loss-scale state when the kernels run); the scale doubles on schedule."""
    batches = [_batch(300 + i) for i in range(5)]
    ls = {"initial_loss_scale": 1024.0, "growth_steps": 2, "multiplier": 2.0}
    res = {}
    for tag, use_graph, kw in (("plain", False, dict(clip)), ("eager", False, dict(clip, loss_scale=ls)),
                               ("graph", True, dict(clip, loss_scale=ls)), ("free", False, {})):
        model, step, opt = _setup("float32", 0.0, use_graph, **kw)
        for b in batches:
            step(b)
        torch.cuda.synchronize()
        res[tag] = (model.store.master.clone(), opt.v.clone(), step)
    want, v, _ = res["plain"]
    for tag in ("eager", "graph"):
        got, _, step = res[tag]
        assert float(((got - want).abs() * _solid(v)).max()) <= 5e-5, tag
        assert [float(x) for x in step._ls_state[:3]] == [4096.0, 1.0, 1.0], tag      # 5 good steps: x2 after 2 and 4
    assert res["graph"][2].replays == len(batches) - 1
    assert float(((res["free"][0] - want).abs() * _solid(v)).max()) > 1e-3       # the clip bites at these thresholds


def test_graph_replay_draws_new_dropout_masks_and_new_step_sizes():
    """The same batch replayed with a zero learning rate: the loss changes from step to step only through the masks."""
    model, step, opt = _setup("float32", 0.3, True, lr=0.0)
    b = _batch(5)
    losses = [float(step(b)) for _ in range(5)]
    assert step.replays == 4 and len(set(round(l, 6) for l in losses)) == 5, losses
    # eager steps of a fresh model with the same seeds see the same masks
    model2, step2, _ = _setup("float32", 0.3, False, lr=0.0)
    losses2 = [float(step2(b)) for _ in range(5)]
    assert max(abs(a - c) for a, c in zip(losses, losses2)) < 1e-5
    # and with a learning rate the replays follow the schedule (the step size lives in device memory)
    m3, s3, o3 = _setup("float32", 0.0, True, lr=1e-2)
    m4, s4, o4 = _setup("float32", 0.0, False, lr=1e-2)
    for i in range(4):
        s3(b), s4(b)
    torch.cuda.synchronize()
    assert float(((m3.store.master - m4.store.master).abs() * _solid(o4.v)).max()) < 2e-5


def test_graph_capture_is_cut_where_the_reducer_issues_buckets():
    """A reducer that behaves like world_size 2 (its exchange replaced by a recorder): the capture must be cut at every
    bucket, the buckets must be issued in the same order and ranges as in the eager step, and the result must be the eager
    result (gradients scaled by 1/2)."""
    from neurst_amd.training.distributed import GradientReducer

    def factory(log):
        def make(store):
            red = GradientReducer(store, bucket_bytes=1 << 16, min_bucket_bytes=1 << 14)
            red.world, red.active, red.overlap = 2, True, False
            red.issue = lambda s, e: log.append((s, e))
            return red
        return make
    batches = [_batch(200 + i) for i in range(4)]
    logs, out = {}, {}
    for use_graph in (False, True):
        logs[use_graph] = []
        model, step, opt = _setup("float32", 0.1, use_graph, reducer_factory=factory(logs[use_graph]))
        losses = [float(step(b)) for b in batches]
        torch.cuda.synchronize()
        out[use_graph] = (losses, model.store.master.clone(), step, opt.v.clone())
    per_step = len(logs[False]) // len(batches)
    assert per_step >= 3 and logs[True] == logs[False]
    assert len(out[True][2]._captured) == 1
    cap = next(iter(out[True][2]._captured.values()))
    planned = [bk for sg in cap.segments for bk in sg.buckets]
    assert planned == logs[False][:per_step] and not cap.segments[-1].buckets      # the tail (clip / Adam) follows every bucket
    assert sum(1 for sg in cap.segments if sg.wgrad is not None) >= 4             # weight gradients in graphs of their own
    assert max(abs(a - b) for a, b in zip(out[False][0], out[True][0])) < 1e-5
    assert float(((out[False][1] - out[True][1]).abs() * _solid(out[False][3])).max()) < 2e-5


def test_graph_replay_at_the_benchmark_configuration():
    """bench.py replays the step from HIP graphs by default: speech_transformer_s itself (12 + 6 layers, d = 256, V = 8008,
    dropout 0.1, bf16) at 80 x 900 frames -- 18 000 encoder rows, i.e. the one-launch feed-forward pair, the conv2 patch
    kernels, batched split-K second stages -- through a reducer that behaves like world_size 2.  Graph replay must follow the
    eager run (same masks: the step counter is a device scalar), cut the capture at the same buckets, and no captured
    segment may be an empty graph."""
    import warnings
    from neurst_amd.criterions import build_criterion
    from neurst_amd.data.datasets.synthetic_speech import SyntheticSpeechDataset
    from neurst_amd.optimizers import build_lr_schedule, build_optimizer
    from neurst_amd.tasks import build_task
    from neurst_amd.training.distributed import GradientReducer
    from neurst_amd.training.train_step import TrainStep
    from neurst_amd.utils import compat
    from neurst_amd.utils.hparams_sets import get_hyper_parameters
    B, T, F, V = 80, 900, 80, 8008
    L = T // 12
    hp = get_hyper_parameters("speech_transformer_s")
    task = build_task({"task.class": "speech2text", "task.params": {"audio_feature_dim": F, "vocab_size": V}})
    ds = SyntheticSpeechDataset({"batch_per_gpu": B, "frames": T, "feature_dim": F, "trg_len": L, "vocab_size": V,
                                 "ragged": True, "seed": 4321})
    it = ds.build_iterator(map_func=lambda b: task.example_to_input(b, compat.ModeKeys.TRAIN), shard_id=0, total_shards=1,
                           device=DEV)
    batches = [next(it) for _ in range(2)]
    out, logs = {}, {}
    for use_graph in (False, True):
        log = logs[use_graph] = []
        model = task.build_model(hp, device=DEV, dtype="bfloat16", seed=99, init_seed=42)
        crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
        opt = build_optimizer({"optimizer.class": hp["optimizer.class"], "optimizer.params": hp["optimizer.params"]})
        opt.bind(model.store)
        opt.learning_rate = build_lr_schedule({"lr_schedule.class": hp["lr_schedule.class"],
                                               "lr_schedule.params": hp["lr_schedule.params"]})
        red = GradientReducer(model.store)
        red.world, red.active, red.overlap = 2, True, False
        red.issue = lambda s, e, log=log: log.append((s, e))
        step = TrainStep(model, crit, opt, red, use_graph=use_graph)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            losses = [float(step(batches[i % 2])) for i in range(5)]
            torch.cuda.synchronize()
        assert not [w for w in caught if "Graph is empty" in str(w.message)], "an empty graph segment was captured"
        out[use_graph] = (losses, model.store.master.clone(), opt.v.clone(), step)
        del model, opt, step
    (le, we, ve, _), (lg, wg, _, gstep) = out[False], out[True]
    assert gstep.replays == 4 and len(gstep._captured) == 1
    per_step = len(logs[False]) // 5
    assert per_step >= 8 and logs[True] == logs[False]                      # the same >= 8 MiB buckets, in the same order
    assert all((e - s) * 4 >= (1 << 20) for s, e in logs[False]), "a bucket of less than 1 MiB travels alone"
    cap = next(iter(gstep._captured.values()))
    assert all(sg.main is not None or sg.wgrad is not None or sg.buckets or sg.join for sg in cap.segments), "a segment that carries nothing"
    for a, b in zip(le, lg):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (le, lg)
    d = (we - wg).abs() * _solid(ve, 1e-4)
    assert float((d > 1e-3).float().mean()) < 1e-3, float(d.max())


def test_capture_survives_garbage_of_an_earlier_captured_step():
    """A captured step that has become garbage (reference cycle) must not be collected in the middle of the next capture: its
    graphs' destructors ran inside the new capture and aborted the process (TrainStep._capture collects before it starts and
    keeps the cyclic collector off until the capture has ended).  The collector's thresholds are set so low here that an
    unprotected capture collects within its first few allocations."""
    import gc
    batches = [_batch(300 + i) for i in range(3)]
    model, step, _ = _setup("bfloat16", 0.1, True)
    for b in batches:
        step(b)
    torch.cuda.synchronize()
    assert step.replays >= 1
    cycle = [model, step]
    cycle.append(cycle)          # garbage only the cyclic collector can free, holding the first step's graphs
    del model, step, cycle
    old = gc.get_threshold()
    gc.set_threshold(1, 1, 1)
    try:
        model2, step2, _ = _setup("bfloat16", 0.1, True, seed=12)
        losses = [float(step2(b)) for b in batches]
        torch.cuda.synchronize()
    finally:
        gc.set_threshold(*old)
    assert gc.isenabled() and step2.replays >= 1 and all(l == l for l in losses)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--caller-stream"]])
def test_bench_line_carries_the_contract_fields(extra):
    """bench.py as the driver runs it (one JSON line on stdout): the fields of the measurement contract, the roofline object of the
    dominant kernel family measured live, and both loops (on the step's own stream -- the default -- and called from the default
    stream) complete with a finite loss."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "2",
                          "--no-cpu-baseline", "--roofline-steps", "1"] + extra,
                         cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and d["unit"] == "frames/s" and d["vs_baseline"] is None
    assert "speech_transformer_s" in d["config"]["workload"] and "model" not in d["config"]
    assert abs(d["value"] - 128 * 900 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["final_loss"] == d["final_loss"] and 0.0 < d["final_loss"] < 20.0
