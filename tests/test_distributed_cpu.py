"""Data-parallel path on CPU with the gloo backend, world_size 2 (no GPU needed): the reducer averages the flat
gradient buffer like hvd.Average, broadcasts rank 0's weights, reduces metrics with one packed all-reduce, and the
component hooks + finish() cover every element exactly once."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from neurst_amd.models import build_model
    from neurst_amd.training.distributed import GradientReducer, init_distributed
    from neurst_amd.utils import compat
    from neurst_amd.utils.hparams_sets import get_hyper_parameters
    r, lr, w = init_distributed(backend="gloo")
    assert (r, w) == (rank, world) and compat.get_distributed_worker_setting()[:2] == (rank, world)
    hp = get_hyper_parameters("speech_transformer_toy")
    # different init seeds per rank: the broadcast must make them identical
    model = build_model(hp, {"audio_feature_dim": 16, "audio_feature_channels": 1},
                        {"vocab_size": 11, "eos_id": 10, "bos_id": 9, "unk_id": 8}, device="cpu", dtype="float32",
                        init_seed=100 + rank)
    st = model.store
    red = GradientReducer(st, bucket_bytes=1024)  # tiny buckets: exercise the slicing
    before = st.master.clone()
    red.broadcast_parameters(0)
    gathered = [torch.zeros_like(st.master) for _ in range(world)]
    dist.all_gather(gathered, st.master)
    same = all(torch.equal(g, gathered[0]) for g in gathered)
    changed = (rank == 0) == torch.equal(before, st.master)

    # per-rank gradients g_r = (rank+1) * pattern ; hvd.Average -> mean over ranks
    pattern = torch.arange(st.total, dtype=torch.float32) % 7 - 3
    st.grad.copy_(pattern * (rank + 1))
    # the hooks fire in backward order: decoder, embedding, encoder, front end (see EncoderDecoderModel.backward)
    # ... with the per-layer reports in between (last layer first); a component report only adds what its layers left
    # (output_ln), adjacent reports coalesce, nothing is reduced twice
    for i in (1, 0):
        red.component_ready([f"TransformerDecoder/layer_{i}/"])
    red.component_ready(["TransformerDecoder/"])
    red.component_ready(["target_symbol_modality/"])
    for i in (1, 0):
        red.component_ready([f"TransformerEncoder/layer_{i}/"])
    red.component_ready(["TransformerEncoder/"])
    assert red._uncovered(*red.range_of(["TransformerDecoder/"])) == []
    scale = red.finish()          # covers what the hooks did not (input_audio_modality + padding gaps)
    assert red.last_messages >= 4
    avg = st.grad * scale
    expect = pattern * (sum(range(1, world + 1)) / world)
    ok_avg = torch.allclose(avg, expect, atol=1e-6)
    # a second step must work identically (state reset)
    st.grad.copy_(pattern * (rank + 1))
    scale2 = red.finish()
    ok_avg2 = torch.allclose(st.grad * scale2, expect, atol=1e-6)
    m = red.reduce_metrics({"loss": 1.0 + rank, "src_real_tokens": 100.0 * (rank + 1)})
    ranges = [red.range_of([p]) for p in ("target_symbol_modality/", "input_audio_modality/", "TransformerEncoder/",
                                          "TransformerDecoder/")]
    q.put((rank, same, changed, ok_avg, ok_avg2, scale, m, ranges, st.total))
    dist.destroy_process_group()


def test_gradient_reducer_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, changed, ok_avg, ok_avg2, scale, m, ranges, total in res:
        assert same, "broadcast did not equalise the weights"
        assert changed, "rank 0 must keep its weights, other ranks must receive them"
        assert ok_avg and ok_avg2, "all-reduce average is wrong"
        assert scale == 0.5
        assert m["loss"] == 3.0 and m["src_real_tokens"] == 300.0
        # the four components tile the flat buffer in registration (= forward) order without overlap
        flat = sorted(ranges)
        assert flat[0][0] == 0 and flat[-1][1] == total
        for (s0, e0), (s1, e1) in zip(flat, flat[1:]):
            assert e0 == s1


def test_single_process_reducer_is_identity():
    sys.path.insert(0, ROOT)
    from neurst_amd.runtime import ParamStore
    from neurst_amd.training.distributed import GradientReducer
    st = ParamStore()
    st.add("a/x", (5,), torch.ones(5))
    st.add("b/y", (3,), torch.ones(3))
    st.finalize("cpu", torch.float32)
    st.grad.fill_(2.0)
    red = GradientReducer(st)
    red.component_ready(["b/"])
    assert red.finish() == 1.0 and float(st.grad.sum()) == 2.0 * st.total
    assert red.reduce_metrics({"x": 4.0}) == {"x": 4.0}


def _forced_worker(port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", NST_DIST_FORCE="1")
    sys.path.insert(0, ROOT)
    from neurst_amd.runtime import ParamStore
    from neurst_amd.training.distributed import GradientReducer, init_distributed
    assert init_distributed(backend="gloo") == (0, 0, 1) and dist.is_initialized() and dist.get_world_size() == 1
    st = ParamStore()
    st.add("a/x", (5000,), torch.ones(5000))
    st.add("b/y", (3000,), torch.ones(3000))
    st.finalize("cpu", torch.float32)
    st.grad.fill_(2.0)
    plain = GradientReducer(st)                               # an initialised one-rank group alone changes nothing
    forced = GradientReducer(st, bucket_bytes=4096, min_bucket_bytes=1024, force=True)
    forced.component_ready(["b/"])
    scale = forced.finish()
    q.put((plain.active, forced.active, scale, forced.last_messages, float(st.grad.sum()), forced.reduce_metrics({"x": 4.0})))
    dist.destroy_process_group()


def test_forced_one_rank_group_runs_the_exchange_path_as_identities():
    """NST_DIST_FORCE=1: the process group exists for ONE rank and a reducer built with force=True issues its bucket
    all-reduces for real (sum over one rank = identity, factor 1/1) -- how a one-GPU box exercises the exchange path
    (tests/test_gpu_multi.py, bench.py).  Without force the reducer stays inactive even if a group is initialised."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(_free_port(), q))
    p.start()
    plain_active, forced_active, scale, messages, gsum, metrics = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert plain_active is False and forced_active is True and scale == 1.0
    assert messages >= 8 and gsum == 2.0 * 8000 and metrics == {"x": 4.0}      # 32 KB of gradients in <= 4 KB messages


def _bcast_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neurst_amd.training.distributed import GradientReducer

    class _S(object):
        pass
    st = _S()
    st.grad = torch.zeros(8)
    st.master = torch.full((8,), float(rank))
    st.params, st.total = {}, 8
    st.refresh_shadow = lambda: None
    red = GradientReducer(st, overlap=False)
    red.broadcast_parameters(0)
    m, v = torch.full((5,), 3.0 + rank), torch.full((5,), 7.0 + rank)
    red.broadcast_tensors([m, v], 0)                       # optimizer moments after a resume (exps/trainer.py)
    start = red.reduce_metrics({"start_step": 40.0 if rank == 0 else 0.0})["start_step"]
    q.put((rank, st.master.tolist(), m.tolist(), v.tolist(), start))
    dist.destroy_process_group()


def test_resume_state_is_broadcast_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29731
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    for rank, master, m, v, start in out:
        assert master == [0.0] * 8 and m == [3.0] * 5 and v == [7.0] * 5 and start == 40.0


def _wire_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from neurst_amd.models import build_model
    from neurst_amd.training.distributed import GradientReducer, init_distributed
    from neurst_amd.utils.hparams_sets import get_hyper_parameters
    init_distributed(backend="gloo")
    hp = get_hyper_parameters("speech_transformer_toy")
    model = build_model(hp, {"audio_feature_dim": 16, "audio_feature_channels": 1},
                        {"vocab_size": 11, "eos_id": 10, "bos_id": 9, "unk_id": 8}, device="cpu", dtype="float32", init_seed=1)
    st = model.store
    g = torch.Generator().manual_seed(5 + rank)
    mine = torch.randn(st.total, generator=g)
    others = [torch.randn(st.total, generator=torch.Generator().manual_seed(5 + r)) for r in range(world)]
    # (a) 16-bit wire: every rank's slice is rounded to bf16, summed in bf16 by the collective, widened again
    red = GradientReducer(st, bucket_bytes=4096, wire_dtype="bf16")
    st.grad.copy_(mine)
    issued = []
    orig = red._allreduce_slice
    red._allreduce_slice = lambda s, e: (issued.append((s, e)), orig(s, e))[1]
    # the model's own report order (EncoderDecoderModel.backward): the top layer carries output_ln
    for i in (1, 0):
        red.component_ready([f"TransformerDecoder/layer_{i}/"] + (["TransformerDecoder/output_ln/"] if i == 1 else []))
    red.component_ready(["TransformerDecoder/"])
    red.component_ready(["target_symbol_modality/"])
    for i in (1, 0):
        red.component_ready([f"TransformerEncoder/layer_{i}/"] + (["TransformerEncoder/output_ln/"] if i == 1 else []))
    red.component_ready(["TransformerEncoder/"])
    scale = red.finish()
    acc = others[0].to(torch.bfloat16)
    for o in others[1:]:
        acc = acc + o.to(torch.bfloat16)          # bf16 + bf16 -> bf16, like the collective's reduction
    # (two ranks: the one bf16 addition is the same whichever rank performs it; more ranks: the collective's order of the bf16
    # additions is its own -- the result must be bf16 values, i.e. summed on the 16-bit wire, close to the exact sum)
    ok_wire = torch.equal(st.grad, acc.float()) if world == 2 else torch.equal(st.grad, st.grad.to(torch.bfloat16).float())
    exact = sum(others)
    rel = float((st.grad - exact).norm() / exact.norm())
    # (b) no slice of a few hundred elements travels alone: output_ln was part of its top layer's report
    ln = [red.range_of([f"{s}/output_ln/"]) for s in ("TransformerDecoder", "TransformerEncoder")]
    alone = [r for r in issued if r in ln]
    # (c) fp16 wire: gradients are scaled by 1/world BEFORE the cast, so values whose SUM leaves the fp16 range (but whose
    # average does not) survive, and finish() hands the optimizer a factor of 1 instead of 1/world
    red16 = GradientReducer(st, bucket_bytes=4096, wire_dtype="fp16")
    big = torch.full((st.total,), 40000.0) + rank          # 2 x 40000 > 65504 = fp16 max
    st.grad.copy_(big)
    scale16 = red16.finish()
    want16 = torch.full((st.total,), 40000.0) + (world - 1) / 2.0
    fp16_ok = bool(torch.isfinite(st.grad).all()) and float((st.grad * scale16 - want16).abs().max()) <= 40000.0 * 2e-3
    q.put((rank, ok_wire, rel, scale, alone, len(issued), fp16_ok, scale16))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_bf16_wire_and_no_standalone_output_ln_message_gloo(world):
    """16-bit gradient wire (the reference's fp16 compression, neurst/training/training_utils.py:381-384) and the report
    layout that keeps the 2 KB output_ln slices inside their top layer's message; at 2 and at 8 ranks (one node of the
    benchmark: the bucket order is identical on all ranks or the collectives would not match up)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_wire_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_wire, rel, scale, alone, n, fp16_ok, scale16 in res:
        assert ok_wire, "the 16-bit wire must carry bf16-rounded slices and widen the bf16 sum"
        assert rel < (1e-2 if world == 2 else 2.5e-2) and scale == 1.0 / world
        assert alone == [] and n >= 4
        assert fp16_ok and scale16 == 1.0, "fp16 wire: pre-scaled by 1/world, no overflow of the sum"
