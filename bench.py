#!/usr/bin/env python3
"""Headline benchmark: audio frames/sec of SpeechTransformer-base (speech_transformer_s) bf16 TRAINING steps
(forward + label-smoothed CE + backward + gradient all-reduce + Adam) on synthetic MuST-C-shaped batches.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W      # no launcher environment: starts the N ranks itself (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see README / DESIGN.md §Measurement for the fields).  `value` is the whole-job
frames/s with the inputs already resident in HBM.  `roofline` is measured live with HIP events around every launch
of the MFMA kernel families in extra steps right after the timed region, on the stream each launch goes to; the object
prices the family with the largest share of GPU time (the dense GEMMs), all families are in `roofline_families`.  `cpu_baseline` times the CPU oracle (oracle/neurst_oracle.py, a torch-CPU
restatement of the reference math -- TensorFlow is not installable here) on the host cores, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

# (GPU_MAX_HW_QUEUES -- ONE hardware queue per stream-priority class -- is applied by neurst_amd.runtime.
# configure_training_process() in main(), before the first HIP call and AFTER the N > 1 autotune has had its say)
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PROBED = ("gemm", "gemm_rows", "gemm_wgrad_group", "ffn_fwd", "ffn_bwd", "conv2_fwd", "conv2_dgrad", "conv2_wgrad", "attention_fwd", "attention_bwd")
FAMILY_KERNELS = {
    "gemm": "dense_gemm_kernel_v3 family (every nst_gemm launch of a step: projections, logits, front dense, their input "
            "and weight gradients incl. split-K reduce; work = sum 2MNK)",
    "gemm_rows": "rowgemm_kernel (whole-row products of round 6: nst_gemm_add_layernorm_fwd / nst_gemm_layernorm_bwd / "
                 "nst_gemm_rowdot256 -- a product of the nst_gemm family TOGETHER with the wrapper's dropout + residual + "
                 "LayerNorm forward, or the LayerNorm backward, of its rows; work = 2MNK only, bytes = operands + the row "
                 "stages' f32 / bf16 rows: the HBM bound is the relevant one)",
    "gemm_wgrad_group": "gemm256_group_kernel (nst_gemm_wgrad_group: the weight gradients of a layer stack as ONE launch of "
                        "256 x 256 phase-staggered tiles, no split-K; work = sum 2MNK)",
    "ffn_fwd": "ffn_pair8_kernel<fwd> (dense1 + ReLU + dropout + dense2 in one launch; work = 4*M*d*ffn)",
    "ffn_bwd": "ffn_pair8_kernel<bwd> (d hidden + gate + d input in one launch; work = 4*M*d*ffn)",
    "conv2_fwd": "conv2_fwd256_kernel (conv2 forward as GEMM M=B*T2*F2, N=C, K=9C on the 256 x 256 phase-staggered tile core)",
    "conv2_dgrad": "conv2_dgrad256_kernel (four stride-parity classes per 256-pixel tile, same core)",
    "conv2_wgrad": "conv2_wgrad256_kernel + conv_splitk_reduce_kernel (nine tap products, reduction in slices, same core)",
    "attention_fwd": "attn_fwd_kernel", "attention_bwd": "attn_bwd_head8_kernel (one workgroup per (batch, head))",
}
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3     # f32-input MFMA peak
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--model", default="speech_transformer_s")
    ap.add_argument("--batch", type=int, default=128, help="utterances per GPU per step")
    ap.add_argument("--frames", type=int, default=900)
    ap.add_argument("--vocab", type=int, default=8008)
    ap.add_argument("--dropout", type=float, default=None, help="override the hparams set's dropout (0.1)")
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --batch is the GLOBAL batch, split over the ranks (default: per-GPU batch, weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--main-stream-priority", type=int, default=0,
                    help="-1: run the step on a high-priority HIP stream (the weight-gradient stream keeps the default priority)")
    ap.add_argument("--graph", dest="graph", action="store_true", default=None,
                    help="replay the step from captured HIP graphs (training/train_step.py graph mode): the default -- on one rank "
                         "since round 3 (~1 ms instead of ~12 ms of host time per step), for any number of ranks since round 4 "
                         "(the exchange is replayed eagerly between the captured segments); NST_TRAIN_GRAPH=0|1 or the flags decide")
    ap.add_argument("--eager", dest="graph", action="store_false", help="eager launches instead of graph replay")
    ap.add_argument("--wire", default=os.environ.get("NST_DIST_WIRE", "fp32"), choices=["fp32", "bf16", "fp16"],
                    help="gradient dtype on the wire (16-bit: the reference's fp16 compression, training_utils.py:381-384)")
    ap.add_argument("--roofline-steps", type=int, default=3, help="extra un-timed steps with per-launch HIP events (0 = skip)")
    ap.add_argument("--no-autotune", dest="autotune", action="store_false", default=True,
                    help="N > 1 only: skip the short child runs that choose {hardware queues per class} x {RCCL channels} x "
                         "{exchange carrier} before the timed run (the variables are read when HIP / RCCL initialise, so each "
                         "candidate needs fresh processes; every candidate and the choice are reported under `autotune`)")
    ap.add_argument("--caller-stream", action="store_true",
                    help="call the step from the default stream (a hand-over to the step's stream and back per call) instead of "
                         "running the loop on the step's stream")
    ap.add_argument("--autotune-budget", type=float, default=240.0, help="wall-clock cap of all autotune child runs, seconds")
    ap.add_argument("--autotune-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def algorithmic_flops(model_args, B, T, F, L, V):
    """Forward FLOPs per step (SURVEY §8(d) formula); training = 3x."""
    d, C = model_args["modality.dim"], model_args["modality.source.channels"]
    H, ffn = model_args["encoder.num_attention_heads"], model_args["encoder.filter_size"]
    Ne, Nd = model_args["encoder.num_layers"], model_args["decoder.num_layers"]
    T1, F1 = (T + 1) // 2, (F + 1) // 2
    T2, F2 = (T1 + 1) // 2, (F1 + 1) // 2
    M, Md, dh = B * T2, B * L, d // H
    conv1 = 2 * B * T1 * F1 * 9 * C
    conv2 = 2 * B * T2 * F2 * 9 * C * C
    dense = 2 * M * F2 * C * d
    enc = Ne * (2 * M * d * 3 * d + 4 * B * H * T2 * T2 * dh + 2 * M * d * d + 4 * M * d * ffn)
    dec = Nd * (2 * Md * d * 3 * d + 4 * B * H * L * L * dh + 2 * Md * d * d + 2 * Md * d * d + 2 * M * d * 2 * d
                + 4 * B * H * L * T2 * dh + 2 * Md * d * d + 4 * Md * d * ffn)
    logits = 2 * Md * d * V
    return {"conv1": conv1, "conv2": conv2, "dense": dense, "encoder": enc, "decoder": dec, "logits": logits,
            "forward": conv1 + conv2 + dense + enc + dec + logits}


def latest_profile(suffix):
    """profiles/rNN_<suffix> of the highest round that has one (the PMC summaries are captured by separate rocprofv3 --pmc passes
    of this command and committed; this run does not re-measure them)."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{suffix}")))
    return hits[-1] if hits else None


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (os.cpu_count() reports the
    host's cores inside a container, and oversubscribing torch's thread pool slows the oracle by orders of magnitude)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, min(n, 64))


def cpu_baseline_worker(model_name, B, T, F, L, V):
    """Oracle (port of the reference math) fwd+bwd+Adam on the host cores, bounded sample.  Runs in a child
    process (see cpu_baseline) so that it can never stall the GPU measurement."""
    import neurst_amd.models  # noqa: F401  (registers the hparams sets)
    from neurst_amd.utils.hparams_sets import get_hyper_parameters
    from oracle import neurst_oracle as O
    p = get_hyper_parameters(model_name)["model.params"]
    cfg = {"num_enc": p["encoder.num_layers"], "num_dec": p["decoder.num_layers"],
           "num_heads": p["encoder.num_attention_heads"], "layer_norm": True, "d_model": p["modality.dim"],
           "channels": p["modality.source.channels"], "ffn": p["encoder.filter_size"]}
    cores = usable_cores()
    torch.set_num_threads(cores)
    W = O.init_speech_transformer_weights(cfg, V, F, 1, seed=42)
    g = torch.Generator().manual_seed(1234)
    trg = torch.randint(0, V - 3, (B, L), generator=g)
    trg[:, -1] = V - 1
    inputs = {"src": torch.randn(B, T, F, 1, generator=g), "src_length": torch.full((B,), T), "trg": trg,
              "trg_length": torch.full((B,), L), "trg_input": torch.cat([torch.full((B, 1), V - 2), trg[:, :-1]], 1)}
    m = {k: torch.zeros_like(v) for k, v in W.items()}
    v_ = {k: torch.zeros_like(v) for k, v in W.items()}
    times = []
    t_start = time.perf_counter()
    for t in range(1, 5):
        t0 = time.perf_counter()
        _, _, grads = O.train_step_reference(W, inputs, cfg, 0.1)
        for k in W:
            W[k], m[k], v_[k] = O.keras_adam_step(W[k], grads[k], m[k], v_[k], t, 1e-4)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > 40 and len(times) >= 2:
            break
    dt = min(times[1:]) if len(times) > 1 else times[0]
    return {"value": B * T / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle fwd+bwd+Adam, fp32, batch {B} x {T} frames, best of {max(len(times) - 1, 1)} timed steps "
                      f"({dt * 1e3:.0f} ms/step); torch {torch.__version__} CPU, {cores} threads"}


def cpu_baseline(args, T, F, L, V, timeout=150):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--model", args.model,
           "--cpu-batch", str(args.cpu_batch), "--frames", str(T), "--vocab", str(V)]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "frames/s", "cores": usable_cores(), "kind": "port",
                "sample": "failed: " + (out.stderr.strip().splitlines() or ["no output"])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "frames/s", "cores": usable_cores(), "kind": "port",
                "sample": f"timed out after {timeout}s"}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher environment: start the N ranks with torch.distributed.run (one process
    per GPU, rendezvous on 127.0.0.1, a free port) and pass the command line through; rank 0 of the children prints the JSON
    line on the inherited stdout.  Returns the launcher's exit code."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and os.environ.get("NST_DIST_BACKEND") != "gloo":
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} device(s) visible (RCCL needs one GPU per rank; "
                         "NST_DIST_BACKEND=gloo runs the host-staged control-flow rehearsal on shared devices)")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, NST_BENCH_CHILD="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


AUTOTUNE_CANDIDATES = [
    # (label, environment of the candidate) -- the first one is the default configuration; one factor changes at a time
    ("hwq1_ch16_torch", {"GPU_MAX_HW_QUEUES": "1", "NCCL_MAX_NCHANNELS": "16", "NST_DIST_NATIVE": "0"}),
    ("hwq2_ch16_torch", {"GPU_MAX_HW_QUEUES": "2", "NCCL_MAX_NCHANNELS": "16", "NST_DIST_NATIVE": "0"}),
    ("hwq1_ch32_torch", {"GPU_MAX_HW_QUEUES": "1", "NCCL_MAX_NCHANNELS": "32", "NST_DIST_NATIVE": "0"}),
    ("hwq1_ch16_native", {"GPU_MAX_HW_QUEUES": "1", "NCCL_MAX_NCHANNELS": "16", "NST_DIST_NATIVE": "1"}),
]
_LAUNCHER_VARS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK",
                  "ROLE_WORLD_SIZE", "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT")


def _free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def autotune_path():
    """The hand-off file of one run: keyed by the launcher's rendezvous port, its pid AND its start time (field 22 of
    /proc/<pid>/stat, identical for every rank of the run), so a file left by an earlier run with the same port and pid is never
    this run's file.  Deleted by rank 0 once every rank has confirmed the choice (confirm_autotune)."""
    import tempfile
    ppid = os.getppid()
    try:
        start = open(f"/proc/{ppid}/stat").read().rsplit(")", 1)[1].split()[19]
    except Exception:
        start = "0"
    return os.path.join(tempfile.gettempdir(), f"nst_autotune_{os.environ.get('MASTER_PORT', '0')}_{ppid}_{start}.json")


def confirm_autotune(tune, rank):
    """Behind init_process_group: every rank must hold the same chosen environment (a rank that timed out waiting for rank 0's
    file would otherwise run the timed steps with another carrier / channel count and hang the first collective); then the
    hand-off file goes away."""
    import torch.distributed as dist
    mine = sorted((tune or {}).get("chosen_env", {}).items())
    every = [None] * dist.get_world_size()
    dist.all_gather_object(every, mine)
    if any(e != every[0] for e in every):
        raise SystemExit(f"bench.py: ranks disagree on the autotuned environment: {every}")
    dist.barrier()
    if rank == 0:
        try:
            os.remove(autotune_path())
        except OSError:
            pass


def autotune(args, rank, world):
    """N > 1, BEFORE this process touches HIP: rank 0 times each candidate as a short child job of its own (same N ranks on the
    same GPUs -- the parent ranks hold nothing on them yet), picks the fastest and tells the other ranks through a file next
    to the launcher's rendezvous port; every rank then applies the winner's environment and carries on into the timed run.
    Explicit settings in the caller's environment are never overridden (such a variable is simply not a factor).  Any failure
    -- a candidate that crashes, hangs past its share of --autotune-budget, or prints no JSON line -- is recorded and skipped;
    with no usable candidate the defaults stay.  Returns the report that goes into the JSON line."""
    import subprocess
    import tempfile
    path = autotune_path()
    fixed = {k: os.environ[k] for k in ("GPU_MAX_HW_QUEUES", "NCCL_MAX_NCHANNELS", "NST_DIST_NATIVE") if k in os.environ}
    if rank != 0:
        t_end = time.time() + args.autotune_budget + 90.0
        while time.time() < t_end:
            if os.path.exists(path):
                try:
                    report = json.load(open(path))
                    break
                except Exception:
                    pass
            time.sleep(0.2)
        else:
            report = {"chosen": None, "note": "rank 0 never published a choice: defaults"}
        for k, v in (report.get("chosen_env") or {}).items():
            os.environ[k] = v
        return report
    t0 = time.time()
    report = {"fixed_by_caller": fixed, "candidates": [], "budget_s": args.autotune_budget}
    passthrough = ["--gpus", str(world), "--steps", "6", "--warmup", "3", "--roofline-steps", "0", "--no-cpu-baseline",
                   "--autotune-child", "--dtype", args.dtype, "--model", args.model, "--batch", str(args.batch),
                   "--frames", str(args.frames), "--vocab", str(args.vocab), "--wire", args.wire]
    if args.graph is not None:
        passthrough.append("--graph" if args.graph else "--eager")
    if args.ragged:
        passthrough.append("--ragged")
    if args.strong:
        passthrough.append("--strong")
    seen = set()
    for label, cand in AUTOTUNE_CANDIDATES:
        env_c = {k: v for k, v in cand.items() if k not in fixed}
        key = tuple(sorted(env_c.items()))
        if key in seen:                    # the caller fixed this candidate's factor: it collapses onto an earlier one
            continue
        seen.add(key)
        left = args.autotune_budget - (time.time() - t0)
        entry = {"label": label, "env": env_c}
        report["candidates"].append(entry)
        if left < 20.0:
            entry["skipped"] = "autotune budget spent"
            continue
        env = {k: v for k, v in os.environ.items() if k not in _LAUNCHER_VARS and not k.startswith("TORCHELASTIC")}
        env.update(env_c)
        env["NST_BENCH_CHILD"] = "1"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + passthrough
        t1 = time.time()
        try:
            # own session: a candidate that hangs is killed with its whole process group (exact pgid, never by pattern)
            proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
            try:
                so, se = proc.communicate(timeout=min(100.0, left))
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(proc.pid, signal.SIGKILL)
                so, se = proc.communicate()
                entry["error"] = "timed out"
            line = next((ln for ln in reversed((so or "").strip().splitlines()) if ln.startswith("{")), None)
            if line is not None and "error" not in entry:
                d = json.loads(line)
                entry["ms_per_step"] = d.get("ms_per_step")
                entry["exchange"] = d.get("exchange")
            elif "error" not in entry:
                entry["error"] = "no JSON line (rc %s): %s" % (proc.returncode, ((se or "").strip().splitlines() or ["no stderr"])[-1][:200])
        except Exception as e:      # never let the tuner take the measurement down
            entry["error"] = f"{type(e).__name__}: {e}"[:200]
        entry["seconds"] = round(time.time() - t1, 1)
    ok = [c for c in report["candidates"] if c.get("ms_per_step")]
    best = min(ok, key=lambda c: c["ms_per_step"]) if ok else None
    # a candidate replaces the default only when it wins by more than the run-to-run noise of a 6-step sample
    base = next((c for c in ok if c["label"] == AUTOTUNE_CANDIDATES[0][0]), None)
    if best is not None and base is not None and best is not base and best["ms_per_step"] > 0.98 * base["ms_per_step"]:
        best = base
    report["chosen"] = best["label"] if best else None
    report["chosen_env"] = dict(best["env"]) if best else {}
    report["seconds"] = round(time.time() - t0, 1)
    tmp = path + ".tmp"
    json.dump(report, open(tmp, "w"))
    os.replace(tmp, path)
    for k, v in report["chosen_env"].items():
        os.environ[k] = v
    return report


def main():
    if os.environ.get("NST_BENCH_HANG_DUMP_S"):   # debugging aid: dump every thread's stack and exit if the run takes longer
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["NST_BENCH_HANG_DUMP_S"]), exit=True)
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_worker(args.model, args.cpu_batch, args.frames, 80, max(1, args.frames // 12),
                                             args.vocab)))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("NST_BENCH_CHILD") != "1":
        sys.exit(self_launch(args))
    # process-wide settings first: nothing below may load the HIP library's code objects or touch the device before them
    from neurst_amd.runtime import configure_training_process
    tune = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and args.autotune and not args.autotune_child:
        tune = autotune(args, int(os.environ.get("RANK", "0")), int(os.environ["WORLD_SIZE"]))
    configure_training_process()
    from neurst_amd import kernels as K
    from neurst_amd.criterions import build_criterion
    from neurst_amd.data.datasets.synthetic_speech import SyntheticSpeechDataset
    from neurst_amd.models import build_model  # noqa: F401
    from neurst_amd.optimizers import build_lr_schedule, build_optimizer
    from neurst_amd.tasks import build_task
    from neurst_amd.training.distributed import GradientReducer, init_distributed
    from neurst_amd.training.train_step import TrainStep
    from neurst_amd.utils import compat
    from neurst_amd.utils.hparams_sets import get_hyper_parameters
    import torch.distributed as dist

    # RCCL's footprint next to the step's kernels: a channel is one 256-thread workgroup on a CU of its own for the
    # duration of a collective.  A step exchanges 117 MB of fp32 gradients in >= 8 MiB messages that are issued while the
    # front end's backward (~2.5 ms) still runs; 16 channels move that several times over (>= 100 GB/s over the 7 xGMI links)
    # and leave 240 of the 256 CUs to the compute stream, whose kernels are sized for whole-chip rounds (one 8-wave workgroup
    # per CU).  RCCL's default for large messages takes several times as many.  Override with NCCL_MAX_NCHANNELS.
    os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")
    rank, local_rank, world = init_distributed()
    if tune is not None and world > 1:
        confirm_autotune(tune, rank)
    if args.graph is None:
        # graph replay is the default for ANY number of ranks since round 4: host issue time is ~1 ms instead of ~11 ms per
        # step, and the exchange is replayed eagerly between the captured segments (TrainStep); soak log of the forced
        # exchange path over RCCL in graph mode: profiles/r04_graph_rccl_soak.log.  NST_TRAIN_GRAPH=0 / --eager: eager launches
        env = os.environ.get("NST_TRAIN_GRAPH")
        args.graph = (env != "0") if env is not None else True
    if world != args.gpus and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    dev = f"cuda:{torch.cuda.current_device()}"      # init_distributed pinned it (LOCAL_RANK)
    dtype = "bfloat16" if args.dtype == "bf16" else "float32"
    hp = get_hyper_parameters(args.model)
    if args.dropout is not None:
        for k in list(hp["model.params"]):
            if k.endswith("dropout_rate"):
                hp["model.params"][k] = args.dropout
    B, T, F, V = args.batch, args.frames, 80, args.vocab
    if args.strong:
        if B % world:
            raise SystemExit(f"--strong: global batch {B} is not divisible by {world} ranks")
        B //= world
    L = max(1, T // 12)
    task = build_task({"task.class": "speech2text", "task.params": {"audio_feature_dim": F, "vocab_size": V}})
    model = task.build_model(hp, device=dev, dtype=dtype, seed=1234 + rank, init_seed=42)
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    opt = build_optimizer({"optimizer.class": hp["optimizer.class"], "optimizer.params": hp["optimizer.params"]})
    opt.bind(model.store)
    opt.learning_rate = build_lr_schedule({"lr_schedule.class": hp["lr_schedule.class"],
                                           "lr_schedule.params": hp["lr_schedule.params"]})
    # NST_DIST_FORCE=1 (with one rank): run the exchange path -- buckets, communication stream, RCCL -- on a one-GPU box
    reducer = GradientReducer(model.store, force=os.environ.get("NST_DIST_FORCE", "0") == "1",
                              wire_dtype=args.wire)
    reducer.broadcast_parameters(0)
    step_fn = TrainStep(model, crit, opt, reducer, use_graph=args.graph)
    ds = SyntheticSpeechDataset({"batch_per_gpu": B, "frames": T, "feature_dim": F, "trg_len": L, "vocab_size": V,
                                 "ragged": args.ragged, "seed": 1234})
    it = ds.build_iterator(map_func=lambda b: task.example_to_input(b, compat.ModeKeys.TRAIN), shard_id=rank,
                           total_shards=world, device=dev)
    batches = [next(it) for _ in range(4)]  # resident in HBM before the timed region

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    if args.main_stream_priority != 0:
        main_stream = torch.cuda.Stream(device=dev, priority=args.main_stream_priority)
        main_stream.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(main_stream)
    elif step_fn.stream is not None and not args.caller_stream:
        # the batches are resident: the loop runs ON the step's stream, so a call needs no hand-over between the caller's stream
        # and the step's (two event waits per step, ~30 us of idle GPU in the trace of round 6); --caller-stream: the old loop
        step_fn.stream.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(step_fn.stream)

    verbose = os.environ.get("NST_BENCH_VERBOSE", "0") == "1"
    for i in range(args.warmup):
        step_fn(batches[i % len(batches)])
        if verbose:
            torch.cuda.synchronize()
            print(f"[bench] rank {rank}: warm-up step {i} done", file=sys.stderr, flush=True)
    barrier()
    reducer.diag = bool(reducer.active)      # HIP events around the exchange of every timed step (three records per step)
    t0 = time.perf_counter()
    loss = None
    for i in range(args.steps):
        loss = step_fn(batches[i % len(batches)])
    t_issued = time.perf_counter() - t0      # the host has queued every launch of the K steps (no sync inside)
    barrier()
    elapsed = time.perf_counter() - t0
    reducer.diag = False
    exchange = reducer.exchange_report()     # this rank's: exposed wait, span, bytes, bus-rate bound (None without an exchange)
    if exchange is not None and world > 1:   # the slowest rank's exposed wait is the one the step time contains
        worst = torch.tensor([exchange["exchange_exposed_ms"], exchange["exchange_span_ms"]], dtype=torch.float64, device=dev)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        exchange["exchange_exposed_ms_max_over_ranks"], exchange["exchange_span_ms_max_over_ranks"] = worst.tolist()
    # roofline pass (un-timed, after the measurement): the same steps with HIP events around every launch of the MFMA kernel
    # families, on the stream each launch goes to -- durations are therefore IN-STEP durations (the weight-gradient stream
    # shares the CUs with the dgrad chain), the same thing `rocprofv3 --kernel-trace --stats` of this command reports
    # EVERY rank runs these steps (they contain the gradient exchange: a rank that skipped them would leave rank 0 alone in
    # its all-reduce); only rank 0 records events
    probe = {}
    if args.roofline_steps > 0:
        step_fn.use_graph = False          # per-launch events need eager launches (same kernels, same streams)
        if rank == 0:
            K.PROBE.start(PROBED)
        for i in range(args.roofline_steps):
            step_fn(batches[i % len(batches)])
        if rank == 0:
            probe = K.PROBE.stop()
    barrier()
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    loss_val = float(loss)

    if rank != 0:
        return
    frames = world * B * T * args.steps
    fl = algorithmic_flops(hp["model.params"], B, T, F, L, V)
    step_flops = 3 * fl["forward"]
    value = frames / elapsed
    peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
    families = {}
    for name, rows in probe.items():
        ms, work, nbytes = sum(r[0] for r in rows), sum(r[1] for r in rows), sum(r[2] for r in rows)
        families[name] = {"kernel": FAMILY_KERNELS.get(name, name), "bound": "mfma", "peak": peak, "unit": "TFLOP/s",
                          "launches_per_step": len(rows) / args.roofline_steps, "ms_per_step": ms / args.roofline_steps,
                          "avg_launch_ms": ms / max(len(rows), 1), "algorithmic_flops_per_step": work / args.roofline_steps,
                          "achieved": (work / (ms * 1e-3) / 1e12) if ms > 0 else None, "traffic": None}
        if families[name]["achieved"] is not None:
            families[name]["frac"] = families[name]["achieved"] / peak
        if nbytes > 0 and ms > 0:
            # the same launches against the HBM roof (d_model = 256 puts most of these GEMMs below the machine balance of
            # 312 FLOP/B): algorithmic operand + output bytes per launch / its duration / 8 TB/s
            families[name]["algorithmic_bytes_per_step"] = nbytes / args.roofline_steps
            families[name]["arithmetic_intensity_flop_per_byte"] = work / nbytes
            families[name]["hbm_bound"] = {"bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                           "unit": "GB/s", "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # the roofline object prices the family that takes the most GPU time in the step
    roofline = None
    if families:
        top = max(families, key=lambda n: families[n]["ms_per_step"])
        roofline = dict(families[top])
        roofline["selected_as"] = "largest share of in-step GPU time among the MFMA kernel families (see roofline_families)"
        pmc = latest_profile(f"pmc_{top}.json")
        if args.dtype == "bf16" and B == 128 and T == 900 and pmc:
            # HBM bytes per step of this family from the committed rocprofv3 --pmc passes of the same command
            # (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE); not re-measured by this run
            try:
                t = json.load(open(pmc))
                roofline["traffic"] = t["hbm_bytes_per_launch"]          # per launch, like `achieved`
                roofline["traffic_per_step"] = t["hbm_bytes_per_step"]
                roofline["algorithmic_flops_per_launch"] = roofline["algorithmic_flops_per_step"] / roofline["launches_per_step"]
                roofline["traffic_source"] = os.path.relpath(pmc, ROOT)
                roofline["traffic_captured_at_commit"] = t.get("captured_at_commit")   # stale once the GEMM kernels change
            except Exception:
                pass
    ffn_util = None
    pmc_ffn = latest_profile("pmc_ffn_gemm.json")
    if pmc_ffn:
        try:
            t = json.load(open(pmc_ffn))
            ffn_util = {"source": os.path.relpath(pmc_ffn, ROOT), "captured_at_commit": t.get("captured_at_commit"), **t["summary"]}
        except Exception:
            pass
    out = {
        "metric": ("audio frames/sec, SpeechTransformer-base (speech_transformer_s) training, whole job" if args.model == "speech_transformer_s"
                   else f"audio frames/sec, SpeechTransformer ({args.model}) training, whole job"),
        "value": value, "unit": "frames/s", "value_per_gpu": value / world, "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.model} train step: B={B}/GPU x T={T} frames x F={F} mel, L={L}, V={V}, "
                               f"dropout {hp['model.params']['encoder.ffn_dropout_rate']}, label smoothing 0.1, "
                               f"Adam+Noam, {'ragged' if args.ragged else 'full-length'} inputs",
                   "global_batch": world * B, "seq_len": T, "parallelism": f"dp{world}"},
        "model_tflops_per_s": step_flops * args.steps * world / elapsed / 1e12,
        "model_mfma_frac": step_flops * args.steps / elapsed / 1e12 / peak,
        "final_loss": loss_val,
        "roofline": roofline,
        "roofline_families": families,
        "ffn_gemm_mfma_utilisation": ffn_util,
        "host_issue_ms_per_step": t_issued / args.steps * 1e3,   # ~ ms_per_step means the host, not the GPU, paces the step
        "hip_graph": bool(args.graph), "graph_replays": getattr(step_fn, "replays", 0),
        "rccl_world_size": (dist.get_world_size() if dist.is_initialized() else 1),
        "dist_backend": (dist.get_backend() if dist.is_initialized() else None),
        "gradient_wire_dtype": args.wire,
        "exchange_path_active": bool(reducer.active),
        "reducer_messages_per_step": getattr(reducer, "last_messages", None),
        "exchange": exchange,
        "autotune": tune,
        # every switch of the environment that can change what is measured (the library reads NST_*, HIP / RCCL the others)
        "env": {k: v for k, v in sorted(os.environ.items())
                if k.startswith(("NST_", "NCCL_", "RCCL_", "GPU_MAX_HW_QUEUES", "HSA_ENABLE", "HIP_VISIBLE", "ROCR_VISIBLE"))},
        "hw_queues_note": configure_training_process(),
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, T, F, L, V)
    try:   # RCCL prints its version banner through C stdio, which a pipe only sees at exit: flush it so the JSON line stays last
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    reducer.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    try:
        main()
    finally:
        import torch.distributed as _dist
        if _dist.is_available() and _dist.is_initialized():
            _dist.destroy_process_group()
