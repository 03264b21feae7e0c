"""Execution context of the MI355X path: device, compute dtype, parameter store, dropout streams.

One process drives one GPU.  Parameters live in flat fp32 buffers (master weights, gradients, Adam
moments) plus -- in bf16 mode -- a flat bf16 shadow that the MFMA kernels read and the fused Adam kernel
rewrites.  Gradients are written by the backward kernels straight into the flat gradient buffer, in
forward registration order, so the data-parallel reducer can all-reduce contiguous buckets as soon as the
backward pass has finished the layers they cover (neurst_amd/training/distributed.py).
"""
import collections
import contextlib
import math
import os

import torch

_ALIGN = 8  # elements: keeps every parameter 32-byte (fp32) / 16-byte (bf16) aligned for vector loads


class Param(object):
    __slots__ = ("name", "shape", "offset", "numel", "data", "grad", "compute", "_init", "tail_pad")

    def __init__(self, name, shape, init, tail_pad=0):
        self.name, self.shape = name, tuple(shape)
        self.numel = int(math.prod(self.shape)) if len(self.shape) else 1
        self.tail_pad = int(tail_pad)     # zero elements kept behind the parameter in every flat buffer (ParamStore.padded_views)
        self._init = init
        self.offset = -1
        self.data = self.grad = self.compute = None

    def __repr__(self):
        return f"Param({self.name}, {self.shape})"


class TransposedCopy(object):
    """Handle of a transposed bf16 copy of a 2-D parameter (ParamStore.add_transposed)."""
    __slots__ = ("param", "offset", "t")

    def __init__(self, param):
        self.param, self.offset, self.t = param, -1, None


class PackedCopy(object):
    """Handle of a packed copy of several [rows, cols_i] kernels side by side (ParamStore.add_packed): `w` [rows, sum cols_i] in
    the compute dtype, `b` [sum cols_i] fp32 (the biases)."""
    __slots__ = ("kernels", "biases", "w", "b", "col0")

    def __init__(self, kernels, biases):
        self.kernels, self.biases, self.w, self.b, self.col0 = list(kernels), list(biases), None, None, []


class ParamStore(object):
    def __init__(self):
        self.params = collections.OrderedDict()
        self.finalized = False
        self._touched = set()
        self.accumulate_all = False
        self._transposed = []     # (param, handle): bf16 copies stored transposed (the fused feed-forward's forward operands)
        self._packed = []         # PackedCopy handles: kernels of several layers side by side as one GEMM operand

    def add(self, name, shape, init, tail_pad=0):
        """init: CPU float tensor of `shape` (the reference's initializer already applied).  tail_pad: elements that stay ZERO
        behind the parameter in the master / shadow / gradient buffers, so that kernels may read (and write zero gradients to) a
        few rows past its end -- an embedding table whose vocabulary is not a multiple of 8 is a GEMM operand of 8-row granularity
        that way (padded_views)."""
        if self.finalized:
            raise RuntimeError("ParamStore already finalized")
        if name in self.params:
            raise ValueError(f"duplicate variable name: {name}")
        init = torch.as_tensor(init, dtype=torch.float32).reshape(shape)
        p = Param(name, shape, init, tail_pad)
        self.params[name] = p
        return p

    def padded_views(self, p, rows):
        """(compute, grad) views of parameter p with `rows` leading rows, the rows past p.shape[0] lying in its zero tail pad."""
        inner = p.numel // p.shape[0]
        n = rows * inner
        assert p.shape[0] <= rows and n <= p.numel + p.tail_pad, (p.name, rows, p.tail_pad)
        sl = slice(p.offset, p.offset + n)
        src = self.shadow if self.shadow is not None else self.master
        shape = (rows,) + tuple(p.shape[1:])
        return src[sl].view(shape), self.grad[sl].view(shape)

    def add_transposed(self, param):
        """Registers a transposed bf16 copy of a 2-D parameter; after finalize() `handle.t` is the [cols, rows] view, kept
        in step with the weights by refresh_transposed() (called wherever the bf16 shadow is refreshed)."""
        if self.finalized:
            raise RuntimeError("ParamStore already finalized")
        assert len(param.shape) == 2
        handle = TransposedCopy(param)
        self._transposed.append(handle)
        return handle

    def add_packed(self, kernels, biases):
        """Registers a packed copy of 2-D kernels with equal row counts (and their biases); after finalize() `handle.w` /
        `handle.b` hold them side by side, kept in step with the weights wherever the bf16 shadow is refreshed."""
        if self.finalized:
            raise RuntimeError("ParamStore already finalized")
        assert len({k.shape[0] for k in kernels}) == 1 and len(kernels) == len(biases)
        handle = PackedCopy(kernels, biases)
        self._packed.append(handle)
        return handle

    def finalize(self, device, compute_dtype):
        off = 0
        for p in self.params.values():
            p.offset = off
            off += (p.numel + p.tail_pad + _ALIGN - 1) // _ALIGN * _ALIGN
        self.total = off
        self.device, self.compute_dtype = torch.device(device), compute_dtype
        host = torch.zeros(off, dtype=torch.float32)
        for p in self.params.values():
            host[p.offset:p.offset + p.numel] = p._init.reshape(-1)
            p._init = None
        self.master = host.to(self.device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.shadow = None
        if compute_dtype == torch.bfloat16:
            self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
        for p in self.params.values():
            sl = slice(p.offset, p.offset + p.numel)
            p.data = self.master[sl].view(p.shape)
            p.grad = self.grad[sl].view(p.shape)
            p.compute = (self.shadow if self.shadow is not None else self.master)[sl].view(p.shape)
        self._build_transposed()
        self._build_packed()
        self.finalized = True
        self.refresh_shadow()
        return self

    def _build_transposed(self):
        self.shadow_t = self._tr_table = None
        self._tr_pairs, self._tr_tiles = [], 0
        if self.shadow is None or not self._transposed:
            return
        off = 0
        for hnd in self._transposed:
            hnd.offset = off
            off += (hnd.param.numel + _ALIGN - 1) // _ALIGN * _ALIGN
        self.shadow_t = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
        import numpy as np
        rows = []
        for hnd in self._transposed:
            r, c = hnd.param.shape
            hnd.t = self.shadow_t[hnd.offset:hnd.offset + r * c].view(c, r)
            self._tr_pairs.append((hnd.param.compute, hnd.t))
            tiles_c = (c + 63) // 64
            rows.append((hnd.param.compute.data_ptr(), hnd.t.data_ptr(), r, c, tiles_c, self._tr_tiles))
            self._tr_tiles += ((r + 63) // 64) * tiles_c
        arr = np.array(rows, dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("rows", "<i4"), ("cols", "<i4"),
                                             ("tiles_c", "<i4"), ("tile0", "<i4")]))
        if self.device.type == "cuda":
            self._tr_table = torch.from_numpy(arr.view(np.uint8).copy()).to(self.device)

    def _build_packed(self):
        self._pk_table, self._pk_jobs, self._pk_blocks = None, 0, 0
        if not self._packed:
            return
        import numpy as np
        rows_tab = []
        for hnd in self._packed:
            rows = hnd.kernels[0].shape[0]
            cols = [k.shape[1] for k in hnd.kernels]
            wdt = self.compute_dtype
            hnd.w = torch.zeros(rows, sum(cols), dtype=wdt, device=self.device)
            hnd.b = torch.zeros(sum(cols), dtype=torch.float32, device=self.device)
            hnd.col0, c0 = [], 0
            esz = hnd.w.element_size()
            for k, b in zip(hnd.kernels, hnd.biases):
                hnd.col0.append(c0)
                nb = max(1, min(64, (rows * k.shape[1] * esz // 16 + 255) // 256))
                rows_tab.append((k.compute.data_ptr(), hnd.w.data_ptr() + c0 * esz, rows, k.shape[1] * esz, k.shape[1] * esz,
                                 sum(cols) * esz, self._pk_blocks, nb))
                self._pk_blocks += nb
                if b is not None:
                    rows_tab.append((b.data.data_ptr(), hnd.b.data_ptr() + c0 * 4, 1, k.shape[1] * 4, k.shape[1] * 4, sum(cols) * 4,
                                     self._pk_blocks, 1))
                    self._pk_blocks += 1
                c0 += k.shape[1]
        self._pk_jobs = len(rows_tab)
        arr = np.array(rows_tab, dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("rows", "<i4"), ("row_bytes", "<i4"),
                                                 ("src_pitch", "<i8"), ("dst_pitch", "<i8"), ("block0", "<i4"), ("nblocks", "<i4")]))
        if self.device.type == "cuda":
            self._pk_table = torch.from_numpy(arr.view(np.uint8).copy()).to(self.device)

    def refresh_packed(self):
        """Re-derives the packed copies from the compute weights / fp32 biases: one launch for all of them."""
        if not self._packed:
            return
        if self.master.is_cuda:
            from neurst_amd import kernels
            kernels.pack2d(self._pk_table, self._pk_jobs, self._pk_blocks)
        else:   # host-side tests of the layer scheduling
            for hnd in self._packed:
                for k, b, c0 in zip(hnd.kernels, hnd.biases, hnd.col0):
                    hnd.w[:, c0:c0 + k.shape[1]].copy_(k.compute)
                    if b is not None:
                        hnd.b[c0:c0 + k.shape[1]].copy_(b.data)

    def refresh_transposed(self):
        """Re-derives the transposed bf16 copies from the bf16 shadow: one launch for all of them (and the packed copies)."""
        self.refresh_packed()
        if getattr(self, "shadow_t", None) is None:
            return
        if self.master.is_cuda:
            from neurst_amd import kernels
            kernels.transpose_bf16(self._tr_table, len(self._tr_pairs), self._tr_tiles)
        else:  # host-side tests of the store / layer scheduling (like refresh_shadow's CPU branch)
            for src, dst in self._tr_pairs:
                dst.copy_(src.t())

    def refresh_shadow(self):
        if self.shadow is not None:
            if self.master.is_cuda:
                from neurst_amd import kernels
                kernels.cast_f32_to_bf16(self.master, self.shadow)
            else:  # host-side unit tests of the store itself
                self.shadow.copy_(self.master.to(torch.bfloat16))
        self.refresh_transposed()     # (fp32 too: the packed copies follow the master weights)

    # --- gradient bookkeeping: the first kernel that writes a parameter's gradient in a backward pass
    # overwrites, later writers (tied embedding, gradient accumulation micro-steps) accumulate.
    def begin_backward(self, accumulate=False):
        """One memset of the whole flat gradient buffer per step, then every backward kernel ACCUMULATES: no
        per-tensor zeroing launches (they were ~280 hipMemsetAsync nodes per step)."""
        self._touched.clear()
        if not accumulate:
            self.grad.zero_()
        self.accumulate_all = True

    def acc_flag(self, p):
        if self.accumulate_all or p.name in self._touched:
            return True
        self._touched.add(p.name)
        return False

    def zero_grad(self):
        self.grad.zero_()

    def clip_tables(self, entry_elems=4096):
        """Device tables of nst_grad_clip: one entry per <= 4096 consecutive elements of one parameter, one segment per
        parameter (tf.clip_by_norm clips every gradient tensor by ITS norm)."""
        if getattr(self, "_clip_tables", None) is None:
            import numpy as np
            rows, first = [], []
            for seg, p in enumerate(self.params.values()):
                first.append(len(rows))
                for c in range(0, p.numel, entry_elems):
                    rows.append((p.offset + c, min(entry_elems, p.numel - c), seg))
            first.append(len(rows))
            arr = np.array(rows, dtype=np.dtype([("off", "<i8"), ("n", "<i4"), ("seg", "<i4")]))
            self._clip_tables = (torch.from_numpy(arr.view(np.uint8).copy()).to(self.device), len(rows),
                                 torch.tensor(first, dtype=torch.int32, device=self.device), len(first) - 1)
        return self._clip_tables

    def state_dict(self):
        return {n: p.data.detach().cpu().clone() for n, p in self.params.items()}

    def load_state_dict(self, sd, strict=True):
        missing = [n for n in self.params if n not in sd]
        if strict and missing:
            raise KeyError(f"missing variables: {missing[:5]}...")
        for n, p in self.params.items():
            if n in sd:
                p.data.copy_(torch.as_tensor(sd[n], dtype=torch.float32).reshape(p.shape))
        self.refresh_shadow()

    def num_parameters(self):
        return sum(p.numel for p in self.params.values())


_HWQ_NOTE = [None, None]     # (the line logged by the first call, the value of the variable behind it)


# every NST_* variable something in this tree reads (the library, the host side, bench.py, the tests' kernel-selection switches)
KNOWN_SWITCHES = frozenset((
    "NST_TRAIN_GRAPH", "NST_DIST_BACKEND", "NST_DIST_FORCE", "NST_DIST_WIRE", "NST_DIST_NATIVE", "NST_DIST_TIMEOUT_MIN",
    "NST_DIST_BUCKET_MB", "NST_DIST_DEBUG", "NST_RCCL_PATH", "NST_COMM_DEBUG", "NST_BENCH_CHILD", "NST_BENCH_HANG_DUMP_S",
    "NST_BENCH_VERBOSE", "NST_FFN_MIN_ROWS", "NST_ATTN_FUSED_BWD", "NST_ATTN_MI_FWD", "NST_ATTN_MI_DKDV", "NST_ATTN_MI_DQ",
    "NST_ROW_FUSION", "NST_ROWGEMM_CFG", "NST_TEST_L2_F32", "NST_TEST_L2_BF16", "NST_LIBRARY", "NST_FFN_DBG"))


def warn_unknown_switches():
    """A dead experiment switch must not silently mislabel a measurement (round 5: a profile script still set NST_WGRAD_STREAM=0
    after the variable had been removed): every NST_* variable of the environment that nothing reads any more is reported."""
    unknown = sorted(k for k in os.environ if k.startswith("NST_") and k not in KNOWN_SWITCHES)
    if unknown:
        import warnings
        warnings.warn("environment variables that neurst_amd does not read (removed switches?): " + ", ".join(unknown))
    return unknown


def configure_training_process():
    """Process-wide HIP setting of the TRAINING entry points (init_distributed, bench.py, neurst-run); returns the line it logs.
    Must run before the first HIP call of the process (the runtime reads the variable once, when it initialises).

    GPU_MAX_HW_QUEUES = 1.  ROCm multiplexes the HIP streams of ONE priority class onto at most this many hardware queues
    (default 4).  A training step here has three concurrent activities and keeps each in a priority class of its own
    (make_stream: the step on a high-priority stream, its weight-gradient stream on a low-priority one, the gradient exchange
    in the default class), so one queue per class is all the concurrency it needs -- and MORE queues are what hurt: with 16 per
    class the same step ran at 13.0 ms or at 21-31 ms depending only on how many streams other libraries had touched before
    the first step (a second RCCL communicator, a few idle pool streams); with 1 or 2 queues per class all 14 configurations
    tried run at 12.96-13.22 ms (profiles/r04_history/c26_order.log, c27_streams.log, c28_hwq.log).  Streams that share a
    queue execute in submission order; with one submitting host thread and record-before-wait events that order cannot close
    a wait cycle.  An explicit setting in the environment wins.  The setting relies on the three streams landing in three
    DISTINCT classes: make_stream warns when a class is missing and the stream falls back to the default one.
    Importing neurst_amd alone (decoding, other libraries in the process) does not apply it."""
    import logging
    import torch
    cur = os.environ.get("GPU_MAX_HW_QUEUES")
    if _HWQ_NOTE[0] is None:
        warn_unknown_switches()
    if _HWQ_NOTE[0] is not None and _HWQ_NOTE[1] == cur:
        return _HWQ_NOTE[0]          # a repeated call (bench.py asks again for its JSON line): what the FIRST call did
    if cur is not None:
        note = f"GPU_MAX_HW_QUEUES={cur} (taken from the environment)"
    elif torch.cuda.is_initialized():
        note = "GPU_MAX_HW_QUEUES not set and HIP is already initialised: the runtime default (4 queues per class) stays"
    else:
        os.environ["GPU_MAX_HW_QUEUES"] = "1"
        note = "GPU_MAX_HW_QUEUES=1 (training default: one hardware queue per stream-priority class)"
    logging.getLogger("neurst_amd").info(note)
    _HWQ_NOTE[0], _HWQ_NOTE[1] = note, os.environ.get("GPU_MAX_HW_QUEUES")
    return note


def make_stream(device, priority_class):
    """A HIP stream of priority class -1 (high) / 0 (default) / 1 (low) as a torch stream.  Why classes: the HIP runtime
    multiplexes the streams of ONE class onto a few hardware queues (the least used one at creation time), and two busy streams
    that meet in a queue run in turns -- the same step took 13.0 or 22 ms depending on how many streams other libraries had
    created before (profiles/r04_history/c26_order.log).  The step (high), its weight-gradient stream (low) and the gradient
    exchange (default) are therefore kept in different classes.  torch.cuda.Stream reaches the default class and the higher
    ones; the low class comes from the library (nst_stream_create) and is wrapped as an ExternalStream."""
    device = torch.device(device)
    if priority_class <= 0:
        return torch.cuda.Stream(device, priority=priority_class)
    import ctypes
    from neurst_amd import _lib
    handle = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = _lib.lib.nst_stream_create(priority_class, ctypes.byref(handle))
    if rc != 0 or not handle.value:      # no such class on this device: the default class
        import warnings
        warnings.warn(f"no stream priority class {priority_class} on {device} (nst_stream_create rc={rc}): the stream shares the "
                      "default class" + (" -- with GPU_MAX_HW_QUEUES=1 it then shares ONE hardware queue with the gradient "
                                         "exchange and executes in submission order (no overlap)"
                                         if os.environ.get("GPU_MAX_HW_QUEUES") == "1" else ""))
        return torch.cuda.Stream(device)
    return torch.cuda.ExternalStream(handle.value, device=device)   # lives as long as the process


class Runtime(object):
    """Per-process execution context shared by all layers of a model."""

    def __init__(self, device="cuda:0", dtype="float32", seed=1234):
        self.device = torch.device(device)
        if isinstance(dtype, str):
            dtype = {"float32": torch.float32, "fp32": torch.float32, "bfloat16": torch.bfloat16,
                     "bf16": torch.bfloat16}[dtype]
        self.dtype = dtype
        self.store = ParamStore()
        self.base_seed = int(seed)
        self._step = 0
        # device_step: the step counter that varies the dropout masks lives in the library's device scalar (read by the
        # kernels when they run) instead of in the seed argument -- what a captured HIP graph of the step needs
        self.device_step = False
        # this runtime's own seed-offset scalar: bound before its kernels are launched (bind()), so several models in one
        # process -- and captured graphs -- never see each other's counters
        self._step_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._sites = 0
        self._posenc_cache = {}
        # Weight-gradient GEMMs do not feed the backward chain: they run on a second HIP stream so that their
        # workgroups fill the ramp-up / tail gaps of the dgrad-path kernels.
        self.wgrad_stream = None
        self.capture = None   # set by TrainStep while it captures a step (see run_wgrad)
        if self.device.type == "cuda":
            self.wgrad_stream = make_stream(self.device, 1)      # low-priority class, see make_stream

    @contextlib.contextmanager
    def on_wgrad_stream(self, *tensors):
        """Runs the body on the weight-gradient stream, ordered after everything queued so far on the current
        stream; `tensors` are the operands it reads (kept alive for that stream by the caching allocator)."""
        s = self.wgrad_stream
        if s is None:
            yield
            return
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            yield
        for t in tensors:
            t.record_stream(s)

    def run_wgrad(self, fn, *tensors):
        """Runs fn() -- weight-gradient launches that do not feed the backward chain -- on the weight-gradient stream, after
        everything queued so far on the current stream; `tensors` are the operands it reads.  While a train step is being
        CAPTURED (training/train_step.py) the call is only recorded: the capture replays the recorded calls of a layer into
        a graph of their own, which the replay launches on the weight-gradient stream next to the following layers."""
        cap = self.capture
        if cap is not None:
            cap.defer(fn, tensors)
            return
        with self.on_wgrad_stream(*tensors):
            fn()

    def wgrad_batch(self):
        """The split-K batch of the weight-gradient stream (kernels.SplitkBatch), created on first use; None on the CPU tier
        ."""
        if self.device.type != "cuda":
            return None
        if getattr(self, "_wgrad_batch", None) is None:
            from neurst_amd import kernels
            self._wgrad_batch = kernels.SplitkBatch(self.device)
        return self._wgrad_batch

    def wgrad_group(self):
        """The pending weight-gradient group (kernels.WgradGroup) or None (a runtime without a device)."""
        if self.device.type != "cuda" and not getattr(self, "_wgrad_group_on_cpu", False):
            return None
        if getattr(self, "_wgrad_group", None) is None:
            from neurst_amd import kernels
            self._wgrad_group = kernels.WgradGroup(self.device)
            self._deferred_reports = []
        return self._wgrad_group

    def report_or_defer(self, report):
        """`report()` tells the data-parallel reducer that everything writing some gradients has been queued.  While weight
        gradients wait in the group that is not true yet: the report then runs right behind the group's launch."""
        g = getattr(self, "_wgrad_group", None)
        if g is not None and len(g):
            self._deferred_reports.append(report)
        else:
            report()

    def drop_pending_wgrads(self):
        """Forgets weight gradients that were queued for the grouped launch, and the reducer reports waiting behind it, without
        launching them; returns how many there were.  Called at the start of every backward pass and when one aborts (an
        exception between Dense.backward_params and launch_wgrad_group, a failed capture): a dead batch's products must never
        run inside the next step's launch (they would overwrite / accumulate into its gradients with stale accumulate flags),
        and a stale report would make this rank issue one collective more than its peers."""
        g = getattr(self, "_wgrad_group", None)
        stale = 0
        if g is not None and len(g):
            stale += len(g)
            g.items = []
        if getattr(self, "_deferred_reports", None):
            stale += len(self._deferred_reports)
            self._deferred_reports = []
        return stale

    def launch_wgrad_group(self, side=False):
        """One launch for every weight gradient queued since the last call -- on the current stream, or (side=True) on the
        weight-gradient stream behind everything queued so far on the current one -- then the reports that waited for it (the
        reducer orders its exchange behind both streams).  (Groups that cannot fill the chip on the weight-gradient stream, and one launch per stack, were measured
        and lost: DESIGN 5e.)"""
        g = getattr(self, "_wgrad_group", None)
        if g is None:
            return
        if len(g):
            if side:
                fn, tensors = g.take()
                self.run_wgrad(fn, *tensors)
                self.sublayer_boundary(force=True)
            else:
                g.launch()
        reports, self._deferred_reports = self._deferred_reports, []
        for r in reports:
            r()

    def flush_wgrads(self):
        """Launches the pending (deferred) second stages of the weight gradients queued so far, on their stream."""
        b = getattr(self, "_wgrad_batch", None)
        if b is not None and (b.n or b.ln_n or self.capture is not None):
            self.run_wgrad(b.flush)

    def wgrad_boundary(self):
        """End of a layer's backward: every launch that writes its gradients is queued (eager) / belongs to the segment being
        closed (capture)."""
        self.flush_wgrads()
        if self.capture is not None:
            self.capture.layer_boundary(report=True)

    def sublayer_boundary(self, force=False):
        """Capture only: the weight-gradient calls recorded so far may start now (their graph is launched on the
        weight-gradient stream next to what follows).  Eager launches start on their own.  force: cut even when only a
        few calls are pending (a LONG weight gradient in front of long compute-stream kernels: without the cut it
        starts when the whole segment has been replayed -- the conv2 weight gradient then ran alone at the end of every
        step, 1.1 ms behind the last compute-stream kernel)."""
        if self.capture is not None:
            self.capture.layer_boundary(force=force)

    def join_wgrad_stream(self):
        """The current stream waits for every weight gradient queued so far."""
        self.launch_wgrad_group()
        self.flush_wgrads()
        b = getattr(self, "_wgrad_batch", None)
        if b is not None:
            b.joined()
        if self.capture is not None:
            self.capture.join()
            return
        if self.wgrad_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.wgrad_stream)

    def new_dropout_site(self):
        self._sites += 1
        return self._sites

    @property
    def step(self):
        return self._step

    @step.setter
    def step(self, value):
        self._step = int(value)
        if self.device_step:
            from neurst_amd import kernels
            with self.bound():
                kernels.dropout_seed_offset_set(self._step)

    @contextlib.contextmanager
    def bound(self):
        """Inside: this runtime's seed-offset scalar is the one kernel launches read (the pointer travels with each launch,
        so it also ends up in captured graphs).  Outside nothing stays bound: direct kernel calls see the library's own zero
        scalar, and no binding outlives the runtime's memory.  Re-entrant."""
        from neurst_amd import kernels
        depth = getattr(self, "_bind_depth", 0)
        if depth == 0:
            kernels.dropout_seed_offset_bind(self._step_dev)
        self._bind_depth = depth + 1
        try:
            yield
        finally:
            self._bind_depth -= 1
            if self._bind_depth == 0:
                kernels.dropout_seed_offset_bind(None)

    def enable_device_step(self):
        if not self.device_step:
            self.device_step = True
            self.step = self._step   # uploads the counter

    def advance_step(self, enqueue=True):
        """End of an optimizer step.  enqueue=False: the device-side increment is part of a graph that was just replayed."""
        self._step += 1
        if self.device_step and enqueue:
            from neurst_amd import kernels
            with self.bound():
                kernels.dropout_seed_offset_add(1)

    @property
    def step_seed(self):
        """The seed ARGUMENT of this step's dropout kernels: with device_step the kernels add the step themselves."""
        return self.base_seed * 1000003 + (0 if self.device_step else self._step)

    @property
    def effective_seed(self):
        """The Philox key the kernels end up using in this step, whichever side adds the step."""
        return self.base_seed * 1000003 + self._step

    def posenc(self, length, channels):
        """Sinusoid timing table [length, channels] f32 (neurst/layers/common_layers.py:356-413), built on
        the host exactly like the reference and cached on the device."""
        key = (length, channels)
        if key not in self._posenc_cache:
            position = torch.arange(0, length, dtype=torch.float32)
            nts = channels // 2
            inc = math.log(1.0e4 / 1.0) / (float(nts) - 1)
            inv = torch.exp(torch.arange(nts, dtype=torch.float32) * -inc)
            scaled = position[:, None] * inv[None, :]
            sig = torch.cat([torch.sin(scaled), torch.cos(scaled)], dim=1)
            if channels % 2:
                sig = torch.nn.functional.pad(sig, (0, 1))
            self._posenc_cache[key] = sig.contiguous().to(self.device)
        return self._posenc_cache[key]
