"""One optimizer step of the hot loop (GradAccumKerasModel.train_step / fit loop,
neurst/training/gradaccum_keras_model.py:162-260, 437-477; hvd optimizer hook hvd_utils.py:85-91):

    forward -> label-smoothed CE -> backward (bucketed RCCL all-reduce overlapped on a side stream)
            -> [update_cycle-1 more micro batches accumulated] -> average [-> clip by value / per-tensor norm] -> fused Adam

No host<->device synchronisation happens inside a step; the loss is returned as a device scalar.

Graph mode (use_graph=True / NST_TRAIN_GRAPH=1; the reference's counterpart is tf.function tracing of the train step,
gradaccum_keras_model.py:262-352): the step issues ~450 library calls, ~10 ms of Python / ctypes work against 13-18 ms of
GPU time, so per input signature (shapes + dtypes of the batch) the whole step is captured ONCE into HIP graphs over static
input buffers and replayed afterwards.  What changes per step is read from device memory when the kernels run: the
dropout step counter (nst_dropout_seed_offset_*: `add 1` is the last node of the graph) and Adam's step size lr_t (a
device scalar the host refreshes before each replay).  With more than one rank the capture is CUT wherever the reducer
issues a bucket: the all-reduce stays an eager RCCL call on the communication stream between two graph launches, so the
exchange still overlaps the rest of the backward pass (no collective is captured).  The first call with a new signature
runs eagerly (allocations, lazy tables), the second captures and replays, later ones only replay.
"""
import os
import warnings

import torch

from neurst_amd import _lib


class _Segment(object):
    """One slice of a captured step: a graph of compute-stream work, optionally followed by a graph of the weight-gradient
    launches recorded while it was captured (replayed on the weight-gradient stream), a join of that stream, and a
    gradient bucket to exchange."""
    __slots__ = ("main", "wgrad", "join", "buckets")

    def __init__(self):
        self.main = self.wgrad = None
        self.join, self.buckets = False, []


class _CapturedStep(object):
    __slots__ = ("static_inputs", "segments", "loss", "pool", "keep")


class _StepCapture(object):
    """The object Runtime.capture points at while a step is captured: collects the deferred weight-gradient calls and cuts
    the capture into segments (see TrainStep._capture)."""

    def __init__(self, rt, pool, main_stream, side_stream):
        self.rt, self.pool, self.main_stream, self.side_stream = rt, pool, main_stream, side_stream
        self.segments, self.deferred, self.keep = [], [], []
        self.cur = None
        self._fresh = False      # nothing has been queued since the current segment was opened by a layer boundary
        self.min_deferred = 6        # weight-gradient calls that make a layer boundary cut the capture (measured in DESIGN 5b)
        self.force_cuts = True       # Runtime.sublayer_boundary(force=True) cuts whatever is pending

    def begin(self):
        self.cur = _Segment()
        self.cur.main = torch.cuda.CUDAGraph()
        # thread-local capture mode: with a process group alive, RCCL's watchdog thread polls events while this thread
        # captures -- in the default (global) mode such a call from ANOTHER thread invalidates the capture and aborts the
        # process (seen once in five forced-exchange runs of bench.py, round 3)
        self.cur.main.capture_begin(pool=self.pool, capture_error_mode="thread_local")
        self._calls_at_begin = _lib.CALLS[0]

    def _close(self):
        """Ends the compute graph of the current segment and captures its weight-gradient graph from the recorded calls."""
        seg = self.cur
        # a segment that only carries deferred weight-gradient calls (a join or a bucket cut right behind another cut) has
        # no compute-stream node: torch warns "The CUDA Graph is empty" -- such a graph is dropped instead of replayed
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            seg.main.capture_end()
        empty = any("Graph is empty" in str(w.message) for w in caught)
        queued = _lib.CALLS[0] - self._calls_at_begin     # library calls issued on this thread since begin()
        if empty and queued == 0:
            seg.main = None       # our own bookkeeping agrees: nothing was queued between two cuts
        else:
            for w in caught:
                warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
            if empty:             # kernels were queued but none was captured: wrong stream or device -- never skip that silently
                raise RuntimeError(f"captured segment is empty although {queued} library calls were issued while it was "
                                   "open: they ran on another stream or device than the one being captured")
        if self.deferred:
            calls, self.deferred = self.deferred, []
            self.side_stream.wait_stream(self.main_stream)
            with torch.cuda.stream(self.side_stream):
                cap, self.rt.capture = self.rt.capture, None      # the calls run for real inside this capture
                try:
                    seg.wgrad = torch.cuda.CUDAGraph()
                    seg.wgrad.capture_begin(pool=self.pool, capture_error_mode="thread_local")
                    calls0 = _lib.CALLS[0]
                    for fn in calls:
                        fn()
                    # recorded calls that turned out to have nothing to launch (a flush of an empty batch): same rule as above
                    with warnings.catch_warnings(record=True) as caught_w:
                        warnings.simplefilter("always")
                        seg.wgrad.capture_end()
                    empty_w = any("Graph is empty" in str(w.message) for w in caught_w)
                    if empty_w and _lib.CALLS[0] == calls0:
                        seg.wgrad = None
                    else:
                        for w in caught_w:
                            warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
                        if empty_w:
                            raise RuntimeError(f"weight-gradient graph is empty although {_lib.CALLS[0] - calls0} library calls "
                                               "were issued for it: they ran on another stream than the one being captured")
                finally:
                    self.rt.capture = cap
            self.main_stream.wait_stream(self.side_stream)
        self.segments.append(seg)
        self.cur = None
        return seg

    # --- called through Runtime while the step body runs
    def defer(self, fn, tensors):
        self._fresh = False
        self.deferred.append(fn)
        self.keep.extend(tensors)      # their memory must not be handed out again inside this capture (the graphs run concurrently)

    def layer_boundary(self, report=False, force=False):
        """report=True: the boundary of a gradient REPORT (Runtime.wgrad_boundary: everything that writes the layer's
        gradients is queued, the reducer's hook fires next).  Only then may the bucket that follows ride on the segment
        closed here (`_fresh`); a mid-layer boundary (sublayer_boundary) is followed by more compute-stream work of the
        same layer (LayerNorm backward, bias gradients), so a bucket reported after it must cut the capture again."""
        # a cut costs two graph launches (~10-20 us of idle compute stream each); waiting for a few weight-gradient calls
        # trades that against a later start of the weight-gradient graph (min_deferred, measured in DESIGN 5b)
        self._fresh = False
        if len(self.deferred) >= self.min_deferred or (force and self.deferred and self.force_cuts):
            self._close()
            self.begin()
            self._fresh = bool(report)

    def join(self):
        seg = self._close()
        seg.join = True
        self.begin()
        self._fresh = False

    def cut_for_bucket(self, start, stop):
        # the model reports a layer right after its boundary: the bucket then belongs to the segment just closed (every writer
        # of these gradients is in it or before it) and no empty segment is opened for it
        seg = self.segments[-1] if (self._fresh and self.segments) else None
        if seg is None:
            seg = self._close()
            self.begin()
            self._fresh = True
        seg.buckets.append((start, stop))

    def finish(self):
        self._close()


class TrainStep(object):
    def __init__(self, model, criterion, optimizer, reducer=None, update_cycle=1, clip_value=None, clip_norm=None,
                 use_graph=None, loss_scale=None):
        self.model, self.criterion, self.optimizer, self.reducer = model, criterion, optimizer, reducer
        # gradaccum_keras_model.py:228-233: clip_value takes precedence over clip_norm; both act on the averaged gradients
        self.clip_value = clip_value if clip_value else None
        self.clip_norm = None if self.clip_value else (clip_norm if clip_norm else None)
        self.update_cycle = max(1, int(update_cycle))
        if reducer is not None:
            model.grad_ready_hook = self._hook
            ws = getattr(model.rt, "wgrad_stream", None)
            if ws is not None and ws not in reducer.extra_streams:
                reducer.extra_streams.append(ws)
        self._last_micro = True
        # dynamic loss scale (training_utils.py:373-419 wraps the optimizer with it for float16; RevisedDynamicLossScale:
        # initial 2^15, x2 every 2000 good steps, /2 on overflow with the step skipped).  bf16 keeps fp32's exponent range and
        # does not need it; it is available for every dtype: loss_scale="dynamic" or a dict of its three constants.
        self.loss_scale = None
        if loss_scale:
            import inspect
            if "loss_scale_dev" not in inspect.signature(criterion.backward).parameters:
                raise TypeError(f"{type(criterion).__name__}.backward() must accept loss_scale_dev for dynamic loss scaling "
                                "(see Criterion.backward)")
            cfg = dict(initial_loss_scale=2.0 ** 15, growth_steps=2000, multiplier=2.0)
            if isinstance(loss_scale, dict):
                cfg.update(loss_scale)
            dev = model.rt.device
            self.loss_scale = cfg
            self._ls_state = torch.tensor([cfg["initial_loss_scale"], 0.0, 1.0, cfg["initial_loss_scale"]], dtype=torch.float32).to(dev)
            self._ls_counter = torch.zeros(4, dtype=torch.int32, device=dev)
            # together with clipping (gradaccum_keras_model.py:224-233: aggregate -> get_unscaled_gradients -> clip -> apply):
            # the factor that un-scales and averages, and the state the optimizer sees afterwards (same finite flag, scale 1)
            self._ls_factor = torch.zeros(1, dtype=torch.float32, device=dev)
            self._ls_apply = self._ls_state.clone()
        if use_graph is None:
            use_graph = os.environ.get("NST_TRAIN_GRAPH", "0") == "1"
        self.use_graph = bool(use_graph) and model.rt.device.type == "cuda"
        self._seen, self._captured = set(), {}
        self._lr_dev = self._cap_stream = self._side_stream = None
        self.replays = 0
        # The step runs on a HIGH-priority stream of its own: the runtime then
        # keeps it on hardware queues apart from the weight-gradient stream (low class) and from the exchange (default class),
        # whatever streams other libraries created before -- see runtime.make_stream.  The caller's stream is ordered before
        # and after the step, so the step still behaves like work queued on the caller's stream.
        self._step_stream = None
        if model.rt.device.type == "cuda":
            from neurst_amd.runtime import make_stream
            self._step_stream = make_stream(model.rt.device, -1)
        if self.use_graph:
            model.rt.enable_device_step()
            self._lr_dev = torch.zeros(1, dtype=torch.float32, device=model.rt.device)
            self._cap_stream = self._step_stream if self._step_stream is not None else torch.cuda.Stream(model.rt.device)

    def _hook(self, prefixes):
        if self._last_micro and self.reducer is not None:
            self.reducer.component_ready(prefixes)

    # ------------------------------------------------------------------------------------------------ the step body
    def _body(self, batches, lr_t_dev=None):
        """Queues one optimizer step on the current stream (eagerly, or into an ongoing capture)."""
        n = len(batches)
        loss_sum = None
        for i, inputs in enumerate(batches):
            self._last_micro = (i == n - 1)
            logits = self.model(inputs, is_training=True)
            loss = self.criterion.reduce_loss(inputs, logits)
            if self.loss_scale:
                dlogits = self.criterion.backward(loss_scale=1.0 / n, loss_scale_dev=self._ls_state[0:1])
            else:
                dlogits = self.criterion.backward(loss_scale=1.0 / n)
            del logits
            self.model.backward(dlogits, accumulate=(i > 0))
            loss_sum = loss if loss_sum is None else loss_sum + loss
        scale = self.reducer.finish() if self.reducer is not None else 1.0
        from neurst_amd import kernels as K
        clip = bool(self.clip_value or self.clip_norm)
        ls_state = None
        if self.loss_scale:
            # the finite check runs on the exchanged, still SCALED gradients (the reference aggregates before it unscales)
            K.loss_scale_update(self.model.store.grad, self._ls_state, self.loss_scale["growth_steps"], self.loss_scale["multiplier"],
                                self._ls_counter)
            ls_state = self._ls_state
            if clip:
                # clipping acts on UNSCALED gradients: divide by the scale they carry (state[3], device-resident: the step may
                # be a graph replay) and average in one pass; the optimizer then only needs the finite flag.  An overflow step
                # clips inf / nan into garbage that the flag keeps out of the weights.
                torch.div(torch.full_like(self._ls_factor, float(scale)), self._ls_state[3:4], out=self._ls_factor)
                self.model.store.grad.mul_(self._ls_factor)
                scale = 1.0
                self._ls_apply.copy_(self._ls_state)
                self._ls_apply[3:4].fill_(1.0)
                ls_state = self._ls_apply
        if clip:
            table, nentries, seg_first, nseg = self.model.store.clip_tables()
            K.grad_clip(self.model.store.grad, table, nentries, seg_first, nseg, pre_scale=scale,
                        clip_value=self.clip_value, clip_norm=self.clip_norm)
            scale = 1.0   # the average is already applied
        if ls_state is not None:
            self.optimizer.apply_gradients(grad_scale=scale, lr_t_dev=lr_t_dev, loss_scale_state=ls_state)
        else:
            self.optimizer.apply_gradients(grad_scale=scale, lr_t_dev=lr_t_dev)
        return loss_sum / n

    @property
    def stream(self):
        """The high-priority stream the step runs on (None: whatever stream is current when the step is called).  A loop whose
        batches are already resident may make it the CURRENT stream: a call from another stream hands over through two event
        waits (caller -> step stream -> caller), ~30 us of idle GPU per step in the graph-replayed benchmark loop."""
        return self._step_stream

    def __call__(self, batches):
        """batches: one model-input dict, or a list of `update_cycle` dicts (gradient accumulation: the mean of
        the micro-batch gradients, GradientAccumulator semantics gradaccum_keras_model.py:62-109)."""
        if isinstance(batches, dict):
            batches = [batches]
        if self._step_stream is None:
            return self._call(batches)
        cur = torch.cuda.current_stream(self.model.rt.device)
        if cur == self._step_stream:
            return self._call(batches)
        self._step_stream.wait_stream(cur)
        with torch.cuda.stream(self._step_stream):
            loss = self._call(batches)
        cur.wait_stream(self._step_stream)
        return loss

    def _call(self, batches):
        if self.use_graph:
            return self._graph_call(batches)
        loss = self._body(batches)
        self.model.rt.advance_step()
        return loss

    # ------------------------------------------------------------------------------------------------ graph mode
    @staticmethod
    def _signature(batches):
        return tuple(tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(b.items()) if torch.is_tensor(v))
                     for b in batches)

    def _graph_call(self, batches):
        key = self._signature(batches)
        cap = self._captured.get(key)
        if cap is None:
            if key not in self._seen:          # first sight of this shape: a plain eager step (also the warm-up)
                self._seen.add(key)
                loss = self._body(batches)
                self.model.rt.advance_step()
                return loss
            cap = self._captured[key] = self._capture(batches)
        for sb, b in zip(cap.static_inputs, batches):
            for k, v in sb.items():
                if b[k] is not v and b[k].data_ptr() != v.data_ptr():
                    v.copy_(b[k], non_blocking=True)
        self._lr_dev.fill_(self.optimizer.step_size())
        red, rt = self.reducer, self.model.rt
        cur = torch.cuda.current_stream(rt.device)
        nb = sum(len(sg.buckets) for sg in cap.segments)
        last = len(cap.segments) - 1
        msgs0 = red.messages if red is not None else 0
        for i, sg in enumerate(cap.segments):
            if i == last and nb:
                red.wait_issued()              # every bucket has been exchanged before clip / Adam
            if sg.main is not None:
                sg.main.replay()
            if sg.wgrad is not None:           # the layer's weight gradients, next to the following layers' backward
                rt.wgrad_stream.wait_stream(cur)
                with torch.cuda.stream(rt.wgrad_stream):
                    sg.wgrad.replay()
            if sg.join:
                cur.wait_stream(rt.wgrad_stream)
            for bucket in sg.buckets:
                red.issue(*bucket)
        if red is not None:     # collectives of THIS replay (a bucket is cut into <= bucket_bytes messages); finish() is not
            red.last_messages, red.messages = red.messages - msgs0, 0   # called on replays, so the counter is reset here
        self.optimizer.advance()
        rt.advance_step(enqueue=False)   # the increment of the device counter is the graph's last node
        self.replays += 1
        return cap.loss

    def _capture(self, batches):
        """Captures one step as a list of segments.  Compute-stream work goes into `main` graphs; the weight-gradient calls
        (Runtime.run_wgrad) are only recorded while a segment is open and, at the next layer boundary, captured into a
        graph of their own that the replay launches on the weight-gradient stream -- two plain graph launches per layer
        instead of a fork / join inside one graph, which the ROCm 7.2 graph executor serialises (19.9 vs 17.9 ms / step)."""
        from neurst_amd import kernels as K
        rt, red = self.model.rt, self.reducer
        if rt.wgrad_stream is None:
            raise RuntimeError("graph mode needs the weight-gradient stream (a Runtime on a ROCm device)")
        cap = _CapturedStep()
        cap.static_inputs = [{k: v.clone() for k, v in b.items() if torch.is_tensor(v)} for b in batches]
        full = [dict(b, **sb) for b, sb in zip(batches, cap.static_inputs)]
        cap.pool = torch.cuda.graph_pool_handle()
        stream = self._cap_stream
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(rt.device)
        stream.wait_stream(torch.cuda.current_stream(rt.device))
        sc = _StepCapture(rt, cap.pool, stream, self._side_stream)
        if red is not None:
            red.capture_cut = sc.cut_for_bucket
        # No destructor of a device object may run while the step is being captured: garbage left by earlier work (an older
        # TrainStep's graphs, events, streams in reference cycles) is collected NOW, on an idle device, and the cyclic collector
        # stays off until the capture has ended -- what torch.cuda.graph() does on entry; a collection that freed a captured
        # graph of an earlier model in the middle of this capture aborted the process (seen when tests/test_gpu_graph.py ran
        # behind tests/test_gpu_model.py in one interpreter).
        import gc
        torch.cuda.synchronize(rt.device)
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        rt.capture = sc
        try:
            with torch.cuda.stream(stream):
                sc.begin()
                loss = self._body(full, lr_t_dev=self._lr_dev)
                with rt.bound():
                    K.dropout_seed_offset_add(1)
                cap.loss = loss
                sc.finish()
        finally:
            if gc_was_on:
                gc.enable()
            rt.capture = None
            if red is not None:
                red.capture_cut = None
            if sc.cur is not None:   # an exception inside the capture: leave capture mode
                try:
                    sc.cur.main.capture_end()
                except Exception:
                    pass
                rt.drop_pending_wgrads()   # nothing of the aborted step may run inside the next one's grouped launch
        torch.cuda.current_stream(rt.device).wait_stream(stream)
        cap.segments, cap.keep = sc.segments, sc.keep
        return cap
