"""MI355X-native SpeechTransformer training path behind NeurST's registries.

Process-wide HIP setting, applied before the HIP runtime initialises (it reads the variable once, at its first call):

GPU_MAX_HW_QUEUES = 1.  ROCm multiplexes the HIP streams of ONE priority class onto at most this many hardware queues (default
4).  A training step here has three concurrent activities and keeps each in a priority class of its own (runtime.make_stream:
the step on a high-priority stream, its weight-gradient stream on a low-priority one, the gradient exchange in the default
class), so one queue per class is all the concurrency it needs -- and MORE queues are what hurts: with 16 per class (rounds 2-3,
chosen when the step's streams still shared a class) the same step ran at 13.0 ms or at 21-31 ms depending only on how many
streams other libraries had touched before the first step (a second RCCL communicator, a few idle pool streams; even a
single-stream eager step: 31 ms).  In the slow runs every kernel is stretched by 30-45 us (rocprofv3 kernel trace), as if the
scheduler time-sliced the process's queues once their number passes a threshold.  With 1 or 2 queues per class all 14
configurations tried run at 12.96-13.22 ms (profiles/r04_history/c26_order.log, c27_streams.log, c28_hwq.log).
Streams that share a queue execute in submission order; with one submitting host thread and record-before-wait events that
order cannot close a wait cycle (round 2's deadlock needed gloo's worker threads submitting copies of their own; that
rehearsal path is host-staged since round 3).  An explicit setting in the environment wins.
"""
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")
