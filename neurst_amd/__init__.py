"""MI355X-native SpeechTransformer training path behind NeurST's registries.

Process-wide HIP setting, applied before the HIP runtime initialises (it reads the variable once, at its first call):

GPU_MAX_HW_QUEUES.  ROCm multiplexes a process's HIP streams onto at most this many hardware queues per device
(default 4); streams that share a queue run strictly one after the other.  A data-parallel rank owns the compute stream,
the weight-gradient stream, the communication stream of the gradient exchange and RCCL's own stream(s): with 4 queues the
weight-gradient stream lands on the compute stream's queue and the two stop overlapping -- measured on one MI355X with
the exchange path active: 20.2 ms per step against 16.4 ms with 8 queues (the kernels then run back to back: kernel-time
sum == busy time in the rocprofv3 trace).  Worse, packets of two streams in one queue execute in SUBMISSION order, so a
cross-stream wait can become a cycle: in a two-rank rehearsal over gloo (whose CUDA path takes a fresh pool stream per
collective, 13 per step) the second step deadlocked with 8 queues -- a copy waiting for the compute stream sat in front of
the weight-gradient GEMMs the compute stream was waiting for -- and ran with 32.  16 leaves headroom over the 4-6 streams
of an RCCL rank and measured the same step time as 8.  An explicit setting in the environment wins.
"""
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
