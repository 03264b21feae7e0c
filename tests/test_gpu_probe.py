"""Measurement-only entry points of the C ABI (include/neurst_hip.h "probes"): they must launch, return and refuse bad arguments.
The lane-map probes of the MFMA / transpose-read instructions are checked in tests/test_gpu_kernels.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu
def test_fetch_probe_runs_and_moves_what_it_says():
    """nst_probe_fetch is measurement code (scripts/fetch_probe.py); here only: every mode / lane map launches, returns, and leaves
    the sink untouched; bad arguments are refused."""
    import ctypes as C
    from neurst_amd._lib import lib
    buf = torch.zeros(8 << 20, dtype=torch.uint8, device="cuda:0")
    sink = torch.zeros(4, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for mode, pat in ((0, 0), (0, 1), (0, 2), (0, 3), (0, 4), (1, 0)):
        for ring in (2, 4):
            assert lib.nst_probe_fetch(buf.data_ptr(), 0, 1 << 20, 16, ring, mode, 64, 64, pat, sink.data_ptr(), st) == 0
            assert lib.nst_probe_fetch(buf.data_ptr(), 1 << 20, 1 << 20, 16, ring, mode, 8, 8, pat, sink.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert float(sink.abs().sum()) == 0.0
    assert lib.nst_probe_fetch(buf.data_ptr(), 0, 1000, 16, 2, 0, 64, 64, 0, sink.data_ptr(), st) != 0     # span not a multiple of 32 KB
    assert lib.nst_probe_fetch(buf.data_ptr(), 0, 1 << 20, 16, 5, 0, 64, 64, 0, sink.data_ptr(), st) != 0   # ring depth 2..4
