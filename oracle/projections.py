"""Fixed random +-1 projections of large tensors (test infrastructure only: tests/ and the fixture generators under
tests/golden/ import it; the product never does).

A gradient set of 29 M parameters cannot travel as a fixture; 64 Rademacher projections per tensor can (KBs), and they keep
what a parity check needs: E |R (g - r)|^2 = 64 |g - r|^2, so the relative L2 distance of two projection sets estimates the
relative L2 distance of the tensors themselves (standard error ~ 1 / sqrt(64 * tensors))."""
import zlib

import torch

K_PROJ = 64
_CHUNK = 1 << 18


def sign_projections(t, name, k=K_PROJ):
    """t: any tensor; name: seeds the signs (crc32) -> [k] float64 = R(name) . t.flatten() with R in {-1, +1}^(k x n)."""
    flat = t.detach().reshape(-1).to(torch.float64).cpu()
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    out = torch.zeros(k, dtype=torch.float64)
    for s in range(0, flat.numel(), _CHUNK):
        piece = flat[s:s + _CHUNK]
        signs = torch.randint(0, 2, (k, piece.numel()), generator=g, dtype=torch.int8).to(torch.float64).mul_(2).sub_(1)
        out += signs @ piece
    return out


def project_all(tensors, k=K_PROJ):
    """{name: tensor} -> (names sorted, [n, k] projections, [n] L2 norms), float64."""
    names = sorted(tensors)
    proj = torch.stack([sign_projections(tensors[n], n, k) for n in names])
    norms = torch.tensor([float(tensors[n].detach().double().norm()) for n in names], dtype=torch.float64)
    return names, proj, norms
