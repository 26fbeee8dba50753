"""Helpers for the -m gpu tests: device buffers are torch CUDA tensors (plumbing only);
every compute call goes through the C ABI (pylbm.Lib)."""
import ctypes as ct

import numpy as np
import torch

import pylbm
from pylbm import _ptr


def dev():
    return torch.device("cuda:0")


def upload_soa(lib, aos):
    """numpy AoS [R,C,Q] (or [R,C]) -> torch SoA [Q,R,C] on the GPU via lbm_aos_to_soa."""
    aos = np.ascontiguousarray(aos, dtype=np.float64)
    if aos.ndim == 2:
        return torch.from_numpy(aos).to(dev())
    R, C, Q = aos.shape
    src = torch.from_numpy(aos).to(dev())
    dst = torch.empty((Q, R, C), dtype=torch.float64, device=dev())
    lib.aos_to_soa(_ptr(dst), _ptr(src), R, C, Q, None)
    torch.cuda.synchronize()
    return dst


def download_aos(lib, soa):
    """torch SoA [Q,R,C] -> numpy AoS [R,C,Q] via lbm_soa_to_aos."""
    if soa.dim() == 2:
        return soa.cpu().numpy()
    Q, R, C = soa.shape
    dst = torch.empty((R, C, Q), dtype=torch.float64, device=dev())
    lib.soa_to_aos(_ptr(dst), _ptr(soa.contiguous()), R, C, Q, None)
    torch.cuda.synchronize()
    return dst.cpu().numpy()


def bits_equal(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def ulp_diff(a, b):
    """max |a-b| in units of the larger magnitude's spacing (diagnostic for failures)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.spacing(np.maximum(np.abs(a), np.abs(b)) + 1e-300)))
