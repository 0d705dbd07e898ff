"""Zero-copy torch views of the library's device buffers (plumbing for torch.distributed collectives over RCCL)."""
from __future__ import annotations


class _Raw:
    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_bytes(ptr: int, nbytes: int, device):
    """uint8 tensor [nbytes] aliasing device memory at `ptr` (the memory stays owned by the library)"""
    import torch
    return torch.as_tensor(_Raw(ptr, nbytes), device=device)


def batch_result_views(ptrs: dict, n_pairs: int, n_features: int, device):
    """torch views of one batch's packed results (the pointers of Context.device_results() / stream_device_results()) in the
    shapes sequence.pack_records takes"""
    import torch
    P, NF = n_pairs, n_features
    kps = device_bytes(ptrs["kps"], 2 * P * NF * 28, device).view(2 * P, NF, 28)
    desc = device_bytes(ptrs["desc"], 2 * P * NF * 32, device).view(2 * P, NF, 32)
    counts = device_bytes(ptrs["counts"], 2 * P * 4, device).view(torch.int32)
    ru = device_bytes(ptrs["right_u"], P * NF * 8, device).view(torch.float64).view(P, NF)
    dp = device_bytes(ptrs["depth"], P * NF * 8, device).view(torch.float64).view(P, NF)
    nm = device_bytes(ptrs["n_match"], P * 4, device).view(torch.int32)
    return kps, desc, counts, ru, dp, nm
