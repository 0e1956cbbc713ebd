"""The process's auxiliary HIP streams: ONE work stream and ONE communication / side stream per device, shared by every
sampler, reducer and loader of the process.

hipGraph capture needs a non-default stream (samplers), the gradient all-reduce and the small-graph fork of the message
kernel want a second one, and the loader copies batches under its own.  Round 2 created such streams per object (a new
DeviceSampler per generate() batch, one in every GradAllReducer, one inside every sampler context of the library): in the
two-processes-on-one-GPU rehearsal the stream count of a process decided whether gloo's device-tensor all-reduce took
milliseconds or seconds (hardware-queue oversubscription between the two processes).  Now the count is
fixed: default stream + `work` + `comm` (+ the loader's copy stream), whatever objects come and go; `comm` has high priority so
that a gradient bucket's exchange is not queued behind the backward kernels issued after it.
"""
from __future__ import annotations

import torch

_streams = {}


def _get(device, kind: str, priority: int) -> "torch.cuda.Stream":
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("auxiliary streams exist on an AMD GPU ('cuda' device) only")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, kind)
    if key not in _streams:
        _streams[key] = torch.cuda.Stream(device=idx, priority=priority)
    return _streams[key]


def work_stream(device="cuda") -> "torch.cuda.Stream":
    """the non-default stream samplers run (and capture hipGraphs) on"""
    return _get(device, "work", 0)


def comm_stream(device="cuda") -> "torch.cuda.Stream":
    """high-priority stream for RCCL collectives issued from inside the backward, and the side stream small-graph samplers
    fork the message kernel to (the two uses never coincide in one process phase)"""
    return _get(device, "comm", -1)


def copy_stream(device="cuda") -> "torch.cuda.Stream":
    """host -> device copies of the loader"""
    return _get(device, "copy", 0)
