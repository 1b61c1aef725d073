"""tf.math.unsorted_segment_{sum,max,mean,sqrt_n} as free functions (utils/param_helpers.py:9-14).

Generic entry used by user-defined MessagePassing subclasses and by the parity tests of the
aggregation semantics; the built-in layers never call it (they reduce over the batch's cached
Graph).  The segment ids are bucketed with the same kernels that build a Graph.
"""
from __future__ import annotations

import torch

from . import ops


def unsorted_segment_reduce(name: str, data: torch.Tensor, segment_ids: torch.Tensor, num_segments: int):
    if name not in ("sum", "max", "mean", "sqrt_n"):
        raise ValueError(f"Unknown aggregation function: {name}")
    M = data.shape[0]
    dev = data.device
    ids = segment_ids.to(torch.int32)
    if M and (int(ids.min()) < 0 or int(ids.max()) >= num_segments):
        raise ValueError("segment id out of range [0, num_segments)")
    flat = data.reshape(M, -1)
    width = flat.shape[1]
    # one-edge-type graph whose "sources" are message indices and "targets" are segments
    n = max(M, num_segments, 1)
    adj = torch.stack([torch.arange(M, dtype=torch.int32, device=dev), ids], dim=1)
    g = ops.Graph([adj], n)
    rowptr = g.array(ops.G_ROWPTR_BY_DST)[: num_segments + 1]
    col = g.array(ops.G_COL_BY_DST)
    row_scale = None
    if name in ("mean", "sqrt_n"):
        cnt = (rowptr[1:] - rowptr[:-1]).clamp(min=1).to(torch.float32)
        row_scale = (1.0 / cnt) if name == "mean" else torch.rsqrt(cnt)
        if name == "sqrt_n":
            row_scale = 1.0 / torch.sqrt(cnt)
    try:
        if width == 0 or num_segments == 0:
            return torch.zeros((num_segments,) + tuple(data.shape[1:]), dtype=data.dtype, device=dev)
        out = ops.gather_reduce(
            rowptr, col, flat, row_scale=row_scale,
            reduce=ops.REDUCE_MAX if name == "max" else ops.REDUCE_SUM,
        )
    finally:
        g.close()
    return out.reshape((num_segments,) + tuple(data.shape[1:]))
