"""Tensor-level wrappers over the C ABI (include/tfgnn.h).

Inputs/outputs are torch tensors living on a ROCm device; torch supplies device memory and the
current HIP stream, every FLOP and byte of the hot path runs in libtfgnn.so.
"""
from __future__ import annotations

import ctypes
import threading
import weakref
import os
from typing import Optional, Sequence

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_TANH, ACT_LEAKY_RELU, ACT_ELU, ACT_SELU, ACT_GELU, ACT_SIGMOID = range(8)
REDUCE_SUM, REDUCE_MAX = 0, 1

_ACT_BY_NAME = {
    "none": ACT_NONE,
    "relu": ACT_RELU,
    "tanh": ACT_TANH,
    "leaky_relu": ACT_LEAKY_RELU,
    "elu": ACT_ELU,
    "selu": ACT_SELU,
    "gelu": ACT_GELU,
    "sigmoid": ACT_SIGMOID,
}


def act_id(name_or_id) -> int:
    if name_or_id is None:
        return ACT_NONE
    if isinstance(name_or_id, int):
        return name_or_id
    try:
        return _ACT_BY_NAME[name_or_id.lower()]
    except KeyError:
        raise ValueError(f"Unknown activation function: {name_or_id}")


class _NullScope:
    __slots__ = ()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL_SCOPE = _NullScope()


def op_scope(name: str, *tensors):
    """Marks a group of library launches that is not a wrapper of this module (the attention kernels of the RGAT layer call
    the C ABI directly): ``with ops.op_scope("rgat_attention_forward", s_src, att): ...``.  Does nothing; a profiler replaces
    this function with one that brackets the region with events (bench.py step_breakdown), like it wraps the wrappers."""
    return _NULL_SCOPE


_ENV_DATA = getattr(os.environ, "_data", None)  # posix: {bytes: bytes}, the dict behind os.environ (sees every later setenv)
_ENV_KEYS = {}


def env(name: str, default: Optional[str] = None) -> Optional[str]:
    """os.environ.get for the switches the layers consult on every call: the Mapping protocol of os.environ costs ~1 us per
    lookup (129 lookups per PPI-sized step, profiles/r06_ppi_host_profile_before.txt); its backing dict a tenth of that."""
    if _ENV_DATA is None:
        return os.environ.get(name, default)
    key = _ENV_KEYS.get(name)
    if key is None:
        key = _ENV_KEYS[name] = os.fsencode(name)
    v = _ENV_DATA.get(key)
    return default if v is None else os.fsdecode(v)


_get_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def _raw_stream() -> int:
    """hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream object per
    call (~4 us: 74 calls = 0.3 ms of the 1.5 ms a PPI-sized step spends on the host, profiles/r06_ppi_host_profile_before.txt);
    the two private accessors it is made of cost ~0.3 us."""
    if _get_raw_stream is not None and _get_device is not None:
        return _get_raw_stream(_get_device())
    return torch.cuda.current_stream().cuda_stream


def _stream() -> int:
    """The launch stream of a library call.  Every wrapper evaluates this right before its C call, which makes it the one
    place where deferred small passes (``aux_defer``) that the coming kernel may depend on are launched first."""
    while _AUX_PENDING and _AUX_URGENT[0]:  # (a launched job may chain an urgent one behind it: the two-pass weight split)
        aux_flush(everything=False)
    return _raw_stream()


# ---- small passes that share a launch (include/tfgnn.h tfgnn_aux_launch) ---------------------------------------------------------
# A deferred job runs at the next library launch on the stream (urgent jobs: something may read their result) or at the next
# flush that has an urgent job / an explicit aux_flush() (non-urgent: results nobody reads before the end of the pass, e.g. the
# split-K reductions of weight gradients).  Deferring only ever moves a pass LATER, and nothing is launched between a deferral
# and the flush that runs it, so every input of a job is complete and none is overwritten in between; non-urgent jobs own
# their inputs (the workspace of a weight-gradient product is kept by the job).  Jobs pending on one stream are flushed before
# work is enqueued for another stream.
# INTERNAL: ``defer=`` / ``defer_combine=`` arguments of the wrappers are for the layer code of this package, which always issues
# the consumer of a deferred result as a library call right behind it.  A caller outside the library that reads a deferred
# result with torch (or synchronises and expects it to exist) must call ``aux_flush()`` first; ``SplitOperand.synced()`` does.
# The queue is guarded by a lock: autograd runs ``GNN.backward`` on its own thread.
_AUX_PENDING = []     # [(AuxJob, keep-alive tuple)]
_AUX_URGENT = [False]
_AUX_STREAM = [None]
_AUX_LOCK = threading.RLock()


def aux_enabled() -> bool:
    import os

    return env("TFGNN_AUX_MERGE", "1") != "0"


def aux_defer(job, keep=(), urgent: bool = True, then=None) -> None:
    """Queue a tfgnn_aux_job (``_lib.AuxJob``; kind 0 = nothing) for the next merged launch.  ``then()`` runs right after the
    launch that carries the job (work that needs the job's result but is itself off the critical path: a weight-gradient
    product behind its factor pass)."""
    if (job.kind == 0 or job.num_blocks == 0) and then is None:
        return
    st = _raw_stream()
    with _AUX_LOCK:
        if _AUX_PENDING and _AUX_STREAM[0] != st:
            aux_flush()
        _AUX_STREAM[0] = st
        _AUX_PENDING.append((job, keep, then))
        if urgent:
            _AUX_URGENT[0] = True


def aux_flush(everything: bool = True) -> None:
    """Launch what has been deferred so far (one launch per 8 jobs) on the stream it was deferred for, then the work chained
    to those jobs.  ``everything``: repeat until nothing is pending (chained work may defer further non-urgent jobs - the
    reduction behind a deferred product); otherwise one round (what ``_stream()`` does before a library call)."""
    while _AUX_PENDING:
        with _AUX_LOCK:
            batch = list(_AUX_PENDING)
            del _AUX_PENDING[:]
            _AUX_URGENT[0] = False
            st = _AUX_STREAM[0]
        live = [j for j, _, _ in batch if j.kind != 0 and j.num_blocks != 0]
        if live:
            jobs = (_lib.AuxJob * len(live))(*live)
            _lib.check(_lib.load().tfgnn_aux_launch(jobs, len(jobs), st))
        for _, _, then in batch:
            if then is not None:
                then()
        if not everything:
            break


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _require_dev(t: torch.Tensor, dtype, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"tf2_gnn_amd: {what} must live on a ROCm device (got {t.device}); there is no CPU fallback"
        )
    if t.dtype != dtype:
        raise TypeError(f"tf2_gnn_amd: {what} must be {dtype}, got {t.dtype}")


def _rowmajor(t: torch.Tensor, what: str):
    """2-D tensor with unit inner stride -> (tensor, leading dimension)."""
    if t.dim() != 2:
        raise ValueError(f"{what} must be 2-D, got shape {tuple(t.shape)}")
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)
    if ld < t.shape[1]:
        t = t.contiguous()
        ld = max(t.shape[1], 1)
    return t, ld


def _writes_out(fn):
    """The kernels write caller-provided ``out`` tensors through raw pointers, which torch does not see: bump the tensor's
    version counter so that everything keyed on (tensor, version) - the remembered split forms of ``sp_rows_of``, the
    split weight operands, the Graph cache - notices the new contents."""
    import functools

    import inspect

    try:
        out_pos = list(inspect.signature(fn).parameters).index("out")
    except ValueError:
        out_pos = None

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        res = fn(*args, **kwargs)
        out = kwargs.get("out")
        if out is None and out_pos is not None and len(args) > out_pos:  # ``out`` passed positionally
            out = args[out_pos]
        if isinstance(out, torch.Tensor):
            torch.autograd.graph.increment_version(out)
        return res

    return wrapper


class _DevArray:
    """Zero-copy view of a device array owned by the C library (``__cuda_array_interface__``)."""

    def __init__(self, ptr: int, count: int, typestr: str, owner):
        self._owner = owner
        self.__cuda_array_interface__ = {
            "shape": (count,),
            "typestr": typestr,
            "data": (ptr, False),
            "version": 2,
            "strides": None,
        }


# graph array ids (include/tfgnn.h tfgnn_graph_array_id)
(G_ROWPTR_BY_DST, G_COL_BY_DST, G_EID_BY_DST, G_COLL_BY_DST, G_ROWPTR_BY_SRC, G_COL_BY_SRC, G_EID_BY_SRC,
 G_COLL_BY_SRC, G_INVDEG_BY_DST, G_INVDEG_EDGE_BY_SRC, G_NODEPTR_BY_DST, G_NODEPTR_BY_SRC,
 G_INVDEG_EDGE_BY_DST, G_SRC2DST_POS, G_TARGET_BY_DST, G_NZ_CPOS_BY_DST, G_NZ_ROW_BY_DST, G_NZ_NODE_BY_DST,
 G_NZ_OFF_BY_DST, G_NZ_NODEPTR_BY_DST, G_NZ_COL_BY_DST, G_NZ_CPOS_BY_SRC, G_NZ_ROW_BY_SRC, G_NZ_NODE_BY_SRC,
 G_NZ_OFF_BY_SRC, G_NZ_NODEPTR_BY_SRC, G_NZ_COL_BY_SRC, G_PATTERN_POS_BY_DST, G_PATTERN_NODE_BY_DST,
 G_PATTERN_TILEMASK_BY_DST, G_PATTERN_NODE_BY_SRC, G_PATTERN_TILEMASK_BY_SRC) = range(32)
_FLOAT_ARRAYS = {G_INVDEG_BY_DST, G_INVDEG_EDGE_BY_SRC, G_INVDEG_EDGE_BY_DST}
_BYTE_ARRAYS = {G_PATTERN_TILEMASK_BY_DST, G_PATTERN_TILEMASK_BY_SRC}
# parts of a graph handle beyond the two sorted edge orders (include/tfgnn.h tfgnn_graph_part)
G_PART_PLAN_TYPED, G_PART_PLAN_NODE, G_PART_COMPACT, G_PART_EDGE_MAPS, G_PART_EDGE_IDS, G_PART_DST_PATTERN, G_PARTS_ALL = 1, 2, 4, 8, 16, 32, 63
G_PARTS_DEFAULT = G_PARTS_ALL & ~G_PART_DST_PATTERN  # the pattern order is built for the layers that ask for it
_VIEW_PARTS = {0: G_PART_PLAN_TYPED, 1: G_PART_PLAN_NODE, 2: G_PART_PLAN_TYPED, 3: G_PART_PLAN_NODE,
               4: G_PART_PLAN_TYPED | G_PART_COMPACT, 5: G_PART_PLAN_TYPED | G_PART_COMPACT,
               6: G_PART_PLAN_TYPED | G_PART_DST_PATTERN, 7: G_PART_PLAN_TYPED | G_PART_DST_PATTERN}


def _array_parts(array_id: int) -> int:
    if array_id == G_SRC2DST_POS:
        return G_PART_EDGE_MAPS | G_PART_EDGE_IDS
    if array_id in (G_EID_BY_DST, G_EID_BY_SRC):
        return G_PART_EDGE_IDS
    if G_PATTERN_POS_BY_DST <= array_id <= G_PATTERN_TILEMASK_BY_SRC:
        return G_PART_DST_PATTERN
    return G_PART_COMPACT if G_NZ_CPOS_BY_DST <= array_id <= G_NZ_COL_BY_SRC else 0


class Graph:
    """Edge-bucketed adjacency of one batch (``tfgnn_graph``); build once, reuse for every layer
    and for forward + backward.  ``adjacency_lists``: sequence of int32 device tensors [E_l, 2]
    with rows (source, target), exactly ``GNNInput.adjacency_lists`` (layers/gnn.py:241-244)."""

    def __init__(self, adjacency_lists: Sequence[torch.Tensor], num_nodes: int, wait: bool = True, parts: int = G_PARTS_DEFAULT):
        """wait=False: the build is only enqueued on the current stream (pipelining the next batch's
        bucketing behind the current step, like the reference's prefetching input pipeline); call
        ``wait()`` - and order the consuming stream after the build stream - before using it.
        parts: which derived tables to build now (G_PART_*; ``GNN.graph_parts`` names what a layer stack reads); a part that
        turns out to be missing is built on first use (``ensure``: correct, but it blocks the host once)."""
        lib = _lib.load()
        adjs = []
        for i, a in enumerate(adjacency_lists):
            _require_dev(a, torch.int32, f"adjacency_lists[{i}]")
            if a.dim() != 2 or a.shape[1] != 2:
                if a.numel() == 0:
                    a = a.reshape(0, 2)
                else:
                    raise ValueError(f"adjacency_lists[{i}] must have shape [E, 2], got {tuple(a.shape)}")
            adjs.append(a.contiguous())
        self._keep = adjs  # the edge lists must stay alive until the build has run
        L = len(adjs)
        ptrs = (ctypes.c_void_p * max(L, 1))(*[a.data_ptr() if a.numel() else None for a in adjs])
        counts = (ctypes.c_int64 * max(L, 1))(*[a.shape[0] for a in adjs])
        handle = ctypes.c_void_p()
        self._h = None
        _lib.check(
            lib.tfgnn_graph_create_parts_async(L, int(num_nodes), ptrs, counts, int(parts) & G_PARTS_ALL, _stream(),
                                               ctypes.byref(handle))
        )
        self._h = handle
        self.parts = int(parts) & G_PARTS_ALL
        if self.parts & G_PART_EDGE_MAPS:
            self.parts |= G_PART_EDGE_IDS
        self._lists = adjs  # ensure(G_PART_EDGE_IDS) reads the edge lists again: they live as long as the handle
        self._pending = True
        self.num_nodes = int(num_nodes)
        self.num_edge_types = L
        self.num_edges = int(sum(a.shape[0] for a in adjs))
        self.edges_per_type = tuple(int(a.shape[0]) for a in adjs)  # host-side: known without touching the device
        self.device = adjs[0].device if adjs else torch.device("cuda")
        self._cache = {}
        if wait:
            self.wait()

    def wait(self):
        """Block the host until the bucketing has finished; raises ValueError for bad node indices."""
        if self._pending:
            if capturing():
                raise RuntimeError("tf2_gnn_amd: a batch's edges cannot be bucketed inside a hipGraph capture (the build reads "
                                   "sizes back); run the step eagerly once on the same adjacency tensors first (CapturedStep warmup)")
            self._pending = False
            self._keep = None
            try:
                _lib.check(_lib.load().tfgnn_graph_wait(self._h))
            except Exception:
                self.close()
                raise
        return self

    def ensure(self, parts: int) -> "Graph":
        """Build the parts that were not requested at creation, on the current stream (tfgnn_graph_ensure; blocks the host)."""
        if parts & ~self.parts:
            self.wait()
            _lib.check(_lib.load().tfgnn_graph_ensure(self._h, int(parts), _stream()))
            self.parts |= int(parts)
        return self

    def array(self, array_id: int) -> torch.Tensor:
        if array_id in self._cache:
            return self._cache[array_id]
        self.ensure(_array_parts(array_id))
        lib = _lib.load()
        p = ctypes.c_void_p()
        n = ctypes.c_int64()
        _lib.check(lib.tfgnn_graph_array(self._h, array_id, ctypes.byref(p), ctypes.byref(n)))
        if n.value == 0 or not p.value:
            dt = torch.float32 if array_id in _FLOAT_ARRAYS else torch.uint8 if array_id in _BYTE_ARRAYS else torch.int32
            t = torch.empty(0, dtype=dt, device=self.device)
        else:
            typestr = "<f4" if array_id in _FLOAT_ARRAYS else "|u1" if array_id in _BYTE_ARRAYS else "<i4"
            t = torch.as_tensor(_DevArray(p.value, n.value, typestr, self), device=self.device)
        self._cache[array_id] = t
        return t

    def nonempty_offsets(self, by_src: bool):
        """host list [L+1]: first compact index of each edge type among the non-empty buckets
        (type-major); the last entry is the number of non-empty buckets."""
        key = ("nz_off", bool(by_src))
        off = self._cache.get(key)
        if off is None:
            self.ensure(G_PART_COMPACT)
            buf = (ctypes.c_int32 * (self.num_edge_types + 1))()
            _lib.check(_lib.load().tfgnn_graph_nonempty_offsets(self._h, int(by_src), buf))
            off = list(buf)
            self._cache[key] = off
        return off

    def close(self):
        """Return the handle's memory to the library; work already enqueued on the current stream may
        still read it (the memory is only reused after that work)."""
        if getattr(self, "_h", None) is not None and self._h:
            self._cache = {}
            h, self._h = self._h, None
            _lib.check(_lib.load().tfgnn_graph_destroy_async(h, _stream()))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_workspaces = {}


def _workspace(device, nbytes: int) -> torch.Tensor:
    key = (device.type, device.index, _raw_stream())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


@_writes_out
def gather_reduce(
    rowptr: torch.Tensor,
    col: torch.Tensor,
    inp: torch.Tensor,
    *,
    edge_weight: Optional[torch.Tensor] = None,
    row_scale: Optional[torch.Tensor] = None,
    reduce: int = REDUCE_SUM,
    pre_act=ACT_NONE,
    post_act=ACT_NONE,
    out: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """out[r] = post_act(row_scale[r] * REDUCE_{e in row r} pre_act(edge_weight[e] * inp[col[e]]))."""
    lib = _lib.load()
    _require_dev(inp, torch.float32, "inp")
    _require_dev(rowptr, torch.int32, "rowptr")
    num_rows = rowptr.numel() - 1
    inp, ld_in = _rowmajor(inp, "inp")
    width = inp.shape[1]
    if out is None:
        out = torch.empty((num_rows, width), dtype=torch.float32, device=inp.device)
    out2, ld_out = _rowmajor(out, "out")
    if out2 is not out:
        raise ValueError("out must have unit inner stride")
    if out.shape[0] != num_rows or out.shape[1] != width:
        raise ValueError(f"out has shape {tuple(out.shape)}, expected {(num_rows, width)}")
    _lib.check(
        lib.tfgnn_csr_gather_reduce(
            _ptr(rowptr), _ptr(col), _ptr(edge_weight), _ptr(row_scale), num_rows, _ptr(inp), ld_in, width,
            _ptr(out), ld_out, int(reduce), act_id(pre_act), act_id(post_act), _stream(),
        )
    )
    return out


(VIEW_BY_DST_TYPED, VIEW_BY_DST_NODE, VIEW_BY_SRC_TYPED, VIEW_BY_SRC_NODE, VIEW_BY_DST_TYPED_COMPACT,
 VIEW_BY_SRC_TYPED_COMPACT, VIEW_BY_DST_TYPED_PATTERN) = range(7)


@_writes_out
def graph_gather(
    graph: "Graph",
    view: int,
    inp: torch.Tensor,
    *,
    col: Optional[torch.Tensor] = None,
    edge_weight: Optional[torch.Tensor] = None,
    row_scale: Optional[torch.Tensor] = None,
    reduce: int = REDUCE_SUM,
    pre_act=ACT_NONE,
    post_act=ACT_NONE,
    out: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """gather_reduce over one of the Graph's bucketed views (with its long-row plan).
    edge_weight: [E] or [E, K] (K heads of width/K floats each)."""
    lib = _lib.load()
    _require_dev(inp, torch.float32, "inp")
    if view == VIEW_BY_DST_TYPED_COMPACT:
        num_rows = graph.nonempty_offsets(False)[-1]
    elif view == VIEW_BY_SRC_TYPED_COMPACT:
        num_rows = graph.nonempty_offsets(True)[-1]
    else:
        num_rows = graph.num_nodes * (graph.num_edge_types if view in (VIEW_BY_DST_TYPED, VIEW_BY_SRC_TYPED) else 1)
    inp, ld_in = _rowmajor(inp, "inp")
    width = inp.shape[1]
    if out is None:
        out = torch.empty((num_rows, width), dtype=torch.float32, device=inp.device)
    out2, ld_out = _rowmajor(out, "out")
    if out2 is not out or out.shape[0] != num_rows or out.shape[1] != width:
        raise ValueError(f"out must be [{num_rows},{width}] with unit inner stride, got {tuple(out.shape)}")
    heads = 1
    if edge_weight is not None:
        edge_weight = edge_weight.contiguous()
        heads = edge_weight.shape[1] if edge_weight.dim() == 2 else 1
    graph.ensure(_VIEW_PARTS[view])
    ws_bytes = lib.tfgnn_graph_gather_workspace_bytes(graph._h, view, width)
    ws = _workspace(inp.device, ws_bytes) if ws_bytes else None
    _lib.check(
        lib.tfgnn_graph_gather_reduce(
            graph._h, view, _ptr(col), _ptr(edge_weight), heads, _ptr(row_scale), _ptr(inp), ld_in, width,
            _ptr(out), ld_out, int(reduce), act_id(pre_act), act_id(post_act), _ptr(ws),
            ws.numel() if ws is not None else 0, _stream(),
        )
    )
    return out


def graph_gather_dot_supported(width: int, heads: int) -> bool:
    return bool(_lib.load().tfgnn_graph_gather_dot_supported(int(width), int(heads)))


def graph_gather_dot(graph: "Graph", view: int, inp: torch.Tensor, *, edge_weight: torch.Tensor, dot_rows: torch.Tensor,
                     dot_pos: Optional[torch.Tensor] = None):
    """graph_gather with per-head edge weights [E, K] that also returns, per edge e of the view and head k, the inner product
    of the gathered row inp[col(e)] with dot_rows[row of e] over the head's columns, written at position dot_pos[e] (None: e)
    (tfgnn_graph_gather_reduce_dot) -> (sums [rows, width], products [E, K])."""
    lib = _lib.load()
    _require_dev(inp, torch.float32, "inp")
    _require_dev(dot_rows, torch.float32, "dot_rows")
    typed = view in (VIEW_BY_DST_TYPED, VIEW_BY_SRC_TYPED)
    if view not in (VIEW_BY_DST_TYPED, VIEW_BY_SRC_TYPED, VIEW_BY_DST_NODE, VIEW_BY_SRC_NODE):
        raise ValueError("graph_gather_dot: the four plain views only")
    num_rows = graph.num_nodes * (graph.num_edge_types if typed else 1)
    inp, ld_in = _rowmajor(inp, "inp")
    dot_rows, ld_dot = _rowmajor(dot_rows, "dot_rows")
    width = inp.shape[1]
    if edge_weight.dim() != 2 or edge_weight.shape[0] != graph.num_edges:
        raise ValueError("graph_gather_dot: edge_weight must be [E, K]")
    edge_weight = edge_weight.contiguous()
    heads = edge_weight.shape[1]
    if tuple(dot_rows.shape) != (num_rows, width):
        raise ValueError(f"dot_rows must be [{num_rows},{width}], got {tuple(dot_rows.shape)}")
    if dot_pos is not None and (dot_pos.dtype != torch.int32 or dot_pos.numel() != graph.num_edges or not dot_pos.is_contiguous()):
        raise ValueError("dot_pos must be a contiguous int32 tensor with one entry per edge")
    dots = torch.empty((graph.num_edges, heads), dtype=torch.float32, device=inp.device)
    if graph.num_edges == 0:
        return torch.zeros((num_rows, width), dtype=torch.float32, device=inp.device), dots
    out = torch.empty((num_rows, width), dtype=torch.float32, device=inp.device)
    graph.ensure(_VIEW_PARTS[view])
    ws_bytes = lib.tfgnn_graph_gather_workspace_bytes(graph._h, view, width)
    ws = _workspace(inp.device, ws_bytes) if ws_bytes else None
    _lib.check(
        lib.tfgnn_graph_gather_reduce_dot(
            graph._h, view, _ptr(edge_weight), heads, _ptr(inp), ld_in, width, _ptr(out), out.stride(0), _ptr(dot_rows), ld_dot,
            _ptr(dot_pos), _ptr(dots), _ptr(ws), ws.numel() if ws is not None else 0, _stream(),
        )
    )
    return out, dots


class DropoutSpec:
    """A layer-input dropout whose mask is not stored: (rate, seed, shape).  The producer of the tensor applied the mask in its
    epilogue (``sp_gemm_nt(..., dropout=...)``); the backward pass hands the spec to a product that RECOMPUTES the mask, or -
    on a path without such an epilogue - materialises it once with ``mask()`` (the tensor ``dropout_forward`` would have
    returned for this seed)."""

    def __init__(self, rate: float, seed: int, shape, device):
        self.rate, self.seed, self.shape, self.device = float(rate), int(seed), tuple(shape), device
        self._mask = None

    @property
    def keep(self) -> float:
        return 1.0 - self.rate

    def mask(self) -> torch.Tensor:
        if self._mask is None:
            self._mask = dropout_mask(self.shape, self.rate, self.seed, self.device)
        return self._mask


def plain_epilogue(out_mul, act_grad):
    """-> (mask tensor | None, (activation, saved) | None): the gradient factors for a kernel WITHOUT the recomputing
    epilogue.  ``out_mul`` may be a DropoutSpec (materialised), ``act_grad`` a triple (activation, saved, saved_scale) -
    the derivative at saved * saved_scale, saved being a dropped activation (one more pass over it here)."""
    if isinstance(out_mul, DropoutSpec):
        out_mul = out_mul.mask()
    if act_grad is not None and len(act_grad) == 3:
        act, saved, scale = act_grad
        if float(scale) != 1.0:
            saved = add_scale(saved, saved, 0.5 * float(scale))
        act_grad = (act, saved)
    return out_mul, act_grad


def _native_epilogue(out_mul, act_grad, dropout, saved_scale):
    """the same specs for the split-operand product, which takes them as they are"""
    spec = out_mul if isinstance(out_mul, DropoutSpec) else None
    if spec is not None:
        if dropout is not None:
            raise ValueError("two dropout masks on one product")
        dropout, out_mul = (spec.rate, spec.seed), None
    if act_grad is not None and len(act_grad) == 3:
        saved_scale = float(act_grad[2])
        act_grad = (act_grad[0], act_grad[1])
        if (spec is not None and act_grad[0] == "relu" and abs(saved_scale - spec.keep) < 1e-12 and saved_scale != 1.0
                and tuple(act_grad[1].shape) == spec.shape):
            # the saved tensor is the relu output dropped with THIS mask: it is positive exactly where the unit was kept and
            # active - no mask needs to be recomputed (tfgnn_sp_gemm_nt_dropout, seed UINT64_MAX)
            dropout = (spec.rate, 0xFFFFFFFFFFFFFFFF)
    return out_mul, act_grad, dropout, saved_scale


@_writes_out
def gemm_grad(a, b, *, trans_b=False, out=None, out_mul=None, act_grad=None, accumulate=False) -> torch.Tensor:
    """out = (a @ op(b)) * out_mul * act'(saved) (+ out if ``accumulate``): an input-gradient product with the element-wise
    factors of the next backward step (dropout mask ``out_mul``, ``act_grad = (activation name, saved tensor)``) applied in the
    GEMM epilogue when the active kernel has one (tfgnn_gemm_grad_epilogue), by separate kernels otherwise.
    Use the RETURN value: on the unfused route the factors are applied out of place and ``out`` only holds the raw
    product (with ``accumulate`` the result is added into ``out`` and ``out`` is returned)."""
    if accumulate and out is None:
        raise ValueError("accumulate=True needs out")
    out_mul, act_grad = plain_epilogue(out_mul, act_grad)
    if out_mul is None and act_grad is None:
        return gemm(a, b, trans_b=trans_b, out=out, accumulate=accumulate)
    lib = _lib.load()
    a2, lda = _rowmajor(a, "a")
    b2, ldb = _rowmajor(b, "b")
    M, K = a2.shape
    N = b2.shape[0] if trans_b else b2.shape[1]
    res = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=a.device)
    res2, ldc = _rowmajor(res, "out")
    act_name, saved = act_grad if act_grad is not None else (None, None)

    def plain(t):
        return t is None or (t.dim() == 2 and t.stride(1) == 1 and tuple(t.shape) == (M, N))

    if plain(out_mul) and plain(saved) and res2 is res:
        ws_bytes = lib.tfgnn_gemm_workspace_bytes(M, N, K)
        ws = _workspace(a.device, ws_bytes) if ws_bytes else None
        rc = lib.tfgnn_gemm_grad_epilogue(
            0, int(trans_b), M, N, K, _ptr(a2), lda, _ptr(b2), ldb, _ptr(res), ldc, _ptr(out_mul),
            out_mul.stride(0) if out_mul is not None else 0, act_id(act_name), _ptr(saved),
            saved.stride(0) if saved is not None else 0, int(accumulate), _ptr(ws), ws.numel() if ws is not None else 0, _stream(),
        )
        if rc == 0:
            return res
        if rc != -4:  # TFGNN_ERR_UNSUPPORTED: no fused epilogue for this mode / shape
            _lib.check(rc)
    res = gemm(a, b, trans_b=trans_b, out=None if accumulate else out)
    if out_mul is not None:
        res = mul(res, out_mul)
    if act_grad is not None:
        res = activation_backward(act_name, res, saved)
    if accumulate:
        out.copy_(add_scale(out, res, 1.0))
        return out
    return res


GEMM_FP32, GEMM_BF16X3, GEMM_BF16X3_EXACT, GEMM_F16X2 = 0, 6, 9, 3
_GEMM_MODE_NAMES = {"fp32": GEMM_FP32, "bf16x3": GEMM_BF16X3, "bf16x3_9": GEMM_BF16X3_EXACT, "f16x2": GEMM_F16X2}
_f16x2 = [None]  # None: not decided yet (environment TFGNN_GEMM_MODE; unset = f16x2, the default since round 3)
_spread_warned = [False]


def _f16x2_on() -> bool:
    """Is the library in mode f16x2 (include/tfgnn.h TFGNN_GEMM_F16X2)?  The library itself holds the mode and demotes it when
    the spread guard of the split weight-gradient product trips (tfgnn_gemm_get_mode); this mirror only warns once."""
    if (_GUARD_HOLD[0] or _in_late_trip[0]) and _f16x2[0]:
        return True  # a caller that checks the guard synchronously at the end of its pass decides what a trip demotes
    if _f16x2[0] and env("TFGNN_GUARD_IGNORE", "0") == "1":
        return True  # PROBING ONLY (what would a workload cost if its products stayed on split operands): results unguarded
    lib = _lib.load()
    on = lib.tfgnn_gemm_get_mode() == GEMM_F16X2  # (the library takes the mode off - sticky - when it finds the flag up)
    if _f16x2[0] and not on and _LATE_TRIP_POLICIES and lib.tfgnn_sp_spread_flag(0) and not capturing():
        # A pass that was not checked synchronously tripped the guard, and the library has just taken the whole mode off the split
        # operands for it.  If the stacks that ran such passes still have a stage of their policy to give (GNN._on_late_guard_trip:
        # only the weight-gradient products whose operand rows are spread change kernels, one stage per pass), the mode goes back
        # on - tfgnn_gemm_set_mode waits for the device and clears the flag - and those stacks' next passes are checked
        # synchronously again.  The tripping pass itself is not recomputed.  (ONE read of the flag decides - the library's: a
        # look at the flag here followed by the library's own could straddle the moment the device raises it.)
        _in_late_trip[0] = True
        try:
            handled = [w for w in (fn() for fn in list(_LATE_TRIP_POLICIES.values())) if w]
            if handled:
                aux_flush()
                _lib.check(lib.tfgnn_gemm_set_mode(GEMM_F16X2))
                on = True
                import warnings

                warnings.warn("tf2_gnn_amd: operand rows of a weight-gradient product spread beyond the range of the split-operand "
                              "product that ran, in a pass that was not checked synchronously (its gradients may have lost "
                              "low-order rows): " + "; ".join(handled) + "; the next passes are checked again")
        finally:
            _in_late_trip[0] = False
    if _f16x2[0] and not on and not _spread_warned[0] and lib.tfgnn_sp_spread_flag(0):
        _spread_warned[0] = True
        import warnings

        warnings.warn("tf2_gnn_amd: operand rows of a weight-gradient product spread over more than 2^20 in magnitude; "
                      "the f16x2 layer paths are switched to the exact bf16x3 kernels (ops.set_gemm_mode('f16x2') re-arms them)")
    _f16x2[0] = on
    return on


def set_gemm_mode(mode) -> int:
    """Select how the Dense products are evaluated: "fp32" (fp32 MFMA), "bf16x3" (exact 3-way bf16 split of both operands,
    6 piece products), "bf16x3_9" (all 9) or "f16x2": the layers hand the hot products pre-split SP16 operands
    (tfgnn_sp_gemm_*: 2-way fp16 split, 3 piece products, the gather writes the operand) and every other product runs as in
    "bf16x3" - include/tfgnn.h, tfgnn_gemm_set_mode.  "f16x2" is the DEFAULT of the library (environment TFGNN_GEMM_MODE
    overrides); its spread guard (tfgnn_sp_spread_flag) demotes it to "bf16x3" when an operand's row scales spread over more
    than 2^20; setting "f16x2" again re-arms the guard (waits for the device).  Returns the previous mode id."""
    lib = _lib.load()
    prev = get_gemm_mode()
    mode = _GEMM_MODE_NAMES.get(mode, mode)
    aux_flush()
    _lib.check(lib.tfgnn_gemm_set_mode(mode))
    _f16x2[0] = mode == GEMM_F16X2
    if mode == GEMM_F16X2:
        REARM_EPOCH[0] += 1  # stacks that demoted their own Dense products take the split kernels again (GNN._dense_f16x2)
    return prev


def get_gemm_mode() -> int:
    if _f16x2_on():
        return GEMM_F16X2
    return _lib.load().tfgnn_gemm_get_mode()


_GUARD_HOLD = [0]
_in_late_trip = [False]
_LATE_TRIP_POLICIES = weakref.WeakValueDictionary()  # id -> bound-method holder of the stacks with a staged guard policy


class _LateTripHook:
    """Keeps a stack's policy callable reachable through a weak dictionary: the hook lives as long as its owner does."""

    def __init__(self, owner):
        self._owner = weakref.ref(owner)

    def __call__(self):
        o = self._owner()
        return o._on_late_guard_trip() if o is not None else None


def register_late_trip_policy(owner) -> "_LateTripHook":
    """``owner._on_late_guard_trip() -> Optional[str]`` is asked when an unchecked pass trips the spread guard (``_f16x2_on``).
    The caller keeps the returned hook alive (an attribute of the owner)."""
    hook = _LateTripHook(owner)
    _LATE_TRIP_POLICIES[id(hook)] = hook
    return hook


REARM_EPOCH = [0]  # times the f16x2 mode was (re-)armed by set_gemm_mode("f16x2")


def demote_gemm_mode() -> int:
    """Let the library act on a set spread flag NOW, also inside ``hold_spread_guard`` -> the mode in force afterwards."""
    held, _GUARD_HOLD[0] = _GUARD_HOLD[0], 0
    try:
        return get_gemm_mode()
    finally:
        _GUARD_HOLD[0] = held


class hold_spread_guard:
    """Inside this context a tripped spread flag does not demote the mode on sight: the caller reads the flag itself
    (``f16x2_guard_tripped_sync``) when its pass is complete and decides what to demote (``GNN.backward``)."""

    def __enter__(self):
        _f16x2_on()
        _GUARD_HOLD[0] += 1
        return self

    def __exit__(self, *exc):
        _GUARD_HOLD[0] -= 1
        return False


def rearm_spread_guard() -> None:
    """Clear the spread flag WITHOUT demoting the mode (the caller has dealt with the product that tripped it).  Waits for the
    stream first: a factor computation still in flight must not set it again."""
    aux_flush()
    torch.cuda.current_stream().synchronize()
    _lib.load().tfgnn_sp_spread_flag(1)


KERNEL_FAMILIES = ("gemm_fp32", "gemm_bf16x3", "sp_nt", "sp_tn", "gather_sp", "gather", "fused_nt", "gemm_stream", "stream_f16x2")


def launch_counts() -> dict:
    """tfgnn_launch_counts as a dict: kernel family -> launches this process has enqueued so far (host counters, no device
    work).  bench.py reports the per-step differences as ``products`` so that a line says which product kernels ran."""
    buf = (ctypes.c_int64 * len(KERNEL_FAMILIES))()
    _lib.check(_lib.load().tfgnn_launch_counts(buf, len(KERNEL_FAMILIES)))
    return dict(zip(KERNEL_FAMILIES, (int(v) for v in buf)))


def f16x2_guard_flag_async() -> bool:
    """The spread flag as the host sees it NOW, without waiting for the device (what products enqueued earlier have reported
    so far)."""
    return bool(_lib.load().tfgnn_sp_spread_flag(0))


def f16x2_guard_tripped_sync() -> bool:
    """Wait for the current stream, then read the spread guard: True if a split weight-gradient product enqueued so far met
    operand rows spread over more than 2^20 (its result may have lost low-order rows).  The synchronous form of the guard:
    ``GNN.backward`` uses it for the first backward passes of a model and recomputes a tripped pass on the exact kernels."""
    aux_flush()
    torch.cuda.current_stream().synchronize()
    return bool(_lib.load().tfgnn_sp_spread_flag(0))


@_writes_out
def gemm(
    a: torch.Tensor,
    b: torch.Tensor,
    *,
    trans_a: bool = False,
    trans_b: bool = False,
    bias: Optional[torch.Tensor] = None,
    act=ACT_NONE,
    out: Optional[torch.Tensor] = None,
    accumulate: bool = False,
) -> torch.Tensor:
    """out = act(op(a) @ op(b) + bias) (+ out).  a, b: 2-D fp32 device tensors, unit inner stride."""
    lib = _lib.load()
    _require_dev(a, torch.float32, "a")
    _require_dev(b, torch.float32, "b")
    a, lda = _rowmajor(a, "a")
    b, ldb = _rowmajor(b, "b")
    M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    Kb, N = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    if K != Kb:
        raise ValueError(f"gemm: inner dimensions differ ({K} vs {Kb})")
    if (not trans_a and not trans_b and M >= 4096 and K * N <= (1 << 22) and K >= 64 and K % 4 == 0
            and (N % 320 == 0 or N % 128 == 0) and get_gemm_mode() != GEMM_FP32):
        # a small [K, N] right operand against many rows: the split-operand kernel stages K-contiguous operands
        # fastest (no register transposes; QM9-sized GRU product 1.44 -> 0.8 ms), so hand it B^T (one small copy)
        b, ldb, trans_b = transpose_batched(b), K, True
    if out is None:
        if accumulate:
            raise ValueError("accumulate=True needs out")
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    out2, ldc = _rowmajor(out, "out")
    if out2 is not out or tuple(out.shape) != (M, N):
        raise ValueError(f"out must be [{M},{N}] with unit inner stride")
    if bias is not None:
        _require_dev(bias, torch.float32, "bias")
        if bias.numel() != N:
            raise ValueError("bias must have N elements")
        bias = bias.contiguous()
    ws_bytes = lib.tfgnn_gemm_workspace_bytes(M, N, K)
    ws = _workspace(a.device, ws_bytes) if ws_bytes else None
    _lib.check(
        lib.tfgnn_gemm(
            int(trans_a), int(trans_b), M, N, K, _ptr(a), lda, _ptr(b), ldb, _ptr(out), ldc, _ptr(bias),
            act_id(act), int(accumulate), _ptr(ws), ws.numel() if ws is not None else 0, _stream(),
        )
    )
    return out


_ident_ptrs = {}


def _identity_rowptr(device, n: int) -> torch.Tensor:
    """[0, 1, ..., n] int32 (rows of one element each: turns tfgnn_csr_gather_reduce into a plain row gather)."""
    key = str(device)
    t = _ident_ptrs.get(key)
    if t is None or t.numel() < n + 1:
        t = torch.arange(max(n + 1, 1 << 16), dtype=torch.int32, device=device)
        _ident_ptrs[key] = t
    return t[: n + 1]


@_writes_out
def gemm_gathered(a: torch.Tensor, row_index: torch.Tensor, b: torch.Tensor, *, trans_b: bool = False,
                  bias: Optional[torch.Tensor] = None, act=ACT_NONE, out: Optional[torch.Tensor] = None,
                  accumulate: Optional[str] = None) -> torch.Tensor:
    """out[m] = act(a[row_index[m]] @ op(b) + bias)  (tfgnn_gemm_gathered: tf.nn.embedding_lookup + Dense, the gathered rows are
    never written).  ``accumulate``: None, "after" (out += act(..)) or "before" (out = act(out + ..), the second half of a
    product over concatenated inputs).  Shapes the streaming kernel does not take run as a row gather + gemm."""
    lib = _lib.load()
    _require_dev(a, torch.float32, "a")
    _require_dev(b, torch.float32, "b")
    _require_dev(row_index, torch.int32, "row_index")
    if accumulate not in (None, "after", "before"):
        raise ValueError('accumulate is None, "after" or "before"')
    a2, lda = _rowmajor(a, "a")
    b2, ldb = _rowmajor(b, "b")
    K = a2.shape[1]
    Kb, N = (b2.shape[1], b2.shape[0]) if trans_b else (b2.shape[0], b2.shape[1])
    if K != Kb:
        raise ValueError(f"gemm_gathered: inner dimensions differ ({K} vs {Kb})")
    M = row_index.numel()
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs out")
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    out2, ldc = _rowmajor(out, "out")
    if out2 is not out or tuple(out.shape) != (M, N):
        raise ValueError(f"out must be [{M},{N}] with unit inner stride")
    if bias is not None:
        _require_dev(bias, torch.float32, "bias")
        bias = bias.contiguous()
    if M == 0:
        return out
    row_index = row_index.contiguous()
    rc = lib.tfgnn_gemm_gathered(
        int(trans_b), M, N, K, _ptr(a2), lda, a2.shape[0], _ptr(row_index), _ptr(b2), ldb, _ptr(out), ldc, _ptr(bias),
        act_id(act), {None: 0, "after": 1, "before": 2}[accumulate], _stream(),
    )
    if rc == 0:
        return out
    if rc != -4:  # TFGNN_ERR_UNSUPPORTED: the caller-side route below
        _lib.check(rc)
    rows = gather_reduce(_identity_rowptr(a.device, M), row_index, a2)
    if accumulate == "before":
        gemm(rows, b2, trans_b=trans_b, bias=bias, out=out, accumulate=True)
        if act_id(act) != act_id(ACT_NONE):
            if out.is_contiguous():
                activation_forward(act, out, out=out)
            else:
                out.copy_(activation_forward(act, out))
        return out
    return gemm(rows, b2, trans_b=trans_b, bias=bias, act=act, out=out, accumulate=accumulate == "after")


@_writes_out
def activation_forward(act, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    _require_dev(x, torch.float32, "x")
    x = x.contiguous()
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib.tfgnn_activation_forward(act_id(act), _ptr(x), _ptr(out), x.numel(), _stream()))
    return out


@_writes_out
def activation_backward(act, dy: torch.Tensor, saved: torch.Tensor, out: Optional[torch.Tensor] = None,
                        mul: Optional[torch.Tensor] = None):
    """dx = dy * act'(.), derivative evaluated from the saved output (saved input for gelu).  ``mul``: dx = (dy * mul) * act'(.)
    in the same pass (tfgnn_activation_backward_mul)."""
    lib = _lib.load()
    _require_dev(dy, torch.float32, "dy")
    dy = dy.contiguous()
    saved = saved.contiguous()
    if out is None:
        out = torch.empty_like(dy)
    if mul is not None:
        mul = mul.contiguous()
        if mul.shape != dy.shape:
            raise ValueError("activation_backward: mul must have the gradient's shape")
        if (dy.data_ptr() | saved.data_ptr() | mul.data_ptr() | out.data_ptr()) % 16 == 0:
            _lib.check(lib.tfgnn_activation_backward_mul(act_id(act), _ptr(dy), _ptr(saved), _ptr(mul), _ptr(out), dy.numel(),
                                                         _stream()))
            return out
        tmp = torch.empty_like(dy)  # (unaligned views: the two-pass form)
        _lib.check(lib.tfgnn_mul(_ptr(dy), _ptr(mul), _ptr(tmp), dy.numel(), _stream()))
        dy = tmp
    _lib.check(
        lib.tfgnn_activation_backward(act_id(act), _ptr(dy), _ptr(saved), _ptr(out), dy.numel(), _stream())
    )
    return out


def gru_gates_forward(mx, mh, h, save_gates: bool = True):
    """h' = GRU gate math on the two pre-computed matmuls ([ext] Keras GRUCell reset_after=True)."""
    lib = _lib.load()
    V, H = h.shape
    h = h.contiguous()
    h_new = torch.empty_like(h)
    gates = torch.empty((V, 3 * H), dtype=torch.float32, device=h.device) if save_gates else None
    _lib.check(
        lib.tfgnn_gru_gates_forward(_ptr(mx), _ptr(mh), _ptr(h), _ptr(h_new), _ptr(gates), V, H, _stream())
    )
    return h_new, gates


_gru_perm = {}


def gru_kernel_regrouped(kernel: torch.Tensor, bias: Optional[torch.Tensor]):
    """Keras GRUCell kernel [K, 3H] (+ bias [3H]) -> the operand form of tfgnn_gemm_gru: [3H, K], K contiguous, rows
    regrouped per 192-row block as [z | r | h] of the same 64 units (an index gather on a small weight matrix)."""
    K, H3 = kernel.shape
    H = H3 // 3
    key = (H, kernel.device)
    perm = _gru_perm.get(key)
    if perm is None:  # (seven tiny launches per call otherwise: once per width and device)
        t = torch.arange(H // 64, device=kernel.device).view(-1, 1, 1)
        g = torch.arange(3, device=kernel.device).view(1, -1, 1)
        c = torch.arange(64, device=kernel.device).view(1, 1, -1)
        perm = _gru_perm[key] = (g * H + t * 64 + c).reshape(-1)
    return transpose_batched(kernel.index_select(1, perm)), (None if bias is None else bias.index_select(0, perm).contiguous())


def gemm_gru(x, kernel, bias, mh, h, save_gates: bool = True):
    """h' = GRUCell gate math on (x @ kernel + bias, mh, h) with the matmul and the gates in one kernel
    (tfgnn_gemm_gru) -> (h', gates or None), or None when the library has no such kernel for this mode / shape."""
    H = h.shape[1]
    if get_gemm_mode() == GEMM_FP32 or H % 64 != 0 or kernel.shape[0] < 64 or kernel.shape[0] % 4 != 0:
        return None
    lib = _lib.load()
    x, ldx = _rowmajor(x, "x")
    mh = mh.contiguous()
    h = h.contiguous()
    kt, bp = gru_kernel_regrouped(kernel, bias)
    V, K = x.shape
    h_new = torch.empty_like(h)
    gates = torch.empty_like(mh) if save_gates else None
    rc = lib.tfgnn_gemm_gru(V, H, K, _ptr(x), ldx, _ptr(kt), _ptr(bp), _ptr(mh), _ptr(h), _ptr(h_new), _ptr(gates), _stream())
    if rc == -4:
        return None
    _lib.check(rc)
    return h_new, gates


def gemm_gru2(x, kernel, bias, h, recurrent_kernel, recurrent_bias):
    """The whole GRUCell forward in ONE kernel (tfgnn_gemm_gru2): h' = GRU(x @ kernel + bias, h @ recurrent_kernel +
    recurrent_bias, h) - mh is neither produced by a product of its own nor read back (3.5 GB per layer at the QM9 size).
    -> (h', gates [V, 3H], mh [V, 3H] of which ONLY the candidate third is written: all the backward pass reads), or None when
    the library has no such kernel (mode other than f16x2, H not in {64, 128}, fewer than 65536 rows)."""
    H = h.shape[1]
    if get_gemm_mode() != GEMM_F16X2 or H not in (64, 128) or kernel.shape[0] != H or h.shape[0] < 65536:
        return None
    lib = _lib.load()
    x, ldx = _rowmajor(x, "x")
    h = h.contiguous()
    kt, bp = gru_kernel_regrouped(kernel, bias)
    rt, rbp = gru_kernel_regrouped(recurrent_kernel, recurrent_bias)
    V = x.shape[0]
    h_new = torch.empty_like(h)
    gates = torch.empty((V, 3 * H), dtype=torch.float32, device=h.device)
    mh = torch.empty((V, 3 * H), dtype=torch.float32, device=h.device)
    rc = lib.tfgnn_gemm_gru2(V, H, _ptr(x), ldx, _ptr(kt), _ptr(bp), _ptr(h), _ptr(rt), _ptr(rbp), _ptr(h_new), _ptr(gates), _ptr(mh),
                             _stream())
    if rc == -4:
        return None
    _lib.check(rc)
    return h_new, gates, mh


def gru_gates_backward(dh_new, gates, mh, h):
    lib = _lib.load()
    V, H = h.shape
    dh_new = dh_new.contiguous()
    dmx = torch.empty((V, 3 * H), dtype=torch.float32, device=h.device)
    dmh = torch.empty_like(dmx)
    dh_direct = torch.empty_like(h)
    _lib.check(
        lib.tfgnn_gru_gates_backward(
            _ptr(dh_new), _ptr(gates), _ptr(mh), _ptr(h.contiguous()), _ptr(dmx), _ptr(dmh), _ptr(dh_direct),
            V, H, _stream(),
        )
    )
    return dmx, dmh, dh_direct


def gru_gates_backward_sp(dh_new, gates, mh, h, out_mul=None):
    """gru_gates_backward with dmx / dmh written ONLY as SP16 split operands (one scale per row) and the bias gradients
    [2, 3H] folded in (tfgnn_gru_gates_backward_sp) -> (dmx_sp, dmh_sp, dh_direct, bias_grad), or None when the library has
    no such kernel for this width (H % 64 != 0 or H > 512).  ``out_mul`` [V, H]: factor of dh_direct (a dropout mask); a
    DropoutSpec: the mask is recomputed from (rate, seed) in the kernel (tfgnn_gru_gates_backward_sp_dropout)."""
    lib = _lib.load()
    V, H = h.shape
    if H % 64 != 0 or H > 512:
        return None
    dh_new = dh_new.contiguous()
    dev = h.device
    dmx = SplitOperand(torch.empty((V, 3 * H * 4), dtype=torch.uint8, device=dev), torch.empty((V, 1), dtype=torch.float32, device=dev),
                       V, 3 * H, 3 * H)
    dmh = SplitOperand(torch.empty((V, 3 * H * 4), dtype=torch.uint8, device=dev), torch.empty((V, 1), dtype=torch.float32, device=dev),
                       V, 3 * H, 3 * H)
    dh_direct = torch.empty_like(h)
    bias_grad = torch.empty((2, 3 * H), dtype=torch.float32, device=dev)
    ws_bytes = lib.tfgnn_gru_gates_backward_sp_workspace_bytes(V, H)
    ws = _workspace(dev, ws_bytes) if ws_bytes else None
    if isinstance(out_mul, DropoutSpec):
        if out_mul.shape != (V, H):
            raise ValueError("gru_gates_backward_sp: the dropout spec is not one of the layer input")
        rc = lib.tfgnn_gru_gates_backward_sp_dropout(_ptr(dh_new), _ptr(gates), _ptr(mh), _ptr(h.contiguous()), _ptr(dmx.data),
                                                     _ptr(dmx.inv_scale), _ptr(dmh.data), _ptr(dmh.inv_scale), _ptr(dh_direct),
                                                     float(out_mul.rate), int(out_mul.seed) & (2**64 - 1), _ptr(bias_grad), V, H,
                                                     _ptr(ws), ws.numel() if ws is not None else 0, _stream())
    else:
        rc = lib.tfgnn_gru_gates_backward_sp(_ptr(dh_new), _ptr(gates), _ptr(mh), _ptr(h.contiguous()), _ptr(dmx.data),
                                             _ptr(dmx.inv_scale), _ptr(dmh.data), _ptr(dmh.inv_scale), _ptr(dh_direct),
                                             _ptr(out_mul.contiguous() if out_mul is not None else None), _ptr(bias_grad), V, H,
                                             _ptr(ws), ws.numel() if ws is not None else 0, _stream())
    if rc == -4:
        return None
    _lib.check(rc)
    return dmx, dmh, dh_direct, bias_grad


def colsum(x: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    x, ld = _rowmajor(x, "x")
    out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
    ws_bytes = lib.tfgnn_colsum_workspace_bytes(x.shape[0], x.shape[1])
    ws = _workspace(x.device, ws_bytes) if ws_bytes else None
    _lib.check(lib.tfgnn_colsum(_ptr(x), x.shape[0], x.shape[1], ld, _ptr(out), _ptr(ws),
                                ws.numel() if ws is not None else 0, _stream()))
    return out


def add_scale(x: torch.Tensor, y: torch.Tensor, alpha: float) -> torch.Tensor:
    """alpha * (x + y)"""
    lib = _lib.load()
    x = x.contiguous()
    y = y.contiguous()
    out = torch.empty_like(x)
    _lib.check(lib.tfgnn_add_scale(_ptr(x), _ptr(y), float(alpha), _ptr(out), x.numel(), _stream()))
    return out


def dropout_forward(x: torch.Tensor, rate: float, seed: int, want_mask: bool = True):
    """-> (y, mask) with mask in {0, 1/(1-rate)}.  In f16x2 mode a 2-D result that can be a split operand (width a
    multiple of 16, <= 512) is also written in the SP16 format by the same kernel and remembered for ``sp_rows_of``.
    want_mask=False: the mask is not stored (-> (y, None)); whoever needs it recomputes it from (rate, seed)
    (``DropoutSpec``, the epilogues of the split-operand products, ``dropout_mask``)."""
    lib = _lib.load()
    x = x.contiguous()
    y = torch.empty_like(x)
    mask = torch.empty_like(x) if want_mask else None
    if _f16x2_on() and x.dim() == 2 and x.shape[1] % 16 == 0 and 32 <= x.shape[1] <= 512 and x.shape[0] > 0:
        rows, cols = x.shape
        op = SplitOperand(torch.empty((rows, cols * 4), dtype=torch.uint8, device=x.device),
                          torch.empty((rows, 1), dtype=torch.float32, device=x.device), rows, cols, cols)
        _lib.check(lib.tfgnn_dropout_forward_sp(_ptr(x), _ptr(y), _ptr(mask), rows, cols, float(rate), int(seed) & (2**64 - 1),
                                                _ptr(op.data), op.data.stride(0), _ptr(op.inv_scale), _stream()))
        _remember_split_rows(y, op)
        return y, mask
    _lib.check(
        lib.tfgnn_dropout_forward(_ptr(x), _ptr(y), _ptr(mask), x.numel(), float(rate), int(seed) & (2**64 - 1), _stream())
    )
    return y, mask


def dropout_epoch() -> int:
    """The dropout epoch (include/tfgnn.h: masks are a function of (seed, element, epoch); 0 unless something advanced it).
    Waits for the current stream."""
    v = ctypes.c_uint32()
    aux_flush()
    _lib.check(_lib.load().tfgnn_dropout_epoch_get(ctypes.byref(v), _raw_stream()))
    return int(v.value)


def dropout_epoch_advance() -> None:
    """epoch += 1 by a kernel on the current stream (capturable: the first node of ``capture.CapturedStep``)."""
    _lib.check(_lib.load().tfgnn_dropout_epoch_advance(_stream()))


def dropout_epoch_set(value: int) -> None:
    _lib.check(_lib.load().tfgnn_dropout_epoch_set(int(value) & 0xFFFFFFFF, _stream()))


def capturing() -> bool:
    """Is the current stream being captured into a hipGraph (``capture.CapturedStep``)?  Host synchronisation is illegal then."""
    return bool(torch.cuda.is_current_stream_capturing())


def dropout_mask(shape, rate: float, seed: int, device=None) -> torch.Tensor:
    """The mask (0 or 1/(1-rate)) a dropout call with this seed draws for a tensor of ``shape`` - what ``dropout_forward``
    returns and what the fused producers (``sp_gemm_nt(..., dropout=(rate, seed))``) apply without storing it."""
    lib = _lib.load()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    mask = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    if mask.numel():
        _lib.check(lib.tfgnn_dropout_forward(None, None, _ptr(mask), mask.numel(), float(rate), int(seed) & (2**64 - 1), _stream()))
    return mask


def clip(x: torch.Tensor, lower: Optional[float], upper: Optional[float]) -> torch.Tensor:
    """tf.minimum(tf.maximum(x, lower), upper); None = no bound."""
    lib = _lib.load()
    x = x.contiguous()
    y = torch.empty_like(x)
    lo = float("-inf") if lower is None else float(lower)
    hi = float("inf") if upper is None else float(upper)
    _lib.check(lib.tfgnn_clip(_ptr(x), x.numel(), lo, hi, _ptr(y), _stream()))
    return y


def clip_backward(dy: torch.Tensor, x: torch.Tensor, lower: Optional[float], upper: Optional[float]) -> torch.Tensor:
    lib = _lib.load()
    dy, x = dy.contiguous(), x.contiguous()
    dx = torch.empty_like(dy)
    lo = float("-inf") if lower is None else float(lower)
    hi = float("inf") if upper is None else float(upper)
    _lib.check(lib.tfgnn_clip_backward(_ptr(dy), _ptr(x), x.numel(), lo, hi, _ptr(dx), _stream()))
    return dx


def mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    a = a.contiguous()
    b = b.contiguous()
    out = torch.empty_like(a)
    _lib.check(lib.tfgnn_mul(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()))
    return out


def layernorm_forward(x, gamma, beta, eps: float = 1e-3):
    lib = _lib.load()
    x = x.contiguous()
    rows, H = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    _lib.check(
        lib.tfgnn_layernorm_forward(_ptr(x), _ptr(gamma), _ptr(beta), float(eps), rows, H, _ptr(y), _ptr(mean),
                                    _ptr(rstd), _stream())
    )
    return y, mean, rstd


def layernorm_backward(dy, x, gamma, mean, rstd):
    """-> (dx, dgamma, dbeta)"""
    lib = _lib.load()
    dy = dy.contiguous()
    rows, H = x.shape
    dx = torch.empty_like(x)
    dy_xhat = torch.empty_like(x)
    _lib.check(
        lib.tfgnn_layernorm_backward(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(rstd), rows, H, _ptr(dx),
                                     _ptr(dy_xhat), _stream())
    )
    return dx, colsum(dy_xhat), colsum(dy)


def permute_021(x: torch.Tensor) -> torch.Tensor:
    """[A, B, C] -> [B, A, C] (contiguous copy)."""
    lib = _lib.load()
    x = x.contiguous()
    A, B, C = x.shape
    out = torch.empty((B, A, C), dtype=torch.float32, device=x.device)
    _lib.check(lib.tfgnn_permute_021(_ptr(x), A, B, C, _ptr(out), _stream()))
    return out


def edge_aggregate_backward(msg, target, grad_agg, *, msg_row=None, edge_weight=None, node_scale=None, pre_act=ACT_NONE,
                            reduce=REDUCE_SUM, agg_max=None, num_selected=None, phase=1) -> torch.Tensor:
    """Per-edge gradient through agg[t] = node_scale[t] * REDUCE_{e->t} pre_act(w_e * msg[row_e]) (phase 1), or the
    0/1 indicator of the edges that attain a target's maximum (phase 0).  -> [E, width]"""
    lib = _lib.load()
    _require_dev(msg, torch.float32, "msg")
    msg, ld = _rowmajor(msg, "msg")
    E = target.numel()
    width = msg.shape[1]
    out = torch.empty((E, width), dtype=torch.float32, device=msg.device)
    for t in (grad_agg, agg_max, num_selected):
        if t is not None and (not t.is_contiguous() or t.shape[-1] != width):
            raise ValueError("grad_agg / agg_max / num_selected must be contiguous [V, width]")
    _lib.check(
        lib.tfgnn_edge_aggregate_backward(
            E, width, _ptr(msg), ld, _ptr(msg_row), _ptr(target), _ptr(edge_weight), _ptr(node_scale), act_id(pre_act),
            int(reduce), _ptr(grad_agg), _ptr(agg_max), _ptr(num_selected), int(phase), _ptr(out), _stream(),
        )
    )
    return out


def transpose_batched(x: torch.Tensor) -> torch.Tensor:
    """[B, R, C] -> [B, C, R] (contiguous copy); a 2-D input is one batch."""
    lib = _lib.load()
    _require_dev(x, torch.float32, "x")
    x = x.contiguous()
    squeeze = x.dim() == 2
    if squeeze:
        x = x.unsqueeze(0)
    B, R, C = x.shape
    out = torch.empty((B, C, R), dtype=torch.float32, device=x.device)
    _lib.check(lib.tfgnn_transpose_batched(_ptr(x), B, R, C, _ptr(out), _stream()))
    return out[0] if squeeze else out


@_writes_out
def gemm_grouped_rows(a, group_off_dev, group_off_host, b_stack, *, trans_b=False, act=ACT_NONE, out=None, act_grad=None):
    """out[rows g] = act(a[rows g] @ op(b_stack[g])) for row groups [off[g], off[g+1]).
    b_stack: [G, K, N] (or [G, N, K] with trans_b).  act_grad = (activation, saved [rows, N]): the result times act'(saved),
    in the product's epilogue where the active kernel has one (tfgnn_gemm_grouped_rows_grad), by a separate pass otherwise."""
    lib = _lib.load()
    a, lda = _rowmajor(a, "a")
    G = b_stack.shape[0]
    K = a.shape[1]
    N = b_stack.shape[1] if trans_b else b_stack.shape[2]
    b_stack = b_stack.contiguous()
    if (not trans_b and a.shape[0] >= 4096 * max(G, 1) // 8 and K >= 64 and K % 4 == 0
            and (N % 320 == 0 or N % 128 == 0) and get_gemm_mode() != GEMM_FP32):
        # as in gemm(): the split-operand kernel stages K-contiguous weights fastest; [G, K, N] -> [G, N, K] once
        b_stack, trans_b = transpose_batched(b_stack), True
    if out is None:
        out = torch.empty((a.shape[0], N), dtype=torch.float32, device=a.device)
    out2, ldc = _rowmajor(out, "out")
    max_rows = max((group_off_host[i + 1] - group_off_host[i] for i in range(G)), default=0)
    if act_grad is not None:
        if act_id(act) != act_id(ACT_NONE):
            raise ValueError("gemm_grouped_rows: act and act_grad exclude each other")
        saved, ld_saved = _rowmajor(act_grad[1], "saved")
        if tuple(saved.shape) != (a.shape[0], N):
            raise ValueError(f"saved must be [{a.shape[0]},{N}]")
        rc = lib.tfgnn_gemm_grouped_rows_grad(
            int(trans_b), G, _ptr(group_off_dev), max_rows, N, K, _ptr(a), lda, _ptr(b_stack), b_stack.stride(1),
            b_stack.stride(0), _ptr(out), ldc, act_id(act_grad[0]), _ptr(saved), ld_saved, _stream())
        if rc == 0:
            return out
        if rc != -4:
            _lib.check(rc)
        res = gemm_grouped_rows(a, group_off_dev, group_off_host, b_stack, trans_b=trans_b, out=out)
        return activation_backward(act_grad[0], res, act_grad[1])
    _lib.check(
        lib.tfgnn_gemm_grouped_rows(
            int(trans_b), G, _ptr(group_off_dev), max_rows, N, K, _ptr(a), lda, _ptr(b_stack), b_stack.stride(1),
            b_stack.stride(0), _ptr(out), ldc, act_id(act), _stream(),
        )
    )
    return out


@_writes_out
def gemm_grouped_k(a, b, group_off_dev, group_off_host, num_groups, out=None):
    """out[g] = a[rows g]^T @ b[rows g]  ->  [G, M, N]."""
    lib = _lib.load()
    a, lda = _rowmajor(a, "a")
    b, ldb = _rowmajor(b, "b")
    M, N = a.shape[1], b.shape[1]
    if out is None:
        out = torch.empty((num_groups, M, N), dtype=torch.float32, device=a.device)
    max_rows = max((group_off_host[i + 1] - group_off_host[i] for i in range(num_groups)), default=0)
    ws_bytes = lib.tfgnn_gemm_grouped_k_workspace_bytes(num_groups, max_rows, M, N)
    ws = _workspace(a.device, ws_bytes) if ws_bytes else None
    _lib.check(
        lib.tfgnn_gemm_grouped_k(
            num_groups, _ptr(group_off_dev), max_rows, M, N, _ptr(a), lda, _ptr(b), ldb, _ptr(out), out.stride(1),
            out.stride(0), _ptr(ws), ws.numel() if ws is not None else 0, _stream(),
        )
    )
    return out


def sigmoid_ce_metrics(logits: torch.Tensor, labels: torch.Tensor, need_grad: bool = True):
    """NodeMulticlassTask._fast_task_metrics (tf2_gnn/models/node_multiclass_task.py:62-70) ->
    (metrics [2] = (loss, micro-F1) on the device, counts [3] int64 = (tp, fp, fn), d loss / d logits or None)."""
    lib = _lib.load()
    _require_dev(logits, torch.float32, "logits")
    _require_dev(labels, torch.float32, "labels")
    logits, ldx = _rowmajor(logits, "logits")
    labels, ldz = _rowmajor(labels, "labels")
    if logits.shape != labels.shape:
        raise ValueError(f"logits {tuple(logits.shape)} and labels {tuple(labels.shape)} differ in shape")
    V, C = logits.shape
    metrics = torch.empty(2, dtype=torch.float32, device=logits.device)
    counts = torch.empty(3, dtype=torch.int64, device=logits.device)
    grad = torch.empty((V, C), dtype=torch.float32, device=logits.device) if need_grad else None
    nbytes = lib.tfgnn_task_metrics_workspace_bytes()
    ws = _workspace(logits.device, nbytes)
    _lib.check(lib.tfgnn_sigmoid_ce_metrics(_ptr(logits), ldx, _ptr(labels), ldz, V, C, _ptr(metrics), _ptr(counts), _ptr(grad),
                                            _ptr(ws), ws.numel(), _stream()))
    return metrics, counts, grad


def regression_metrics(pred: torch.Tensor, target: torch.Tensor, need_grad: bool = True):
    """tf.losses.mean_squared_error / mean_absolute_error of per-graph outputs
    (tf2_gnn/models/graph_regression_task.py:157-158) -> (metrics [2] = (mse, mae), d mse / d pred or None)."""
    lib = _lib.load()
    _require_dev(pred, torch.float32, "pred")
    _require_dev(target, torch.float32, "target")
    pred = pred.contiguous().view(-1)
    target = target.contiguous().view(-1)
    if pred.shape != target.shape:
        raise ValueError(f"pred {tuple(pred.shape)} and target {tuple(target.shape)} differ in shape")
    metrics = torch.empty(2, dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred) if need_grad else None
    nbytes = lib.tfgnn_task_metrics_workspace_bytes()
    ws = _workspace(pred.device, nbytes)
    _lib.check(lib.tfgnn_regression_metrics(_ptr(pred), _ptr(target), pred.numel(), _ptr(metrics), _ptr(grad), _ptr(ws),
                                            ws.numel(), _stream()))
    return metrics, grad


# ---- split-operand ("f16x2") products: include/tfgnn.h tfgnn_sp_*, csrc/gemm_sp.hip ------------------------------
class SplitOperand:
    """An fp32 matrix [rows, cols] in the SP16 operand format: ``data`` uint8 [rows, 4 * cols] (per row and 16 columns
    one 64-byte granule [16 x fp16 h | 16 x fp16 l]) and ``inv_scale`` fp32 [rows, cols // scale_block] (2^-e)."""

    __slots__ = ("data", "inv_scale", "rows", "cols", "scale_block")

    def __init__(self, data, inv_scale, rows, cols, scale_block):
        self.data, self.inv_scale, self.rows, self.cols, self.scale_block = data, inv_scale, rows, cols, scale_block

    def synced(self) -> "SplitOperand":
        """self, after every deferred small pass has been launched: an operand made with ``defer=True`` is written by the
        next merged launch, which only a following LIBRARY call triggers - read ``data`` / ``inv_scale`` with torch (or hand
        them to another library) through this."""
        aux_flush()
        return self


def sp_split_rows(x: torch.Tensor, *, scale_block: int = 0, segments=None, fixed_inv_scale: Optional[torch.Tensor] = None,
                  out: Optional[SplitOperand] = None, defer: bool = False) -> SplitOperand:
    """SP16 form of the rows of ``x`` [R, C] (unit inner stride).  ``segments = (seg_len, seg_stride, cols)``: row r is
    assembled from cols / seg_len pieces x.data[r * ld + j * seg_stride : ... + seg_len] (e.g. row d of
    [W_0[d, :] | W_1[d, :] | ...] from stacked kernels [L, D, H]: x = W[0], segments = (H, D * H, L * H))."""
    lib = _lib.load()
    _require_dev(x, torch.float32, "x")
    x, ld = _rowmajor(x, "x")
    rows = x.shape[0]
    if segments is None:
        seg_len, seg_stride, cols = 0, 0, x.shape[1]
    else:
        seg_len, seg_stride, cols = (int(v) for v in segments)
    sb = int(scale_block) if scale_block and scale_block > 0 else cols
    if out is None:
        data = torch.empty((rows, cols * 4), dtype=torch.uint8, device=x.device)
        if fixed_inv_scale is not None:
            out = SplitOperand(data, fixed_inv_scale, rows, cols, 0)  # one scale for the whole tensor
        else:
            out = SplitOperand(data, torch.empty((rows, cols // sb), dtype=torch.float32, device=x.device), rows, cols, sb)
    if defer and rows > 0 and aux_enabled():  # a job of the next merged small-pass launch (weights: nothing but launch latency)
        job = _lib.AuxJob()
        _lib.check(lib.tfgnn_sp_split_rows_job(_ptr(x), ld, seg_len, seg_stride, rows, cols, sb, _ptr(out.data), out.data.stride(0),
                                               None if fixed_inv_scale is not None else _ptr(out.inv_scale), _ptr(fixed_inv_scale),
                                               ctypes.byref(job)))
        aux_defer(job, keep=(x, out.data, out.inv_scale, fixed_inv_scale))
        return out
    _lib.check(lib.tfgnn_sp_split_rows(_ptr(x), ld, seg_len, seg_stride, rows, cols, sb, _ptr(out.data), out.data.stride(0),
                                       None if fixed_inv_scale is not None else _ptr(out.inv_scale), _ptr(fixed_inv_scale),
                                       _stream()))
    return out


_sp_rows_memo = {}  # id(tensor) -> (weakref, version, SplitOperand): split forms written by the kernel that produced the tensor


def _remember_split_rows(t: torch.Tensor, op: SplitOperand) -> None:
    import weakref

    for k in [k for k, (ref, _, _) in _sp_rows_memo.items() if ref() is None]:  # a handful of entries: layer inputs of one step
        del _sp_rows_memo[k]
    _sp_rows_memo[id(t)] = (weakref.ref(t), t._version, op)


def sp_rows_of(x: torch.Tensor) -> SplitOperand:
    """SP16 rows (one scale per row) of ``x``: the form its producer already wrote (dropout_forward in f16x2 mode), when that
    is still valid for this very tensor object and version, else a split pass."""
    hit = _sp_rows_memo.get(id(x))
    if hit is not None and hit[0]() is x and hit[1] == x._version:
        return hit[2]
    op = sp_split_rows(x)
    _remember_split_rows(x, op)  # forward and backward products of a step share it
    return op


def sp_split_cols(w: torch.Tensor, defer: bool = False, out: Optional[SplitOperand] = None) -> SplitOperand:
    """SP16 form of w^T for a row-major [K, N] matrix (a Keras kernel): rows = N, cols = K, one scale per row.
    defer: as a job of the next merged small-pass launch (``aux_defer``).  out: write into this operand (N rows of K columns:
    a slice of a stack of per-group operands)."""
    lib = _lib.load()
    _require_dev(w, torch.float32, "w")
    w, ld = _rowmajor(w, "w")
    K, N = w.shape
    if out is not None:
        if out.rows != N or out.cols != K:
            raise ValueError("sp_split_cols: out must hold N rows of K columns")
        data, inv = out.data, out.inv_scale
    else:
        data = torch.empty((N, K * 4), dtype=torch.uint8, device=w.device)
        inv = torch.empty((N, 1), dtype=torch.float32, device=w.device)
    if defer and aux_enabled():
        two_pass = lib.tfgnn_sp_split_cols_two_pass_bytes(K, N)
        if two_pass:
            # long K (the stacked kernels of many relations): the column maxima as a pass of their own in THIS merged launch, the
            # conversion - which then reads every byte once instead of once per K slice - in the next one (the chained job is
            # deferred as urgent: ``_stream()`` launches it before the consumer's kernel)
            cm = torch.empty(two_pass, dtype=torch.uint8, device=w.device)
            mjob, sjob = _lib.AuxJob(), _lib.AuxJob()
            _lib.check(lib.tfgnn_sp_split_cols_jobs(_ptr(w), ld, K, N, _ptr(data), data.stride(0), _ptr(inv), _ptr(cm), two_pass,
                                                    ctypes.byref(mjob), ctypes.byref(sjob)))
            aux_defer(mjob, keep=(w, cm), then=lambda: aux_defer(sjob, keep=(w, data, inv, cm)))
            return SplitOperand(data, inv, N, K, K)
        job = _lib.AuxJob()
        _lib.check(lib.tfgnn_sp_split_cols_job(_ptr(w), ld, K, N, _ptr(data), data.stride(0), _ptr(inv), ctypes.byref(job)))
        aux_defer(job, keep=(w, data, inv))
    else:
        _lib.check(lib.tfgnn_sp_split_cols(_ptr(w), ld, K, N, _ptr(data), data.stride(0), _ptr(inv), _stream()))
    return SplitOperand(data, inv, N, K, K)


def _tile_kmask(t, M):
    if t is not None and (t.dtype != torch.uint8 or not t.is_cuda or t.numel() < (M + 127) // 128 or not t.is_contiguous()):
        raise ValueError("tile_kmask must be a contiguous uint8 device tensor with one byte per 128-row tile")
    return t


def _row_map(t, M):
    if t is not None and (t.dtype != torch.int32 or not t.is_cuda or t.numel() != M or not t.is_contiguous()):
        raise ValueError("row_map must be a contiguous int32 device tensor with one entry per product row")
    return t


_SPLITK_WS = {}  # "ws": the workspace tensor of the in-launch K split (tfgnn_sp_gemm_nt_set_splitk_workspace), once per process
_SPLITK_BYTES = 80 << 20


def _ensure_splitk_workspace(device, rows: int) -> None:
    """Products over few row tiles (a batch of some thousand nodes) split K inside their launch so that more than one
    workgroup per tile works (include/tfgnn.h): the workspace is registered the first time such a product is issued - one
    per process, on that product's device (one process drives one GPU).  TFGNN_NT_SPLITK=0 keeps the unsplit product."""
    if rows > 56000 or "done" in _SPLITK_WS or capturing():  # (few row tiles: K split; masked products up to ~430 tiles: helpers)
        return
    _SPLITK_WS["done"] = True
    if env("TFGNN_NT_SPLITK", "1") == "0":
        return
    ws = torch.empty(_SPLITK_BYTES, dtype=torch.uint8, device=device)
    aux_flush()
    _lib.check(_lib.load().tfgnn_sp_gemm_nt_set_splitk_workspace(_ptr(ws), _SPLITK_BYTES))
    _SPLITK_WS["ws"] = ws


def sp_gemm_nt_splitk(enable: Optional[bool] = None):
    """-> (splits can happen, a reducer ever timed out, products launched with a split so far).  enable: switch the in-launch
    K split of products over few row tiles on / off (include/tfgnn.h tfgnn_sp_gemm_nt_set_splitk_workspace)."""
    t, n = ctypes.c_int(0), ctypes.c_int64(0)
    on = _lib.load().tfgnn_sp_gemm_nt_splitk_status(-1 if enable is None else int(bool(enable)), ctypes.byref(t), ctypes.byref(n))
    return bool(on), bool(t.value), int(n.value)


def sp_tile_width(N: int) -> int:
    """Column tile of the split-operand NT product for N output columns (gemm_sp.hip sp_tile_width); 0: unsupported."""
    return 320 if N % 320 == 0 else (256 if N % 256 == 0 else (128 if N % 128 == 0 else 0))


class RowGroups:
    """Consecutive row groups (rows of group g: [offsets[g], offsets[g + 1])) cut into the row tiles of the grouped product
    (tfgnn_sp_gemm_nt_grouped d_tile_table): int32 [tiles, 4] = (first row, rows, group, 0) on the device, built once per
    grouping (a batch's non-empty (source, type) rows: cached on its Graph)."""

    def __init__(self, offsets_host, device):
        import numpy as np

        off = [int(o) for o in offsets_host]
        rows = []
        for gi in range(len(off) - 1):
            for r0 in range(off[gi], off[gi + 1], 128):
                rows.append((r0, min(128, off[gi + 1] - r0), gi, 0))
        self.offsets = off
        self.num_groups = len(off) - 1
        self.num_rows = off[-1]
        self.num_tiles = len(rows)
        tab = np.asarray(rows if rows else [(0, 0, 0, 0)], dtype=np.int32)
        self.table = torch.from_numpy(tab).to(device)
        self._tn = None

    def tn_tables(self):
        """-> (K ranges int32 [S, 2] = (first row, rows <= TN_GROUPED_MAX_CHUNK), first range of every group int32 [G + 1], S):
        the row groups cut into the K ranges of the grouped weight-gradient product (tfgnn_sp_gemm_tn_grouped)"""
        if self._tn is None:
            import numpy as np

            ranges, first = [], [0]
            for gi in range(self.num_groups):
                r0, r1 = self.offsets[gi], self.offsets[gi + 1]
                n = -(-(r1 - r0) // TN_GROUPED_MAX_CHUNK)
                if n:
                    chunk = -(-(-(-(r1 - r0) // n)) // 16) * 16  # equal shares, whole k16 steps
                    for r in range(r0, r1, chunk):
                        ranges.append((r, min(chunk, r1 - r)))
                first.append(len(ranges))
            dev = self.table.device
            self._tn = (torch.from_numpy(np.asarray(ranges if ranges else [(0, 0)], dtype=np.int32)).to(dev),
                        torch.from_numpy(np.asarray(first, dtype=np.int32)).to(dev), len(ranges))
        return self._tn


def sp_gemm_nt_grouped(a: SplitOperand, b: SplitOperand, groups: RowGroups, *, a_rows=None, act=ACT_NONE, act_grad=None,
                       want_fp32: bool = True, want_split: bool = False, b_column_blocks: bool = False):
    """Per-group products on split operands (tfgnn_sp_gemm_nt_grouped): row r of group g -> epilogue(a_row(r) @ b_g^T).
    b: ``groups.num_groups`` stacked [N, K] operands with one scale per row ([G * N, K]), or - b_column_blocks - ONE operand
    [N, G * K] whose column block g is group g's (``sp_split_cols`` of G stacked Keras kernels [G * K, N]: one launch; the row
    scales are shared by the groups).  a_rows (int32 [rows]): row r reads a[a_rows[r]] (a then has any number of rows).
    act_grad = (name, saved [rows, N]).  -> (fp32 [rows, N] | None, SplitOperand | None)"""
    lib = _lib.load()
    M, K = groups.num_rows, a.cols
    G = max(groups.num_groups, 1)
    if b_column_blocks:
        if b.cols != G * K or b.scale_block != b.cols:
            raise ValueError("sp_gemm_nt_grouped: b must be [N, G * K] with one scale per row")
        N, stride_b, stride_scale = b.rows, 4 * K, 0
    else:
        if b.cols != K or b.rows % G or b.scale_block != K:
            raise ValueError("sp_gemm_nt_grouped: b must stack one [N, K] operand per group with one scale per row")
        N = b.rows // G
        stride_b, stride_scale = N * b.data.stride(0), N
    if a_rows is None and a.rows != M:
        raise ValueError(f"sp_gemm_nt_grouped: {a.rows} operand rows for {M} grouped rows")
    if a_rows is not None:
        a_rows = _row_map(a_rows, M)
    dev = a.data.device
    out = torch.empty((M, N), dtype=torch.float32, device=dev) if want_fp32 else None
    op = None
    if want_split:
        bn = sp_tile_width(N)
        op = SplitOperand(torch.empty((M, N * 4), dtype=torch.uint8, device=dev),
                          torch.empty((M, max(1, N // max(bn, 1))), dtype=torch.float32, device=dev), M, N, bn if bn else N)
    if M == 0:
        return out, op
    _, act_grad, _, _ = _native_epilogue(None, act_grad, None, 1.0)
    act_name, saved = act_grad if act_grad is not None else (None, None)
    _lib.check(
        lib.tfgnn_sp_gemm_nt_grouped(
            M, N, K, _ptr(a.data), a.data.stride(0), _ptr(a.inv_scale), a.scale_block if a.scale_block else -1, _ptr(a_rows), a.rows,
            _ptr(groups.table), groups.num_tiles, G, _ptr(b.data), b.data.stride(0), stride_b, _ptr(b.inv_scale), stride_scale,
            _ptr(out), N, None, act_id(act), None, 0, act_id(act_name), _ptr(saved), saved.stride(0) if saved is not None else 0,
            _ptr(op.data) if op is not None else None, op.data.stride(0) if op is not None else 0,
            _ptr(op.inv_scale) if op is not None else None, _stream(),
        )
    )
    if out is not None and op is not None:
        _remember_split_rows(out, op)
    return out, op


TN_WIDE_MAX_ROWS = 512 * 2016  # rows one launch of the two-factor TN product covers (512 K ranges)
TN_GROUPED_MAX_RANGES = 512  # K ranges one launch of the grouped two-factor product takes (tfgnn_sp_gemm_tn_grouped)
TN_GROUPED_MAX_CHUNK = 2016  # rows of a K range of the wide-range product (gemm_sp.hip SP_TN_BSC_MAX_CHUNK)


def sp_gemm_tn_grouped(a: SplitOperand, b: SplitOperand, groups: RowGroups, out: torch.Tensor, transposed: bool = False) -> torch.Tensor:
    """out[g] = a_g^T b_g over the rows of group g, all groups in ONE launch of the wide-range product (tfgnn_sp_gemm_tn_grouped):
    ``a`` [rows, M] with per-(row, block) scales, ``b`` [rows, N] with one scale per row, out [G, M, N] (transposed: [G, N, M],
    element (m, n) of product g at out[g, n, m])."""
    lib = _lib.load()
    if a.scale_block <= 0 or b.scale_block != b.cols or a.rows != b.rows or a.rows != groups.num_rows:
        raise ValueError("sp_gemm_tn_grouped: operands must share the grouped rows; b needs one scale per row")
    M, N, G = a.cols, b.cols, groups.num_groups
    if not out.is_contiguous() or out.numel() != G * M * N:
        raise ValueError("sp_gemm_tn_grouped: out must be contiguous [G, M, N]")
    tab = groups.tn_tables()
    if tab[2] > TN_GROUPED_MAX_RANGES:
        raise ValueError(f"sp_gemm_tn_grouped: {tab[2]} K ranges, at most {TN_GROUPED_MAX_RANGES} per launch (callers take another route)")
    nblk = a.cols // a.scale_block
    ws_bytes = ((nblk * 512 * 4 + 255) & ~255) + tab[2] * ((M + 127) // 128 * 128) * N * 4 + 256
    ws = _workspace(a.data.device, ws_bytes)
    off = (-ws.data_ptr()) % 256
    sr, sc = (1, M) if transposed else (N, 1)
    _lib.check(lib.tfgnn_sp_gemm_tn_grouped(M, N, _ptr(a.data), a.data.stride(0), _ptr(a.inv_scale), a.cols, a.scale_block, _ptr(b.data),
                                            b.data.stride(0), _ptr(b.inv_scale), G, _ptr(tab[1]), tab[2], _ptr(tab[0]), _ptr(out), M * N, sr, sc,
                                            ctypes.c_void_p(ws.data_ptr() + off), ws.numel() - off, _stream()))
    return out


def sp_gather_rows(a: SplitOperand, index: torch.Tensor) -> SplitOperand:
    """rows ``index`` of a split operand, with their scales (tfgnn_sp_gather_rows)."""
    lib = _lib.load()
    _require_dev(index, torch.int32, "index")
    n = int(index.numel())
    per = a.inv_scale.shape[1] if a.inv_scale.dim() == 2 else 1
    out = SplitOperand(torch.empty((n, a.cols * 4), dtype=torch.uint8, device=a.data.device),
                       torch.empty((n, per), dtype=torch.float32, device=a.data.device), n, a.cols, a.scale_block)
    _lib.check(lib.tfgnn_sp_gather_rows(_ptr(a.data), a.data.stride(0), _ptr(a.inv_scale), per, _ptr(index.contiguous()), n, a.rows, a.cols,
                                        _ptr(out.data), out.data.stride(0), _ptr(out.inv_scale), _stream()))
    return out


@_writes_out
def sp_gemm_nt(a: SplitOperand, b: SplitOperand, *, bias=None, act=ACT_NONE, out=None, accumulate=False, out_mul=None,
               act_grad=None, dropout=None, saved_scale: float = 1.0, tile_kmask=None, row_map=None, a_rows=None) -> torch.Tensor:
    """out [M, N] = epilogue(a [M, K] @ b [N, K]^T) from SP16 operands (tfgnn_sp_gemm_nt / tfgnn_sp_gemm_nt_dropout).
    tile_kmask (uint8 [ceil(M / 128)]): bit b = scale block b of ``a`` holds non-zeros in that row tile, the other blocks are
    skipped; row_map (int32 [M]): product row r is written at out[row_map[r]] (Graph pattern order, graph_gather_sp).
    dropout = (rate, seed): the result times the mask ``dropout_forward`` draws for that seed, applied in the epilogue
    (forward: the next layer's input dropout; gradient product: the recomputed forward mask).  saved_scale: the derivative
    of ``act_grad`` is taken at saved * saved_scale (saved is a dropped activation).  a_rows (int32 [M]): product row r reads
    row a_rows[r] of ``a`` (tfgnn_sp_gemm_nt_rows: the by-source pattern order of the input-gradient product)."""
    lib = _lib.load()
    M, K, N = a.rows, a.cols, b.rows
    if b.cols != K:
        raise ValueError(f"sp_gemm_nt: inner dimensions differ ({K} vs {b.cols})")
    if b.scale_block != K:
        raise ValueError("sp_gemm_nt: the right operand must carry one scale per row")
    if out is None:
        if accumulate:
            raise ValueError("accumulate=True needs out")
        out = torch.empty((M, N), dtype=torch.float32, device=a.data.device)
    out2, ldc = _rowmajor(out, "out")
    if out2 is not out or tuple(out.shape) != (M, N):
        raise ValueError(f"out must be [{M},{N}] with unit inner stride")
    out_mul, act_grad, dropout, saved_scale = _native_epilogue(out_mul, act_grad, dropout, saved_scale)
    act_name, saved = act_grad if act_grad is not None else (None, None)
    if bias is not None:
        bias = bias.contiguous()
    rate, seed = dropout if dropout is not None else (0.0, 0)
    _ensure_splitk_workspace(a.data.device, M)
    _lib.check(
        lib.tfgnn_sp_gemm_nt_rows(
            M, N, K, _ptr(a.data), a.data.stride(0), _ptr(a.inv_scale), a.scale_block if a.scale_block else -1,
            _ptr(_row_map(a_rows, M)), _ptr(b.data), b.data.stride(0),
            _ptr(b.inv_scale), _ptr(out), ldc, _ptr(bias), act_id(act), int(accumulate), _ptr(out_mul),
            out_mul.stride(0) if out_mul is not None else 0, act_id(act_name), _ptr(saved),
            saved.stride(0) if saved is not None else 0, float(saved_scale), None, 0, None, float(rate),
            int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(_tile_kmask(tile_kmask, M)), _ptr(_row_map(row_map, M)), _stream(),
        )
    )
    return out


def sp_gemm_nt_split(a: SplitOperand, b: SplitOperand, *, bias=None, act=ACT_NONE, out_mul=None, act_grad=None,
                     want_fp32: bool = True, dropout=None, saved_scale: float = 1.0, tile_kmask=None, row_map=None, a_rows=None):
    """As sp_gemm_nt, with the result ALSO (or only: want_fp32=False) written as an SP16 operand with one scale per row
    by the product's epilogue (tfgnn_sp_gemm_nt_sp): the next product's operand without a split pass.  N must be one
    column tile (128, 256 or 320).  -> (fp32 [M, N] | None, SplitOperand); the fp32 tensor remembers its split form
    (``sp_rows_of``)."""
    lib = _lib.load()
    M, K, N = a.rows, a.cols, b.rows
    if b.cols != K:
        raise ValueError(f"sp_gemm_nt_split: inner dimensions differ ({K} vs {b.cols})")
    if b.scale_block != K:
        raise ValueError("sp_gemm_nt_split: the right operand must carry one scale per row")
    dev = a.data.device
    out = torch.empty((M, N), dtype=torch.float32, device=dev) if want_fp32 else None
    bn = sp_tile_width(N)  # the result carries one scale per row and COLUMN TILE (N = 512: blocks of 256 columns)
    op = SplitOperand(torch.empty((M, N * 4), dtype=torch.uint8, device=dev),
                      torch.empty((M, max(1, N // max(bn, 1))), dtype=torch.float32, device=dev), M, N, bn if bn else N)
    out_mul, act_grad, dropout, saved_scale = _native_epilogue(out_mul, act_grad, dropout, saved_scale)
    act_name, saved = act_grad if act_grad is not None else (None, None)
    if bias is not None:
        bias = bias.contiguous()
    rate, seed = dropout if dropout is not None else (0.0, 0)
    _ensure_splitk_workspace(dev, M)
    _lib.check(
        lib.tfgnn_sp_gemm_nt_rows(
            M, N, K, _ptr(a.data), a.data.stride(0), _ptr(a.inv_scale), a.scale_block if a.scale_block else -1,
            _ptr(_row_map(a_rows, M)), _ptr(b.data),
            b.data.stride(0), _ptr(b.inv_scale), _ptr(out), N, _ptr(bias), act_id(act), 0, _ptr(out_mul),
            out_mul.stride(0) if out_mul is not None else 0, act_id(act_name), _ptr(saved),
            saved.stride(0) if saved is not None else 0, float(saved_scale), _ptr(op.data), op.data.stride(0), _ptr(op.inv_scale),
            float(rate), int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(_tile_kmask(tile_kmask, M)), _ptr(_row_map(row_map, M)), _stream(),
        )
    )
    if out is not None:
        _remember_split_rows(out, op)
    return out, op


def split_rows_remembered(x: torch.Tensor) -> SplitOperand:
    """Split ``x`` now and remember the operand for ``sp_rows_of(x)`` (an input pipeline preparing the node features of a
    batch for the first product, e.g. on the stream that also buckets the batch's edges)."""
    op = sp_split_rows(x)
    _remember_split_rows(x, op)
    return op


def graph_gather_sp(graph: "Graph", view: int, inp: torch.Tensor, *, col=None, edge_weight=None, row_scale=None,
                    fixed_inv_scale: Optional[torch.Tensor] = None, rows_per_operand_row: int = 1,
                    defer_combine: bool = False) -> SplitOperand:
    """graph_gather (plain sums) with the result written as an SP16 operand (tfgnn_graph_gather_reduce_sp).
    ``rows_per_operand_row`` = L folds the rows (v, l) of a typed view into the [V, L * width] operand with one scale
    block per edge type.  Compact views: one operand row (and scale) per non-empty bucket."""
    lib = _lib.load()
    _require_dev(inp, torch.float32, "inp")
    typed = view in (VIEW_BY_DST_TYPED, VIEW_BY_SRC_TYPED, VIEW_BY_DST_TYPED_PATTERN)
    if view in (VIEW_BY_DST_TYPED_COMPACT, VIEW_BY_SRC_TYPED_COMPACT):
        if rows_per_operand_row != 1:
            raise ValueError("graph_gather_sp: compact rows are operand rows (rows_per_operand_row = 1)")
        num_rows = int(graph.nonempty_offsets(view == VIEW_BY_SRC_TYPED_COMPACT)[-1])
    else:
        num_rows = graph.num_nodes * (graph.num_edge_types if typed else 1)
    if view == VIEW_BY_DST_TYPED_PATTERN and graph.num_edge_types > 8:
        raise ValueError("graph_gather_sp: the pattern order exists for at most 8 edge types")
    inp, ld_in = _rowmajor(inp, "inp")
    width = inp.shape[1]
    R = int(rows_per_operand_row)
    if num_rows % R:
        raise ValueError("rows_per_operand_row must divide the number of rows of the view")
    data = torch.empty((num_rows // R, R * width * 4), dtype=torch.uint8, device=inp.device)
    inv = None
    if fixed_inv_scale is None:
        inv = torch.empty((num_rows // R, R), dtype=torch.float32, device=inp.device)
    graph.ensure(_VIEW_PARTS[view])
    ws_bytes = lib.tfgnn_graph_gather_workspace_bytes(graph._h, view, width)
    ws = _workspace(inp.device, ws_bytes) if ws_bytes else None
    if defer_combine and aux_enabled():
        # the combine pass of the long buckets rides with whatever other small pass precedes the consumer (a weight split): the
        # operand is complete once the next library call has been issued (or after aux_flush())
        job = _lib.AuxJob()
        _lib.check(
            lib.tfgnn_graph_gather_reduce_sp_deferred(
                graph._h, view, _ptr(col), _ptr(edge_weight), _ptr(row_scale), _ptr(inp), ld_in, width, _ptr(data), width * 4,
                _ptr(inv), _ptr(fixed_inv_scale), _ptr(ws), ws.numel() if ws is not None else 0, ctypes.byref(job), _stream(),
            )
        )
        aux_defer(job, keep=(graph, row_scale, ws, data, inv, fixed_inv_scale))
    else:
        _lib.check(
            lib.tfgnn_graph_gather_reduce_sp(
                graph._h, view, _ptr(col), _ptr(edge_weight), _ptr(row_scale), _ptr(inp), ld_in, width, _ptr(data), width * 4,
                _ptr(inv), _ptr(fixed_inv_scale), _ptr(ws), ws.numel() if ws is not None else 0, _stream(),
            )
        )
    if fixed_inv_scale is not None:
        return SplitOperand(data, fixed_inv_scale, num_rows // R, R * width, 0)  # scale_block 0: one scale for the tensor
    return SplitOperand(data, inv, num_rows // R, R * width, width)


# ---- layer-level entry points (include/tfgnn.h tfgnn_mp_forward / tfgnn_mp_backward, csrc/mp_layer.hip) --------------------------
def mp_entry_enabled() -> bool:
    """The aggregate-first layers drive ONE C call per pass (default) instead of the op-level calls it is made of
    (TFGNN_MP_ENTRY=0: the op-level route - same kernels, same arguments, bit-identical results)."""
    return env("TFGNN_MP_ENTRY", "1") != "0"


def _take_pending_jobs():
    """What ``aux_defer`` holds for the current stream, handed to a layer-level call that launches it with its own small passes.
    -> (ctypes array | None, count, callbacks to run after the call, keep-alive)"""
    with _AUX_LOCK:
        batch = list(_AUX_PENDING)
        del _AUX_PENDING[:]
        _AUX_URGENT[0] = False
    live = [j for j, _, _ in batch if j.kind != 0 and j.num_blocks != 0]
    arr = (_lib.AuxJob * len(live))(*live) if live else None
    return arr, len(live), [t for _, _, t in batch if t is not None], batch


def _fresh_weight_operand(w: torch.Tensor, kind: str, rows: int, cols: int):
    """-> (operand, stale): the cached split form of weight ``w`` or a new, still EMPTY one registered in the cache - the
    layer-level call that receives ``stale`` = True builds it in its merged small-pass launch."""
    made = []

    def build():
        made.append(True)
        return SplitOperand(torch.empty((rows, cols * 4), dtype=torch.uint8, device=w.device),
                            torch.empty((rows, 1), dtype=torch.float32, device=w.device), rows, cols, cols)

    return sp_weight_operand(w, kind, build), bool(made)


def mp_forward(graph: "Graph", view: int, x: torch.Tensor, W: torch.Tensor, *, row_scale=None, act=ACT_NONE, dropout=None,
               tile_kmask=None, row_map=None, want_split: bool = False, want_fp32: bool = True):
    """One aggregate-first message-passing layer forward in ONE library call (tfgnn_mp_forward): gather of the source rows per
    (node, type) bucket written as the split operand, W^T split when its cached form is stale, the small passes merged, the
    product with its epilogue.  W: stacked kernels [L, D, H].  -> (out [V, H] | None, split form of out | None)."""
    lib = _lib.load()
    _require_dev(x, torch.float32, "x")
    L, D, H = (int(v) for v in W.shape)
    V = graph.num_nodes
    if L != graph.num_edge_types or x.shape[1] != D or not W.is_contiguous():
        raise ValueError("mp_forward: W must be the contiguous [L, D, H] stack of the graph's edge types and of x's width")
    x, ldx = _rowmajor(x, "x")
    dev = x.device
    graph.ensure(_VIEW_PARTS[view])
    wt, stale = _fresh_weight_operand(W, "cols", H, L * D)
    agg = torch.empty((V, L * D * 4), dtype=torch.uint8, device=dev)
    agg_inv = torch.empty((V, L), dtype=torch.float32, device=dev)
    out = torch.empty((V, H), dtype=torch.float32, device=dev) if want_fp32 else None
    op = None
    if want_split:
        bn = sp_tile_width(H)
        op = SplitOperand(torch.empty((V, H * 4), dtype=torch.uint8, device=dev),
                          torch.empty((V, max(1, H // max(bn, 1))), dtype=torch.float32, device=dev), V, H, bn if bn else H)
    ws_bytes = lib.tfgnn_graph_gather_workspace_bytes(graph._h, view, D)
    ws = _workspace(dev, ws_bytes) if ws_bytes else None
    rate, seed = dropout if dropout is not None else (0.0, 0)
    _ensure_splitk_workspace(dev, V)
    stream = _stream()  # (launches what was deferred as urgent before this layer: its inputs may depend on it)
    jobs, njobs, after, keep = _take_pending_jobs()
    a = _lib.MpForwardArgs()
    a.struct_size = ctypes.sizeof(_lib.MpForwardArgs)
    a.kind = 0
    a.graph = graph._h
    a.view = int(view)
    a.x, a.ldx, a.in_dim, a.hidden_dim = x.data_ptr(), ldx, D, H
    a.row_scale = row_scale.data_ptr() if row_scale is not None else None
    a.w = W.data_ptr() if stale else None
    a.wt_sp, a.ld_wt_sp_bytes, a.wt_inv_scale = wt.data.data_ptr(), wt.data.stride(0), wt.inv_scale.data_ptr()
    a.agg_sp, a.agg_inv_scale = agg.data_ptr(), agg_inv.data_ptr()
    a.bias = None
    a.act = act_id(act)
    a.dropout_rate, a.dropout_seed = float(rate), int(seed) & 0xFFFFFFFFFFFFFFFF
    a.tile_kmask = _tile_kmask(tile_kmask, V).data_ptr() if tile_kmask is not None else None
    a.row_map = _row_map(row_map, V).data_ptr() if row_map is not None else None
    a.out, a.ld_out = (out.data_ptr(), H) if out is not None else (None, 0)
    if op is not None:
        a.out_sp, a.ld_out_sp_bytes, a.out_inv_scale = op.data.data_ptr(), op.data.stride(0), op.inv_scale.data_ptr()
    a.extra_jobs = ctypes.cast(jobs, ctypes.c_void_p) if jobs is not None else None
    a.num_extra_jobs = njobs
    a.workspace, a.workspace_bytes = (ws.data_ptr(), ws.numel()) if ws is not None else (None, 0)
    try:
        _lib.check(lib.tfgnn_mp_forward(ctypes.byref(a), stream))
    except Exception:
        if stale:  # the cache holds an operand this call was to fill: forget it
            notify_weights_changed(W)
        raise
    del keep
    for then in after:
        then()
    if out is not None and op is not None:
        _remember_split_rows(out, op)
    return out, op


def mp_backward(graph: "Graph", d_pre: torch.Tensor, W: torch.Tensor, x_sp: Optional[SplitOperand], *, edge_weight=None, out=None,
                accumulate: bool = False, out_mul=None, act_grad=None, want_split: bool = False, skip=None,
                need_weight_grad: bool = True):
    """The backward pass of that layer in ONE library call (tfgnn_mp_backward): the gather of ``d_pre`` over the by-source buckets
    as a split operand, the rows form of the kernels when stale, d(node states) = epilogue(G W^T) - ``out`` / ``accumulate`` /
    ``out_mul`` / ``act_grad`` as in ``sp_gemm_nt``; ``skip`` = dict(tile_kmask, a_rows, row_map) of the by-source pattern order -
    and the kernel gradients dW [L, D, H] = X^T G_l from the layer input's split form ``x_sp``.
    -> (dX [V, D], split form of dX | None, dW | None)."""
    lib = _lib.load()
    _require_dev(d_pre, torch.float32, "d_pre")
    L, D, H = (int(v) for v in W.shape)
    V = graph.num_nodes
    if L != graph.num_edge_types or d_pre.shape[1] != H or not W.is_contiguous():
        raise ValueError("mp_backward: W must be the contiguous [L, D, H] stack of the graph's edge types, d_pre [V, H]")
    d_pre, ldg = _rowmajor(d_pre, "d_pre")
    dev = d_pre.device
    graph.ensure(_VIEW_PARTS[VIEW_BY_SRC_TYPED])
    wh, stale = _fresh_weight_operand(W, "rows", D, L * H)
    g_sp = torch.empty((V, L * H * 4), dtype=torch.uint8, device=dev)
    g_inv = torch.empty((V, L), dtype=torch.float32, device=dev)
    if out is None:
        if accumulate:
            raise ValueError("accumulate=True needs out")
        out = torch.empty((V, D), dtype=torch.float32, device=dev)
    out2, ldc = _rowmajor(out, "out")
    if out2 is not out or tuple(out.shape) != (V, D):
        raise ValueError(f"out must be [{V},{D}] with unit inner stride")
    op = None
    if want_split:
        if accumulate:
            raise ValueError("mp_backward: the split result needs no accumulation")
        bn = sp_tile_width(D)
        op = SplitOperand(torch.empty((V, D * 4), dtype=torch.uint8, device=dev),
                          torch.empty((V, max(1, D // max(bn, 1))), dtype=torch.float32, device=dev), V, D, bn if bn else D)
    out_mul, act_grad, dropout, saved_scale = _native_epilogue(out_mul, act_grad, None, 1.0)
    act_name, saved = act_grad if act_grad is not None else (None, None)
    rate, seed = dropout if dropout is not None else (0.0, 0)
    dW = torch.empty_like(W) if need_weight_grad else None
    ws_bytes = lib.tfgnn_graph_gather_workspace_bytes(graph._h, VIEW_BY_SRC_TYPED, H)
    ws = _workspace(dev, ws_bytes) if ws_bytes else None
    tn_ws, tn_ptr, tn_len = None, None, 0
    if need_weight_grad:
        if x_sp is None or x_sp.scale_block != x_sp.cols or x_sp.rows != V or x_sp.cols != D:
            raise ValueError("mp_backward: the kernel gradients need the layer input as a split operand with one scale per row")
        tn_bytes = lib.tfgnn_sp_gemm_tn_workspace_bytes(L * H, D, V, L * H, H)
        if tn_bytes:
            # (its own buffer: the gather's workspace above is the shared one of this stream)
            tn_ws = torch.empty(tn_bytes + 256, dtype=torch.uint8, device=dev)
            off = (-tn_ws.data_ptr()) % 256
            tn_ptr, tn_len = tn_ws.data_ptr() + off, tn_ws.numel() - off
    _ensure_splitk_workspace(dev, V)
    stream = _stream()
    jobs, njobs, after, keep = _take_pending_jobs()
    skip = skip or {}
    a = _lib.MpBackwardArgs()
    a.struct_size = ctypes.sizeof(_lib.MpBackwardArgs)
    a.kind = 0
    a.graph = graph._h
    a.d_pre, a.ld_d_pre, a.in_dim, a.hidden_dim = d_pre.data_ptr(), ldg, D, H
    a.edge_weight = edge_weight.data_ptr() if edge_weight is not None else None
    a.w = W.data_ptr() if stale else None
    a.wh_sp, a.ld_wh_sp_bytes, a.wh_inv_scale = wh.data.data_ptr(), wh.data.stride(0), wh.inv_scale.data_ptr()
    a.g_sp, a.g_inv_scale = g_sp.data_ptr(), g_inv.data_ptr()
    a.dx, a.ld_dx, a.accumulate = out.data_ptr(), ldc, int(bool(accumulate))
    if out_mul is not None:
        a.mul, a.ld_mul = out_mul.data_ptr(), out_mul.stride(0)
    a.act_of_saved = act_id(act_name)
    if saved is not None:
        a.saved, a.ld_saved = saved.data_ptr(), saved.stride(0)
    a.saved_scale = float(saved_scale)
    a.dropout_rate, a.dropout_seed = float(rate), int(seed) & 0xFFFFFFFFFFFFFFFF
    if op is not None:
        a.dx_sp, a.ld_dx_sp_bytes, a.dx_inv_scale = op.data.data_ptr(), op.data.stride(0), op.inv_scale.data_ptr()
    for name, check in (("tile_kmask", _tile_kmask), ("a_rows", _row_map), ("row_map", _row_map)):
        t = skip.get(name)
        if t is not None:
            setattr(a, name, check(t, V).data_ptr())
    if dW is not None:
        a.dw = dW.data_ptr()
        a.x_sp, a.ld_x_sp_bytes, a.x_inv_scale = x_sp.data.data_ptr(), x_sp.data.stride(0), x_sp.inv_scale.data_ptr()
        a.tn_workspace, a.tn_workspace_bytes = tn_ptr, tn_len
    a.extra_jobs = ctypes.cast(jobs, ctypes.c_void_p) if jobs is not None else None
    a.num_extra_jobs = njobs
    a.workspace, a.workspace_bytes = (ws.data_ptr(), ws.numel()) if ws is not None else (None, 0)
    try:
        _lib.check(lib.tfgnn_mp_backward(ctypes.byref(a), stream))
    except Exception:
        if stale:
            notify_weights_changed(W)
        raise
    del keep, tn_ws
    for then in after:
        then()
    torch.autograd.graph.increment_version(out)
    if op is not None:
        _remember_split_rows(out, op)
    return out, op, dW


@_writes_out
def sp_gemm_tn(a: SplitOperand, b: SplitOperand, *, a_cols=None, b_cols=None, out: Optional[torch.Tensor] = None,
               scatter=None, accumulate: bool = False, defer_reduce: bool = False, wide: bool = False) -> torch.Tensor:
    """C[m, n] = sum_k a[k, a0 + m] * b[k, b0 + n] (tfgnn_sp_gemm_tn).  ``a`` carries one scale per (row, block),
    ``b`` one per row - what sp_split_rows / graph_gather_sp write.  ``a_cols`` / ``b_cols`` = (first column, count)
    select column ranges.  ``scatter`` = (group_rows, stride_group, stride_row, stride_col) writes element (m, n) at
    out.flatten()[(m // group_rows) * stride_group + (m % group_rows) * stride_row + n * stride_col]; default: row-major.
    wide: the two-factor form (tfgnn_sp_gemm_tn_wide) - each operand's row scales may spread over 2^22 within a K range,
    whatever their products do (rows that are un-normalised sums on both sides)."""
    lib = _lib.load()
    if a.scale_block <= 0 or b.scale_block != b.cols:
        raise ValueError("sp_gemm_tn: the left operand needs per-(row, block) scales, the right one one scale per row")
    if a.rows != b.rows:
        raise ValueError(f"sp_gemm_tn: K differs ({a.rows} vs {b.rows})")
    if wide and a.rows > TN_WIDE_MAX_ROWS:
        # the two-factor product takes at most 512 K ranges of 2016 rows per launch: longer operands (10^6 node rows) run as
        # consecutive row ranges, each adding into ``out``
        if out is None or defer_reduce:
            raise ValueError("sp_gemm_tn(wide=True) over more than %d rows needs out and no deferred reduction" % TN_WIDE_MAX_ROWS)
        for r0 in range(0, a.rows, TN_WIDE_MAX_ROWS):
            r1 = min(a.rows, r0 + TN_WIDE_MAX_ROWS)
            sp_gemm_tn(SplitOperand(a.data[r0:r1], a.inv_scale[r0:r1], r1 - r0, a.cols, a.scale_block),
                       SplitOperand(b.data[r0:r1], b.inv_scale[r0:r1], r1 - r0, b.cols, b.scale_block),
                       a_cols=a_cols, b_cols=b_cols, out=out, scatter=scatter, accumulate=accumulate or r0 > 0, wide=True)
        return out
    a0, M = a_cols if a_cols is not None else (0, a.cols)
    b0, N = b_cols if b_cols is not None else (0, b.cols)
    K = a.rows
    if out is None:
        if scatter is not None or accumulate:
            raise ValueError("scatter / accumulate need out")
        out = torch.empty((M, N), dtype=torch.float32, device=a.data.device)
    if not out.is_contiguous() or out.numel() != M * N:
        raise ValueError(f"out must be contiguous with {M * N} elements")
    gr, sg, sr, sc = scatter if scatter is not None else (M, 0, N, 1)
    ws_bytes = (lib.tfgnn_sp_gemm_tn_wide_workspace_bytes if wide else lib.tfgnn_sp_gemm_tn_workspace_bytes)(M, N, K, a.cols, a.scale_block)
    defer_reduce = bool(defer_reduce) and aux_enabled() and ws_bytes > 0
    if defer_reduce:
        # ``out`` is complete only after the next aux_flush(): a weight gradient is not read before the end of the backward pass,
        # so the split reductions of all layers share a launch.  The job owns its workspace until then.
        ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=a.data.device)
    else:
        ws = _workspace(a.data.device, ws_bytes + 256) if ws_bytes else None
    ws_ptr, ws_len = None, 0
    if ws is not None:
        off = (-ws.data_ptr()) % 256
        ws_ptr, ws_len = ctypes.c_void_p(ws.data_ptr() + off), ws.numel() - off
    if wide:
        args = (M, N, K, _ptr(a.data), a.data.stride(0), a0, _ptr(a.inv_scale), a.cols, a.scale_block, _ptr(b.data),
                b.data.stride(0), b0, _ptr(b.inv_scale), _ptr(out), gr, sg, sr, sc, int(accumulate), ws_ptr, ws_len)
        if defer_reduce:
            rjob = _lib.AuxJob()
            _lib.check(lib.tfgnn_sp_gemm_tn_wide(*args, ctypes.byref(rjob), _stream()))
            aux_defer(rjob, keep=(ws, out, a.data, a.inv_scale, b.data, b.inv_scale), urgent=False)
        else:
            _lib.check(lib.tfgnn_sp_gemm_tn_wide(*args, None, _stream()))
        return out
    if defer_reduce:
        args = (M, N, K, _ptr(a.data), a.data.stride(0), a0, _ptr(a.inv_scale), a.cols, a.scale_block, _ptr(b.data),
                b.data.stride(0), b0, _ptr(b.inv_scale), _ptr(out), gr, sg, sr, sc, int(accumulate), ws_ptr, ws_len)
        keep = (ws, out, a.data, a.inv_scale, b.data, b.inv_scale)
        rjob = _lib.AuxJob()
        if K <= 131072 and env("TFGNN_TN_CHAINED", "0") == "1":  # opt-in: measured slower (NOTEBOOK.md 4.7)
            # the whole product waits for the next merged launch: its factor pass rides there, the product follows it, the
            # reduction rides in a later one (a weight gradient is off the critical path of the backward pass)
            fjob = _lib.AuxJob()
            _lib.check(lib.tfgnn_sp_gemm_tn_jobs(*args, ctypes.byref(fjob), ctypes.byref(rjob)))

            def product():
                _lib.check(lib.tfgnn_sp_gemm_tn_phase(2, *args, _stream()))
                aux_defer(rjob, keep=keep, urgent=False)

            aux_defer(fjob, keep=keep, urgent=False, then=product)
            return out
        _lib.check(lib.tfgnn_sp_gemm_tn_deferred(1, *args, ctypes.byref(rjob), _stream()))
        aux_defer(rjob, keep=keep, urgent=False)
        return out
    _lib.check(
        lib.tfgnn_sp_gemm_tn(
            M, N, K, _ptr(a.data), a.data.stride(0), a0, _ptr(a.inv_scale), a.cols, a.scale_block, _ptr(b.data),
            b.data.stride(0), b0, _ptr(b.inv_scale), _ptr(out), gr, sg, sr, sc, int(accumulate), ws_ptr, ws_len, _stream(),
        )
    )
    return out


_aux_streams = {}
_aux_pending = []
def aux_stream(device) -> "torch.cuda.Stream":
    """The library's second stream of a device: small passes that are off the critical path (the factor and reduction
    passes of a weight-gradient product) run there beside the big kernels of the main stream."""
    key = (device.type, device.index)
    st = _aux_streams.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _aux_streams[key] = st
    return st


def join_aux_stream() -> None:
    """Make the current stream wait for everything handed to the second stream (call before the results - weight
    gradients - are read)."""
    cur = torch.cuda.current_stream()
    while _aux_pending:
        cur.wait_event(_aux_pending.pop())


class SpGemmTnOverlapped:
    """sp_gemm_tn in three calls, the small passes on the second stream:
         h = SpGemmTnOverlapped(a, b, out=..., scatter=...)   # factor pass: second stream, beside what the caller enqueues next
         ...                                                   # e.g. the input-gradient product that precedes it
         h.product()                                           # the product: current stream, after the factors
         h.finish()                                            # reduction + scatter: second stream, beside what follows
       ``join_aux_stream()`` before the result is read."""

    def __init__(self, a: "SplitOperand", b: "SplitOperand", *, out: torch.Tensor, scatter=None, accumulate: bool = False):
        if a.scale_block <= 0 or b.scale_block != b.cols or a.rows != b.rows:
            raise ValueError("SpGemmTnOverlapped: operands as for sp_gemm_tn")
        lib = _lib.load()
        self.M, self.N, self.K = a.cols, b.cols, a.rows
        if not out.is_contiguous() or out.numel() != self.M * self.N:
            raise ValueError(f"out must be contiguous with {self.M * self.N} elements")
        self.a, self.b, self.out, self.accumulate = a, b, out, accumulate
        self.scatter = scatter if scatter is not None else (self.M, 0, self.N, 1)
        ws_bytes = lib.tfgnn_sp_gemm_tn_workspace_bytes(self.M, self.N, self.K, a.cols, a.scale_block)
        self.ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=a.data.device)
        self.main = torch.cuda.current_stream()
        self.side = aux_stream(a.data.device)
        for t in (self.ws, out, a.inv_scale, b.inv_scale):
            t.record_stream(self.side)
        self.side.wait_stream(self.main)  # the scales are written
        with torch.cuda.stream(self.side):
            self._phase(1)
            self.ev_factors = self.side.record_event()

    def _phase(self, phases: int):
        a, b = self.a, self.b
        off = (-self.ws.data_ptr()) % 256
        gr, sg, sr, sc = self.scatter
        _lib.check(_lib.load().tfgnn_sp_gemm_tn_phase(
            phases, self.M, self.N, self.K, _ptr(a.data), a.data.stride(0), 0, _ptr(a.inv_scale), a.cols, a.scale_block, _ptr(b.data),
            b.data.stride(0), 0, _ptr(b.inv_scale), _ptr(self.out), gr, sg, sr, sc, int(self.accumulate),
            ctypes.c_void_p(self.ws.data_ptr() + off), self.ws.numel() - off, _stream()))

    def product(self):
        cur = torch.cuda.current_stream()
        cur.wait_event(self.ev_factors)
        self._phase(2)
        self.ev_product = cur.record_event()

    def finish(self):
        self.side.wait_event(self.ev_product)
        with torch.cuda.stream(self.side):
            self._phase(4)
            _aux_pending.append(self.side.record_event())
        torch.autograd.graph.increment_version(self.out)


def tensor_inv_scale(bound: torch.Tensor) -> torch.Tensor:
    """2^-e with bound * 2^e in [2^14, 2^15) for a positive finite device scalar ``bound`` (>= the largest magnitude
    of the tensor that will be written with this scale): the fixed_inv_scale of the SP16 producers."""
    lib = _lib.load()
    out = torch.empty(1, dtype=torch.float32, device=bound.device)
    _lib.check(lib.tfgnn_sp_inv_scale_from_bound(_ptr(bound.contiguous()), _ptr(out), _stream()))
    return out


def absmax(x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """device scalar scale * max |x| (tfgnn_absmax; order-independent, hence reproducible)."""
    lib = _lib.load()
    _require_dev(x, torch.float32, "x")
    x = x.contiguous()
    out = torch.zeros(1, dtype=torch.float32, device=x.device)
    _lib.check(lib.tfgnn_absmax(_ptr(x), x.numel(), float(scale), _ptr(out), _stream()))
    return out


_sp_weight_cache: "dict" = {}
_CHECK_WEIGHT_CACHE = [None]


def clear_weight_operand_cache() -> None:
    """Forget the derived forms of the weights (split SP16 operands, transposed / re-stacked copies).  bench.py calls it
    once per step so that the per-step cost of splitting the updated weights is inside the timed region."""
    _sp_weight_cache.clear()


def notify_weights_changed(tensor: Optional[torch.Tensor] = None) -> None:
    """Tell the library that weight values were changed by something torch's version counter does not see - an update
    through ``tensor.data``, a raw-pointer optimizer kernel, another framework writing into the buffer.  Every derived form
    of ``tensor`` (or of all weights) is dropped and rebuilt at its next use.  ``Variable.assign`` / ``Variable.mark_updated``
    call this; in-place torch arithmetic on ``Variable.value`` itself is seen through the version counter as well."""
    aux_flush()  # a deferred split job reads its source weight when it is LAUNCHED: run it on the old values, then drop it
    if tensor is None:
        _sp_weight_cache.clear()
        return
    base = tensor._base if tensor._base is not None else tensor
    lo = base.data_ptr()
    hi = lo + base.numel() * base.element_size()
    for key in [k for k in _sp_weight_cache if lo <= k[0] < hi]:
        del _sp_weight_cache[key]


def _weight_key(w: torch.Tensor, kind: str):
    return (w.data_ptr(), tuple(w.shape), tuple(w.stride()), kind)


def _weight_checksum(w: torch.Tensor):
    """debug mode TFGNN_CHECK_WEIGHT_CACHE=1: a checksum of the weight beside every cached form; a hit whose weight no
    longer has that checksum was updated behind the library's back (see notify_weights_changed) and raises."""
    if _CHECK_WEIGHT_CACHE[0] is None:
        import os

        _CHECK_WEIGHT_CACHE[0] = os.environ.get("TFGNN_CHECK_WEIGHT_CACHE", "0") == "1"
    if not _CHECK_WEIGHT_CACHE[0]:
        return None
    v = w.detach().double()
    return (float(v.sum()), float(v.abs().sum()))


def sp_split_weights(stacks) -> None:
    """Split the stacked kernels [L, D, H] of several layers (one shape) into both operand forms with ONE launch
    (tfgnn_sp_split_weights) and put them where ``sp_weight_operand(W, "cols" | "rows", ...)`` finds them.  Stacks whose
    current value is already split are skipped."""
    import weakref

    todo = []
    for w in stacks:
        hits = [_sp_weight_cache.get(_weight_key(w, kind)) for kind in ("cols", "rows")]
        if not all(h is not None and h[0] == w._version and h[2]() is not None for h in hits):
            todo.append(w)
    if not todo:
        return
    lib = _lib.load()
    L, D, H = todo[0].shape
    for w in todo:
        if tuple(w.shape) != (L, D, H) or not w.is_contiguous():
            raise ValueError("sp_split_weights: contiguous [L, D, H] stacks of one shape")
        _require_dev(w, torch.float32, "kernel stack")
    for start in range(0, len(todo), 16):
        group = todo[start : start + 16]
        n = len(group)
        dev = group[0].device
        cols = [SplitOperand(torch.empty((H, L * D * 4), dtype=torch.uint8, device=dev), torch.empty((H, 1), dtype=torch.float32, device=dev),
                             H, L * D, L * D) for _ in group]
        rows = [SplitOperand(torch.empty((D, L * H * 4), dtype=torch.uint8, device=dev), torch.empty((D, 1), dtype=torch.float32, device=dev),
                             D, L * H, L * H) for _ in group]

        def ptrs(ts):
            return (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])

        _lib.check(lib.tfgnn_sp_split_weights(n, ptrs(group), L, D, H, ptrs([c.data for c in cols]), ptrs([c.inv_scale for c in cols]),
                                              ptrs([r.data for r in rows]), ptrs([r.inv_scale for r in rows]), _stream()))
        if len(_sp_weight_cache) > 256:
            _sp_weight_cache.clear()
        for w, c, r in zip(group, cols, rows):
            base = w._base if w._base is not None else w
            chk = _weight_checksum(w)
            _sp_weight_cache[_weight_key(w, "cols")] = (w._version, c, weakref.ref(base), chk)
            _sp_weight_cache[_weight_key(w, "rows")] = (w._version, r, weakref.ref(base), chk)


def sp_weight_operand(w: torch.Tensor, kind: str, build):
    """A derived form of a weight tensor (its SP16 operands; also the transposed / re-stacked fp32 copies the bf16x3 products
    take), built once per value: keyed on the tensor's storage and version (an in-place torch update bumps the version;
    ``Variable.assign`` and ``notify_weights_changed`` drop the entry explicitly), so forward and backward passes of a step -
    and every step of an evaluation loop - share it.  ``build()`` makes the operand.  Updates torch cannot see
    (``var.value.data.add_(...)``, raw-pointer kernels) MUST be followed by ``Variable.mark_updated()`` /
    ``notify_weights_changed``; TFGNN_CHECK_WEIGHT_CACHE=1 verifies every hit against a checksum of the weight."""
    key = _weight_key(w, kind)
    hit = _sp_weight_cache.get(key)
    if hit is not None and hit[0] == w._version and hit[2]() is not None:
        if hit[3] is not None and hit[3] != _weight_checksum(w):
            raise RuntimeError("a weight tensor changed without a version bump (update through .data or a raw pointer?): "
                               "call Variable.mark_updated() / ops.notify_weights_changed() after such an update")
        return hit[1]
    import weakref

    if hit is not None:
        aux_flush()  # the stale form may still have its (deferred) conversion pending: launch it before the new one is queued
    op = build()
    if len(_sp_weight_cache) > 256:
        _sp_weight_cache.clear()
    base = w._base if w._base is not None else w  # views are temporaries: the entry lives as long as the parameter buffer
    _sp_weight_cache[key] = (w._version, op, weakref.ref(base), _weight_checksum(w))
    return op
