"""Debug tool (not product code): wrap the ops the layers call - gemm, gemm_grad, graph_gather, gather_reduce - and
compare every call with a torch evaluation of the same op ON THE DEVICE (fp64 matmul / index_add), printing shape,
GEMM mode and the error relative to sum |a||b| (products) or sum |w||x| (gathers).  Localises a wrong layer result to
the op (and shape) that produced it.  Usage:

    from tools.op_checker import checking
    with checking(threshold=1e-5) as log: layer(...); layer.backward(...)
    for row in log: print(row)
"""
import contextlib

import torch

from tf2_gnn_amd import ops

_ACTS = {None: lambda x: x, 0: lambda x: x, "relu": torch.relu, "tanh": torch.tanh}


def _act(name, x):
    if name is None or name == 0 or name == "none":
        return x
    name = str(name).lower()
    if name == "relu":
        return torch.relu(x)
    if name == "tanh":
        return torch.tanh(x)
    if name == "sigmoid":
        return torch.sigmoid(x)
    if name == "gelu":
        return torch.nn.functional.gelu(x, approximate="tanh")
    if name == "leaky_relu":
        return torch.nn.functional.leaky_relu(x, 0.2)
    if name == "elu":
        return torch.nn.functional.elu(x)
    if name == "selu":
        return torch.selu(x)
    raise ValueError(name)


@contextlib.contextmanager
def checking(threshold=1e-5, verbose=True):
    log = []
    real = {k: getattr(ops, k) for k in ("gemm", "graph_gather", "gather_reduce")}

    def note(kind, desc, err):
        row = (kind, desc, err)
        log.append(row)
        if verbose and (err != err or err > threshold):
            print(f"[op_checker] {kind} {desc}: error {err:.3e}", flush=True)

    def gemm(a, b, *, trans_a=False, trans_b=False, bias=None, act=0, out=None, accumulate=False):
        prev = out.clone() if (accumulate and out is not None) else None
        res = real["gemm"](a, b, trans_a=trans_a, trans_b=trans_b, bias=bias, act=act, out=out, accumulate=accumulate)
        A = a.double().t() if trans_a else a.double()
        B = b.double().t() if trans_b else b.double()
        ref = A @ B
        mag = A.abs() @ B.abs()
        if bias is not None:
            ref = ref + bias.double()
        ref = _act(act, ref)
        if prev is not None:
            ref = ref + prev.double()
            mag = mag + prev.double().abs()
        err = float(((res.double() - ref).abs() / mag.clamp(min=1e-30)).max()) if ref.numel() else 0.0
        note("gemm", f"ta={int(trans_a)} tb={int(trans_b)} M={ref.shape[0]} N={ref.shape[1]} K={A.shape[1]} act={act} acc={int(accumulate)} "
             f"a.stride={tuple(a.stride())} b.stride={tuple(b.stride())} mode={ops.get_gemm_mode()}", err)
        return res

    def _csr_ref(rowptr, col, inp, edge_weight, row_scale, num_rows):
        counts = (rowptr[1:] - rowptr[:-1]).long()
        row_of = torch.repeat_interleave(torch.arange(num_rows, device=inp.device), counts)
        x = inp.double()[col.long()]
        mag = x.abs()
        if edge_weight is not None:
            ew = edge_weight.double()
            if ew.dim() == 2 and ew.shape[1] > 1:
                k = ew.shape[1]
                x = (x.view(x.shape[0], k, -1) * ew.unsqueeze(-1)).view(x.shape[0], -1)
                mag = (mag.view(mag.shape[0], k, -1) * ew.abs().unsqueeze(-1)).view(mag.shape[0], -1)
            else:
                x = x * ew.view(-1, 1)
                mag = mag * ew.abs().view(-1, 1)
        ref = torch.zeros((num_rows, inp.shape[1]), dtype=torch.float64, device=inp.device).index_add_(0, row_of, x)
        m = torch.zeros_like(ref).index_add_(0, row_of, mag)
        if row_scale is not None:
            ref = ref * row_scale.double().view(-1, 1)
            m = m * row_scale.double().abs().view(-1, 1)
        return ref, m

    def gather_reduce(rowptr, col, inp, *, edge_weight=None, row_scale=None, reduce=0, pre_act=0, post_act=0, out=None):
        res = real["gather_reduce"](rowptr, col, inp, edge_weight=edge_weight, row_scale=row_scale, reduce=reduce,
                                    pre_act=pre_act, post_act=post_act, out=out)
        if reduce == 0 and pre_act in (0, None) and post_act in (0, None):
            n = rowptr.numel() - 1
            ref, m = _csr_ref(rowptr, col, inp, edge_weight, row_scale, n)
            err = float(((res.double() - ref).abs() / m.clamp(min=1e-30)).max()) if ref.numel() else 0.0
            note("gather_reduce", f"rows={n} width={inp.shape[1]} E={col.numel()}", err)
        return res

    def graph_gather(graph, view, inp, *, col=None, edge_weight=None, row_scale=None, reduce=0, pre_act=0, post_act=0, out=None):
        res = real["graph_gather"](graph, view, inp, col=col, edge_weight=edge_weight, row_scale=row_scale, reduce=reduce,
                                   pre_act=pre_act, post_act=post_act, out=out)
        if reduce == 0 and pre_act in (0, None) and post_act in (0, None) and view < 4:
            rp = {0: ops.G_ROWPTR_BY_DST, 1: ops.G_NODEPTR_BY_DST, 2: ops.G_ROWPTR_BY_SRC, 3: ops.G_NODEPTR_BY_SRC}[view]
            dc = {0: ops.G_COL_BY_DST, 1: ops.G_COLL_BY_DST, 2: ops.G_COL_BY_SRC, 3: ops.G_COLL_BY_SRC}[view]
            rowptr = graph.array(rp)
            c = col if col is not None else graph.array(dc)
            n = rowptr.numel() - 1
            ref, m = _csr_ref(rowptr, c, inp, edge_weight, row_scale, n)
            err = float(((res.double() - ref).abs() / m.clamp(min=1e-30)).max()) if ref.numel() else 0.0
            note("graph_gather", f"view={view} rows={n} width={inp.shape[1]} col_override={col is not None} ew={None if edge_weight is None else tuple(edge_weight.shape)}", err)
        return res

    ops.gemm, ops.graph_gather, ops.gather_reduce = gemm, graph_gather, gather_reduce
    try:
        yield log
    finally:
        for k, v in real.items():
            setattr(ops, k, v)
