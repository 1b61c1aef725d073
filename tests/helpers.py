"""Shared helpers of the parity tests: seeded graphs, weight transfer HIP layer <-> oracle."""
from __future__ import annotations

import numpy as np
import torch
import torch.overrides


def random_graph(num_nodes, num_edges, num_types, seed=0, empty_types=(), hub=None):
    """Unsorted multigraph; ``hub`` = (node, extra_in_edges) adds a high in-degree target."""
    rng = np.random.default_rng(seed)
    adjs = []
    for l in range(num_types):
        if l in empty_types:
            adjs.append(np.zeros((0, 2), dtype=np.int32))
            continue
        e = num_edges // max(1, num_types - len(empty_types))
        a = rng.integers(0, num_nodes, size=(e, 2)).astype(np.int32)
        if hub is not None and l == 0:
            node, extra = hub
            h = np.stack([rng.integers(0, num_nodes, size=extra), np.full(extra, node)], axis=1).astype(np.int32)
            a = np.concatenate([a, h], axis=0)
            rng.shuffle(a, axis=0)
        adjs.append(a)
    return adjs


def to_dev(arrs, dev):
    return tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs)


def edge_mlp_weights_from_layer(layer):
    """HIP GNN_Edge_MLP-family layer -> oracle weights dict (CPU tensors, same values)."""
    mlps = layer._edge_type_mlps
    w = {"edge_mlps": [[k[l].detach().cpu().clone() for k in mlps.kernels] for l in range(mlps.L)]}
    if getattr(layer, "_aggregation_mlp", None) is not None:
        w["aggr_mlp"] = [k.detach().cpu().clone() for k in layer._aggregation_mlp]
    else:
        w["aggr_mlp"] = None
    film = getattr(layer, "_film_mlps", None)
    if film is not None:
        w["film_mlps"] = [[k[l].detach().cpu().clone() for k in film.kernels] for l in range(film.L)]
    ru = getattr(layer, "_recurrent_unit", None)
    if ru is not None:
        w["gru_kernel"] = ru["kernel"].value.detach().cpu().clone()
        w["gru_recurrent_kernel"] = ru["recurrent_kernel"].value.detach().cpu().clone()
        w["gru_bias"] = ru["bias"].value.detach().cpu().clone()
    return w


def rgat_weights_from_layer(layer):
    H = layer._hidden_dim
    L = layer._num_edge_types
    return {
        "kernels": [layer._kernels[:, l * H : (l + 1) * H].detach().cpu().clone() for l in range(L)],
        "attn": [layer._attn[l].detach().cpu().clone() for l in range(L)],
    }


def mp_weights_from_layer(layer):
    if type(layer).__name__ == "RGAT":
        return rgat_weights_from_layer(layer)
    return edge_mlp_weights_from_layer(layer)


_PARITY_LOG = {}
_MODE_NAMES = {0: "fp32", 6: "bf16x3", 9: "bf16x3_9", 3: "f16x2"}


def current_gemm_mode() -> str:
    try:
        from tf2_gnn_amd import ops

        return _MODE_NAMES.get(ops.get_gemm_mode(), "?")
    except Exception:  # no library (CPU-only run): nothing on the device was measured
        return "cpu"


def launch_counts():
    """tfgnn_launch_counts as a dict (kernel family -> launches so far)."""
    import ctypes

    from tf2_gnn_amd import _lib

    buf = (ctypes.c_int64 * 9)()
    _lib.check(_lib.load().tfgnn_launch_counts(buf, 9))
    names = ["gemm_fp32", "gemm_bf16x3", "sp_nt", "sp_tn", "gather_sp", "gather", "fused_nt", "gemm_stream", "stream_f16x2"]
    return dict(zip(names, list(buf)))


class KernelsUsed:
    """with KernelsUsed() as k: ...; k.delta["sp_nt"] = launches of that family inside the block."""

    def __enter__(self):
        self._before = launch_counts()
        self.delta = {}
        return self

    def __exit__(self, *exc):
        after = launch_counts()
        self.delta = {k: after[k] - self._before[k] for k in after}
        return False


PARITY_LOG_NAME = "parity_r06.json"


def record_parity(key: str, **numbers):
    """Measured parity numbers of the GPU tests, merged into gpurun_out/parity_r06.json (copied to profiles/ after a
    run on the GPU box), keyed by GEMM MODE first: {mode: {check: {max error, bound, test id}}}.  A check that runs
    several times in one mode (parametrised tests sharing a label) keeps its worst case."""
    import json
    import os

    if not key:
        return
    mode = current_gemm_mode()
    entry = {k: (float(v) if isinstance(v, (int, float)) else v) for k, v in numbers.items()}
    entry["test"] = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
    slot = _PARITY_LOG.setdefault(mode, {})
    err_keys = [k for k in entry if k.startswith("max_")]
    old = slot.get(key)
    if old is not None and err_keys and all(old.get(k, 0.0) >= entry[k] for k in err_keys):
        old["count"] = old.get("count", 1) + 1
    else:
        entry["count"] = (old or {}).get("count", 0) + 1
        slot[key] = entry
    _PARITY_LOG["_n"] = _PARITY_LOG.get("_n", 0) + 1
    if _PARITY_LOG["_n"] % 200 == 1:
        _flush_parity_log()


def _worse(a: dict, b: dict) -> dict:
    """Of two records of the same check, the one with the larger measured errors (ties: the first)."""
    keys = [k for k in a if k.startswith("max_")] or [k for k in b if k.startswith("max_")]
    if keys and all(float(b.get(k, 0.0)) >= float(a.get(k, 0.0)) for k in keys) and any(
            float(b.get(k, 0.0)) > float(a.get(k, 0.0)) for k in keys):
        keep = dict(b)
    else:
        keep = dict(a)
    keep["count"] = max(int(a.get("count", 1)), int(b.get("count", 1)))
    return keep


def _flush_parity_log():
    """MERGES this process's records into gpurun_out/parity_r06.json (a partial re-run used to
    overwrite the snapshot of the full suite).  Idempotent: per (mode, check) the record with the larger error stays."""
    import json
    import os

    if not _PARITY_LOG:
        return
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", PARITY_LOG_NAME)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        merged = {}
        try:
            with open(path) as f:
                merged = json.load(f)
        except (OSError, ValueError):
            merged = {}
        for mode, checks in _PARITY_LOG.items():
            if mode == "_n":
                continue
            slot = merged.setdefault(mode, {})
            for key, entry in checks.items():
                slot[key] = _worse(slot[key], entry) if key in slot else entry
        tmp = path + f".{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            json.dump(merged, f, indent=1, sort_keys=True)
        os.replace(tmp, path)
    except OSError:
        pass


import atexit  # noqa: E402

atexit.register(_flush_parity_log)


def assert_close(actual: torch.Tensor, expected: torch.Tensor, tol=1e-5, what=""):
    """|a - b| <= tol * max(1, |b|)  (SURVEY.md section 7 hard part 2: north_star's 1e-5 on fp32 node states)."""
    a = actual.detach().cpu().double()
    b = expected.detach().cpu().double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    if a.numel() == 0:
        return
    err = (a - b).abs() / b.abs().clamp(min=1.0)
    worst = float(err.max())
    if actual.is_cuda or what:
        record_parity(what, max_scaled_error=worst, bound=tol)
    assert worst <= tol, f"{what}: max scaled error {worst:.3e} > {tol:.1e} at {int(err.argmax())}"


def scaled_error(actual: torch.Tensor, expected: torch.Tensor) -> float:
    a = actual.detach().cpu().double()
    b = expected.detach().cpu().double()
    if a.numel() == 0:
        return 0.0
    return float(((a - b).abs() / b.abs().clamp(min=1.0)).max())


# ---- activation kinks ---------------------------------------------------------------------------------------------------
# relu / leaky_relu are not differentiable at 0: a unit whose pre-activation is within fp32 rounding of 0 takes one branch in
# an fp32 forward pass (the reference's TensorFlow one, ours) and possibly the other one in the fp64 oracle, and the two
# GRADIENTS then differ by that unit's whole contribution - O(0.1), not O(1e-7).  With 10^5 .. 10^8 units per test that is
# a certainty at full size and a frequent event at width 128+.  Gradient parity is therefore defined branch by branch:
#   * small cases draw their inputs so that no unit is within `kink_margin` of 0 (`kink_clearance` measures it on the fp64
#     oracle BEFORE the HIP path runs: the selection does not look at HIP results);
#   * full-size cases evaluate the fp64 reference on the branch the HIP forward took (`ForcedKinks` with the masks read
#     back from the layer's saved activations) and record how many decisions differ from fp64's own.
class ForcedKinks(torch.overrides.TorchFunctionMode):
    """Intercepts torch.relu / F.relu / F.leaky_relu inside the oracle.  ``provider(call_index, x)`` returns a bool mask
    (True = positive branch) or None (natural decision).  Counts calls, decisions that differ from x > 0, and the smallest
    |x| relative to max(1, largest |x| of the row)."""

    def __init__(self, provider=None):
        super().__init__()
        self.provider = provider
        self.calls = 0
        self.units = 0
        self.flipped = 0
        self.clearance = float("inf")

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        F = torch.nn.functional
        if func in (torch.relu, F.relu, F.leaky_relu) and len(args) >= 1 and isinstance(args[0], torch.Tensor):
            x = args[0]
            idx = self.calls
            self.calls += 1
            if x.numel():
                xa = x.detach().abs()
                self.units += x.numel()
                self.clearance = min(self.clearance, float((xa / xa.amax(dim=-1, keepdim=True).clamp(min=1.0)).min()))
            mask = self.provider(idx, x) if self.provider is not None else None
            if mask is not None:
                assert mask.shape == x.shape, (idx, tuple(mask.shape), tuple(x.shape))
                self.flipped += int(((x.detach() > 0) != mask).sum())
                slope = 0.0
                if func is F.leaky_relu:
                    slope = args[1] if len(args) > 1 else kwargs.get("negative_slope", 0.01)
                return torch.where(mask, x, slope * x)
        if func is torch.Tensor.scatter_reduce and (kwargs.get("reduce") == "amax" or (len(args) > 4 and args[4] == "amax")):
            # max aggregation (tf.math.unsorted_segment_max): a segment whose two largest entries are within fp32 rounding of
            # each other has its gradient routed to a different edge by an fp32 and an fp64 forward pass - the same kind of
            # kink.  Exact ties (duplicate edges of a multigraph: identical messages) are consistent in both and ignored.
            out = func(*args, **kwargs)
            base, dim, index, src = args[0], args[1], args[2], args[3]
            if src.numel():
                winner = out.gather(dim, index)
                rest = torch.where(src == winner, torch.full_like(src, float("-inf")), src)
                second = torch.full_like(base, float("-inf")).scatter_reduce(dim, index, rest, reduce="amax", include_self=True)
                gap = (out - second) / out.abs().clamp(min=1.0)
                ok = torch.isfinite(second) & torch.isfinite(out)
                if bool(ok.any()):
                    self.clearance = min(self.clearance, float(gap[ok].min()))
            return out
        return func(*args, **kwargs)


def kink_clearance(fn):
    """Smallest relative distance of any relu / leaky_relu input from 0 while ``fn()`` runs (fp64 oracle forward)."""
    with ForcedKinks() as w:
        fn()
    return w.clearance


KINK_MARGIN = 2e-6  # ~10x the fp32 rounding error of an O(1) pre-activation
