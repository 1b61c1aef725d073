"""Generate tests/golden/reference_kats.json from the reference checkout (/root/reference).

Run in the build container (the GPU box has no /root/reference):  python tests/golden/make_reference_kats.py

What is extracted, and how:
  * message_passing_kats   - the 4 known-answer vectors of
        tf2_gnn/test/layers/test_message_passing.py:35-71, read with ``ast`` (TensorFlow cannot be
        imported here, so the literals inside tf.constant(...) are evaluated directly);
  * num_incoming_doctest   - the doctest of calculate_type_to_num_incoming_edges,
        tf2_gnn/layers/message_passing/message_passing.py:238-249, parsed from the docstring;
  * adjacency_cases        - tf2_gnn/test/data/test_utils.py:50-115: inputs AND outputs produced by
        EXECUTING the reference's own tf2_gnn/data/utils.py (pure numpy, loaded as a stand-alone
        module), plus 6 seeded random cases run through the same reference function;
  * default_hyperparameters - the dict literals of get_default_hyperparameters of MessagePassing,
        GNN_Edge_MLP, RGCN, RGIN, GGNN, RGAT, GNN_FiLM and GNN (ast), for the API-compatibility tests;
  * rgcn_shape_cases / rgat_shape_cases - test/layers/test_RGCN.py:8-12, test_RGAT.py:9-28.
"""
import ast
import importlib.util
import json
import re
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent / "reference_kats.json"


def _const_arg(call: ast.Call):
    return ast.literal_eval(call.args[0])


def message_passing_kats():
    src = (REF / "tf2_gnn/test/layers/test_message_passing.py").read_text()
    tree = ast.parse(src)
    kats = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "id", None) == "TestInput":
            mp_in = node.args[0]
            kw = {k.arg: k.value for k in mp_in.keywords}
            feats = _const_arg(kw["node_embeddings"])
            adjs = [_const_arg(c) for c in kw["adjacency_lists"].elts]
            expected = _const_arg({k.arg: k.value for k in node.keywords}["aggregated_states"])
            kats.append({"node_embeddings": feats, "adjacency_lists": adjs, "aggregated_states": expected})
    assert len(kats) == 4, len(kats)
    return kats


def num_incoming_doctest():
    src = (REF / "tf2_gnn/layers/message_passing/message_passing.py").read_text()
    doc = src[src.index("def calculate_type_to_num_incoming_edges") :]
    adjs = [ast.literal_eval(m) for m in re.findall(r"tf\.constant\((\[\[.*?\]\]), dtype=tf\.int32\)", doc)]
    rows = re.findall(r"\[\[?([0-9. ]+)\]", doc[doc.index("tf.Tensor(") : doc.index("shape=(3, 5)")])
    expected = [[float(x) for x in r.split()] for r in rows]
    assert len(adjs) == 3 and len(expected) == 3, (adjs, expected)
    return {"num_nodes": 5, "adjacency_lists": adjs, "expected": expected}


def load_reference_data_utils():
    spec = importlib.util.spec_from_file_location("ref_data_utils", REF / "tf2_gnn/data/utils.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def adjacency_cases():
    ref = load_reference_data_utils()
    cases = []

    def run(adjs, num_nodes, add_self, tie, self_type):
        tied = ref.get_tied_edge_types(tie, len(adjs))
        out_adj, counts = ref.process_adjacency_lists(
            adjacency_lists=[list(map(tuple, a)) for a in adjs],
            num_nodes=num_nodes,
            add_self_loop_edges=add_self,
            tied_fwd_bkwd_edge_types=tied,
            self_loop_edge_type=self_type,
        )
        cases.append(
            {
                "adjacency_lists": [[list(e) for e in a] for a in adjs],
                "num_nodes": num_nodes,
                "add_self_loop_edges": add_self,
                "tie_fwd_bkwd_edges": tie,
                "self_loop_edge_type": self_type,
                "expected_adjacency_lists": [a.tolist() for a in out_adj],
                "expected_counts": counts.tolist(),
            }
        )

    one = [[(0, 1), (1, 2)]]
    two = [[(0, 1)], [(1, 2)]]
    # the 8 cases of test/data/test_utils.py:50-115, in order
    run(one, 3, False, False, 0)
    run(one, 3, False, True, 0)
    run(one, 3, True, False, 0)
    run(one, 3, True, True, 0)
    run(one, 3, True, False, -1)
    run(one, 3, True, True, -1)
    run(two, 3, False, [0], 0)
    run(two, 3, False, [1], 0)
    # expected values stated in the reference test file for those 8 cases (checked below)
    stated = [
        ([[(0, 1), (1, 2)], [(1, 0), (2, 1)]], [[0, 1, 1], [1, 1, 0]]),
        ([[(0, 1), (1, 2), (1, 0), (2, 1)]], [[1, 2, 1]]),
        ([[(0, 0), (1, 1), (2, 2)], [(0, 1), (1, 2)], [(1, 0), (2, 1)]], [[1, 1, 1], [0, 1, 1], [1, 1, 0]]),
        ([[(0, 0), (1, 1), (2, 2)], [(0, 1), (1, 2), (1, 0), (2, 1)]], [[1, 1, 1], [1, 2, 1]]),
        ([[(0, 1), (1, 2)], [(1, 0), (2, 1)], [(0, 0), (1, 1), (2, 2)]], [[0, 1, 1], [1, 1, 0], [1, 1, 1]]),
        ([[(0, 1), (1, 2), (1, 0), (2, 1)], [(0, 0), (1, 1), (2, 2)]], [[1, 2, 1], [1, 1, 1]]),
        ([[(0, 1), (1, 0)], [(1, 2)], [(2, 1)]], [[1, 1, 0], [0, 0, 1], [0, 1, 0]]),
        ([[(0, 1)], [(1, 2), (2, 1)], [(1, 0)]], [[0, 1, 0], [0, 1, 1], [1, 0, 0]]),
    ]
    for c, (adj, cnt) in zip(cases, stated):
        assert c["expected_adjacency_lists"] == [[list(e) for e in a] for a in adj], c
        assert c["expected_counts"] == [[float(x) for x in r] for r in cnt], c
    # extra seeded cases through the same reference function
    rng = np.random.default_rng(0)
    for i in range(6):
        n = int(rng.integers(1, 9))
        n_types = int(rng.integers(1, 4))
        adjs = []
        for _ in range(n_types):
            m = int(rng.integers(0, 7))
            adjs.append([(int(rng.integers(0, n)), int(rng.integers(0, n))) for _ in range(m)])
        tie = [True, False, [0]][i % 3]
        run(adjs, n, bool(i % 2), tie, [0, -1, 1][i % 3] if n_types >= 1 else 0)
    return cases


def default_hyperparameters():
    files = {
        "MessagePassing": "tf2_gnn/layers/message_passing/message_passing.py",
        "GNN_Edge_MLP": "tf2_gnn/layers/message_passing/gnn_edge_mlp.py",
        "RGCN": "tf2_gnn/layers/message_passing/rgcn.py",
        "RGIN": "tf2_gnn/layers/message_passing/rgin.py",
        "GGNN": "tf2_gnn/layers/message_passing/ggnn.py",
        "RGAT": "tf2_gnn/layers/message_passing/rgat.py",
        "GNN_FiLM": "tf2_gnn/layers/message_passing/gnn_film.py",
        "GNN": "tf2_gnn/layers/gnn.py",
    }
    out = {}
    for cls, rel in files.items():
        tree = ast.parse((REF / rel).read_text())
        for node in ast.walk(tree):
            if isinstance(node, ast.ClassDef) and node.name == cls:
                for fn in node.body:
                    if isinstance(fn, ast.FunctionDef) and fn.name == "get_default_hyperparameters":
                        for sub in ast.walk(fn):
                            if isinstance(sub, ast.Dict) and sub.keys:
                                out[cls] = ast.literal_eval(sub)
                                break
    assert set(out) == set(files), out.keys()
    return out


def shape_cases():
    rgcn = [[3, 3, 5], [1, 1, 1], [7, 14, 7]]  # (D, L, hidden_dim)  test_RGCN.py:8-12
    rgat = [[3, 3, 16, 8], [1, 1, 2, 1], [7, 14, 64, 4]]  # (D, L, hidden_dim, heads)  test_RGAT.py:9-28
    src = (REF / "tf2_gnn/test/layers/test_RGCN.py").read_text()
    assert "dims=(None, 7)" in src and "range(14)), 7)" in src
    src = (REF / "tf2_gnn/test/layers/test_RGAT.py").read_text()
    assert "range(14))" in src and "64," in src
    return rgcn, rgat


def main():
    rgcn, rgat = shape_cases()
    data = {
        "_generated_by": "tests/golden/make_reference_kats.py from /root/reference (microsoft/tf2-gnn v2.14.0)",
        "message_passing_kats": message_passing_kats(),
        "num_incoming_doctest": num_incoming_doctest(),
        "adjacency_cases": adjacency_cases(),
        "default_hyperparameters": default_hyperparameters(),
        "rgcn_shape_cases": rgcn,
        "rgat_shape_cases": rgat,
    }
    OUT.write_text(json.dumps(data, indent=1))
    print("wrote", OUT, {k: (len(v) if hasattr(v, "__len__") else v) for k, v in data.items()})


if __name__ == "__main__":
    sys.exit(main())
