"""Generate tests/golden/reference_molecule_batch.json by EXECUTING the reference's own data pipeline on its own
fixture (tf2_gnn/test/test_datasets/train.jsonl.gz, 10 molecules).

Run in the build container (the GPU box has no /root/reference):  python tests/golden/make_reference_molecule_batch.py

TensorFlow and dpu_utils are not installed, but the code that turns JSON lines into a minibatch -
JsonLGraphPropertyDataset._process_raw_datapoint (data/jsonl_graph_property_dataset.py:75-93), process_adjacency_lists
(data/utils.py:9-58) and GraphDataset.graph_batch_iterator_from_graph_iterator / _add_graph_to_batch / _finalise_batch
(data/graph_dataset.py:161-246) - is pure python + numpy: the two missing packages are replaced by inert stand-ins in
sys.modules (they are only touched by type annotations and by methods that are never called here) and the reference
classes run unmodified.  The fixture holds the raw per-graph data, the per-graph processed samples and the batches the
reference produced for two dataset configurations."""
import gzip
import json
import sys
import types
from pathlib import Path
from unittest import mock

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent / "reference_molecule_batch.json"


def _install_stand_ins():
    for name in ("tensorflow", "dpu_utils", "dpu_utils.utils", "dpu_utils.tf2utils", "docopt", "h5py"):
        m = mock.MagicMock(name=name)
        m.__path__ = []  # behaves as a package for "from x.y import z"
        sys.modules[name] = m
    sys.path.insert(0, str(REF))


def main():
    _install_stand_ins()
    # import the data sub-package only (tf2_gnn/__init__ pulls in the Keras layers)
    pkg = types.ModuleType("tf2_gnn")
    pkg.__path__ = [str(REF / "tf2_gnn")]
    sys.modules["tf2_gnn"] = pkg
    data_pkg = types.ModuleType("tf2_gnn.data")
    data_pkg.__path__ = [str(REF / "tf2_gnn" / "data")]
    sys.modules["tf2_gnn.data"] = data_pkg
    from tf2_gnn.data.graph_dataset import DataFold  # noqa: E402
    from tf2_gnn.data.jsonl_graph_property_dataset import JsonLGraphPropertyDataset  # noqa: E402

    with gzip.open(REF / "tf2_gnn/test/test_datasets/train.jsonl.gz", "rt") as f:
        raw = [json.loads(line) for line in f]
    graphs = [{"node_features": r["graph"]["node_features"], "adjacency_lists": r["graph"]["adjacency_lists"],
               "Property": r["Property"]} for r in raw]

    configs = []
    for tie, self_loops, max_nodes in ((True, True, 10000), (False, False, 60)):
        params = JsonLGraphPropertyDataset.get_default_hyperparameters()
        params.update({"num_fwd_edge_types": 4, "tie_fwd_bkwd_edges": tie, "add_self_loop_edges": self_loops,
                       "max_nodes_per_batch": max_nodes})
        ds = JsonLGraphPropertyDataset(params)
        samples = [ds._process_raw_datapoint(r) for r in raw]
        batches = []
        for feats, labels in ds.graph_batch_iterator_from_graph_iterator(iter(samples)):
            batches.append({
                "node_features": np.asarray(feats["node_features"]).tolist(),
                "node_to_graph_map": feats["node_to_graph_map"].tolist(),
                "num_graphs_in_batch": int(feats["num_graphs_in_batch"]),
                "adjacency_lists": [feats[f"adjacency_list_{i}"].astype(np.int64).tolist() for i in range(ds.num_edge_types)],
                "target_value": [float(v) for v in labels["target_value"]],
            })
        configs.append({
            "params": {k: params[k] for k in ("num_fwd_edge_types", "tie_fwd_bkwd_edges", "add_self_loop_edges",
                                              "max_nodes_per_batch")},
            "num_edge_types": ds.num_edge_types,
            "samples": [{"adjacency_lists": [a.astype(np.int64).tolist() for a in s.adjacency_lists],
                         "type_to_node_to_num_inedges": np.asarray(s.type_to_node_to_num_inedges).tolist(),
                         "target_value": float(s.target_value)} for s in samples],
            "batches": batches,
        })
    OUT.write_text(json.dumps({"source": "tf2_gnn/test/test_datasets/train.jsonl.gz run through the reference's "
                                         "JsonLGraphPropertyDataset (TensorFlow / dpu_utils replaced by inert stand-ins)",
                               "graphs": graphs, "configs": configs}))
    print("wrote", OUT, OUT.stat().st_size, "bytes;", [(c["num_edge_types"], len(c["batches"])) for c in configs])


if __name__ == "__main__":
    main()
