"""CPU oracle for the tf2-gnn message-passing hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only
as the checker.  The product path (``tf2_gnn_amd``) never imports this package and fails loudly when
its HIP library is missing.

Parity status: the reference (TensorFlow 2 + dpu_utils) cannot be imported in the build container
(neither package is installed, no network), so this is a *restatement* that follows the reference's
literal op sequence file by file.  It is pinned by every numeric known-answer vector the reference's
own tests hold for the path (``tests/golden/reference_kats.json``; see ``tests/test_oracle_golden.py``):

  * tf2_gnn/test/layers/test_message_passing.py:35-71     (4 gather/segment-sum/relu vectors)
  * tf2_gnn/layers/message_passing/message_passing.py:238-249 (in-degree doctest)
  * tf2_gnn/test/data/test_utils.py:50-115                (8 adjacency-processing vectors)
  * tf2_gnn/test/layers/test_RGCN.py / test_RGAT.py       (parameter-shape contracts)

Everything the reference's tests do not pin numerically (degree normalisation, RGAT softmax, GRU,
pooling, [ext] TensorFlow / dpu_utils semantics) is "parity unpinned": stated in DESIGN.md 7.
"""
