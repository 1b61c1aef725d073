"""``tf2_gnn_amd.capture.CapturedStep``: a training step on a static batch captured into one hipGraph and replayed
(the reference traces its step into one tf.function graph: tf2_gnn/models/graph_task_model.py:327-357).  A replay must be the
step: same outputs and gradients as the eager run, bit for bit; weights updated in place between replays are seen; dropout
masks change per replay (the epoch word, include/tfgnn.h) and equal the eager masks of the same (seed, epoch)."""
import pytest
import torch

from tests.helpers import random_graph, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def epoch_zero_afterwards():
    from tf2_gnn_amd import ops

    yield
    ops.dropout_epoch_set(0)  # every other test draws the masks of epoch 0
    torch.cuda.synchronize()
    assert ops.dropout_epoch() == 0


def _stack(dev, mp_style, H, rate, layers=2, V=700, E=9000, L=3, D=64):
    from tf2_gnn_amd.layers import GNN, GNNInput
    from tf2_gnn_amd.layers.message_passing import set_seed

    params = GNN.get_default_hyperparameters(mp_style)
    params.update({"hidden_dim": H, "num_layers": layers, "global_exchange_every_num_layers": 10000,
                   "layer_input_dropout_rate": rate, "dense_every_num_layers": 2, "residual_every_num_layers": 2})
    set_seed(5)
    gnn = GNN(params)
    gen = torch.Generator().manual_seed(8)
    X = torch.randn((V, D), generator=gen).to(dev)
    dOut = torch.randn((V, H), generator=gen).to(dev)
    inp = GNNInput(X, to_dev(random_graph(V, E, L, seed=9, hub=(4, 150)), dev), torch.zeros(V, dtype=torch.int32, device=dev), 1)
    return gnn, inp, dOut


@pytest.mark.parametrize("mp_style,H", [("rgcn", 128), ("ggnn", 128), ("rgat", 96), ("gnn_edge_mlp", 64)])
def test_replay_equals_the_eager_step_and_follows_weight_updates(dev, mp_style, H):
    from tf2_gnn_amd import CapturedStep, ops

    gnn, inp, dOut = _stack(dev, mp_style, H, rate=0.0)

    def step():
        out = gnn(inp, training=True)
        dx = gnn.backward(dOut, need_input_grad=True)
        # whatever a caller wants from a replay is RETURNED by the step: replays rewrite these tensors; attributes the step
        # assigns (v.grad) are Python state of the run that was captured and are re-bound by any later eager run
        return out, dx, [v.grad for v in gnn.trainable_variables]

    def snapshot(res):
        torch.cuda.synchronize()
        return res[0].clone(), res[1].clone(), [g.clone() for g in res[2]]

    def same(a, b):
        return torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and all(torch.equal(x, y) for x, y in zip(a[2], b[2]))

    cap = CapturedStep(step)
    cap.capture()
    for _ in range(2):
        res = cap.replay()
    replayed = snapshot(res)
    assert same(replayed, snapshot(step()))
    # an optimizer step in place (what torch.optim does to a parameter that aliases the weight): the next replay uses the
    # new values - its weight conversions are nodes of the graph
    for v in gnn.trainable_variables:
        v.value.mul_(0.5)
    after = snapshot(cap.replay())
    assert not torch.equal(after[0], replayed[0])
    assert same(after, snapshot(step()))
    assert cap.replays == 3 and not cap.guard_tripped()
    assert ops.dropout_epoch() == 3  # one advance per replay (the capture itself executes nothing)


def test_every_replay_draws_fresh_dropout_masks_that_match_the_eager_masks_of_that_epoch(dev):
    from tf2_gnn_amd import CapturedStep, ops

    gnn, inp, dOut = _stack(dev, "rgcn", 128, rate=0.3)

    def step():
        out = gnn(inp, training=True)
        dx = gnn.backward(dOut, need_input_grad=True)
        return out, dx, [v.grad for v in gnn.trainable_variables]

    for _ in range(4):  # the warm-up by hand: the test needs the stack's seed counter as it stands when the capture runs
        step()
    seeds_at = gnn._dropout_calls
    cap = CapturedStep(step, warmup=0)
    cap.capture()
    outs = []
    for _ in range(3):
        o, d, gr = cap.replay()
        torch.cuda.synchronize()
        outs.append((o.clone(), d.clone(), [g.clone() for g in gr]))
    assert not torch.equal(outs[0][0], outs[1][0]) and not torch.equal(outs[1][0], outs[2][0])
    assert ops.dropout_epoch() == 3
    # eager, same seeds, epoch 2 = the second replay
    ops.dropout_epoch_set(2)
    gnn._dropout_calls = seeds_at
    o, d, gr = step()
    torch.cuda.synchronize()
    assert torch.equal(o, outs[1][0]) and torch.equal(d, outs[1][1])
    for v, g, ge in zip(gnn.trainable_variables, outs[1][2], gr):
        assert torch.equal(g, ge), v.name
    # and epoch 0 is the mask of (seed, element) alone: the stored-mask path and the epilogue path agree there as everywhere
    ops.dropout_epoch_set(0)
    m0 = ops.dropout_mask((64, 32), 0.3, 1234, device=dev)
    ops.dropout_epoch_set(7)
    m7 = ops.dropout_mask((64, 32), 0.3, 1234, device=dev)
    ops.dropout_epoch_set(0)
    m0b = ops.dropout_mask((64, 32), 0.3, 1234, device=dev)
    assert torch.equal(m0, m0b) and not torch.equal(m0, m7)
    assert 0.6 < float((m7 > 0).float().mean()) < 0.8


def test_node_multiclass_training_step_replayed(dev):
    """BASELINE configs[0]'s step (bench.py --workload ppi): NodeMulticlassTask forward + sigmoid cross-entropy / micro-F1 +
    backward, captured after the batch was finalised and bucketed once."""
    from tf2_gnn_amd import CapturedStep
    from tf2_gnn_amd.data import make_ppi_shaped_batch, process_adjacency_lists
    from tf2_gnn_amd.layers.message_passing import set_seed
    from tf2_gnn_amd.tasks import NodeMulticlassTask

    feats, fwd, n2g, labels = make_ppi_shaped_batch(2, 300, 6, 50, 121, seed=3)
    V = feats.shape[0]
    X = torch.from_numpy(feats).to(dev)
    adjs, _ = process_adjacency_lists([torch.from_numpy(fwd).to(dev)], V, add_self_loop_edges=True, tied_fwd_bkwd_edge_types=set())
    params = NodeMulticlassTask.get_default_hyperparameters("rgcn")
    params.update({"gnn_hidden_dim": 320, "gnn_num_layers": 2, "gnn_layer_input_dropout_rate": 0.0})  # (H = 320, 3 types: the products split K in their launch)
    set_seed(1)
    model = NodeMulticlassTask(params, num_edge_types=3, num_node_target_labels=121)
    batch = {"node_features": X, "node_to_graph_map": torch.from_numpy(n2g).to(dev), "num_graphs_in_batch": 2,
             **{f"adjacency_list_{i}": a for i, a in enumerate(adjs)}}
    lab = {"node_labels": torch.from_numpy(labels).to(dev)}

    def step():
        out = model(batch, training=True)
        metrics = model.compute_task_metrics(batch, out, lab)
        return metrics, [g for _, g in model.backward()]

    cap = CapturedStep(step)
    cap.capture()
    m_c, g_c = cap.replay()
    torch.cuda.synchronize()
    loss_c, f1_c = float(m_c["loss"]), float(m_c["f1_score"])
    grads_c = [g.clone() for g in g_c]
    m_e, g_e = step()
    torch.cuda.synchronize()
    assert float(m_e["loss"]) == loss_c and float(m_e["f1_score"]) == f1_c
    for v, g, ge in zip(model.trainable_variables, grads_c, g_e):
        assert torch.equal(g, ge), v.name
    # new features / labels written INTO the captured tensors are what the next replay trains on
    X.mul_(-1.0)
    m_c, _ = cap.replay()
    torch.cuda.synchronize()
    loss_c = float(m_c["loss"])
    assert loss_c != float(m_e["loss"]) and float(step()[0]["loss"]) == loss_c
