"""torch.autograd wrappers (tf2_gnn_amd/autograd.py): a torch-written head on top of the HIP layers trains with
``loss.backward()`` - the reference's tf.GradientTape contract (tf2_gnn/models/graph_task_model.py:347-357)."""
import pytest
import torch

from tests.helpers import random_graph, to_dev

pytestmark = pytest.mark.gpu


def _stack(dev, mp_style="rgcn", H=32, layers=3, rate=0.2):
    from tf2_gnn_amd.layers import GNN, GNNInput
    from tf2_gnn_amd.layers.message_passing import set_seed

    V, L, Din = 300, 3, 16
    params = GNN.get_default_hyperparameters(mp_style)
    params.update({"hidden_dim": H, "num_layers": layers, "global_exchange_every_num_layers": 10000,
                   "layer_input_dropout_rate": rate, "dense_every_num_layers": 2, "residual_every_num_layers": 2})
    set_seed(11)
    gnn = GNN(params)
    adjs = random_graph(V, 4000, L, seed=5)
    X = torch.randn((V, Din), generator=torch.Generator().manual_seed(2)).to(dev)
    inp = GNNInput(X, to_dev(adjs, dev), torch.zeros(V, dtype=torch.int32, device=dev), 1)
    return gnn, inp, V, H


@pytest.mark.parametrize("mp_style,H", [("rgcn", 32), ("ggnn", 128), ("rgat", 96)])
def test_torch_head_on_gnn_equals_the_manual_backward(dev, mp_style, H):
    """loss = head(gnn(x)) written in torch: d loss / d (every GNN weight, node_features) through TorchGNN equals
    gnn.backward(d loss / d gnn output) called by hand, bit for bit (same kernels, same masks)."""
    from tf2_gnn_amd import TorchGNN

    gnn, inp, V, _ = _stack(dev, mp_style, H)
    mod = TorchGNN(gnn).train()
    head = torch.nn.Linear(H, 3).to(dev)
    x = inp.node_features.clone().requires_grad_(True)
    calls0 = gnn._dropout_calls
    out = mod(inp._replace(node_features=x))
    assert out.requires_grad and len(list(mod.parameters())) == len(gnn.trainable_variables)
    loss = head(out).tanh().square().sum()
    loss.backward()
    got = [p.grad.clone() for p in mod.parameters()]
    got_x = x.grad.clone()
    # by hand: the head alone in torch, then the explicit reverse pass with the same dropout masks
    gnn._dropout_calls = calls0
    out2 = gnn(inp, training=True).detach().requires_grad_(True)
    assert torch.equal(out2, out.detach())
    head(out2).tanh().square().sum().backward()
    dx = gnn.backward(out2.grad.contiguous(), need_input_grad=True)
    for p, g, v in zip(mod.parameters(), got, gnn.trainable_variables):
        assert p.data_ptr() == v.value.data_ptr()
        assert torch.equal(g, v.grad.reshape(g.shape)), v.name
    assert torch.equal(got_x, dx)
    with pytest.raises(RuntimeError):
        loss.backward()  # the saved state is the layer's: one backward per forward


def test_optimizer_steps_update_the_layer_weights_and_the_loss_goes_down(dev):
    """parameters alias the Variables: an Adam step is an in-place update of the stack's own (stacked, split-cached) weights"""
    from tf2_gnn_amd import TorchGNN

    gnn, inp, V, H = _stack(dev, "rgcn", 128, layers=2, rate=0.0)
    mod = TorchGNN(gnn).build(inp).train()
    target = torch.randn((V, H), generator=torch.Generator().manual_seed(3)).to(dev)
    opt = torch.optim.Adam(mod.parameters(), lr=3e-3)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        loss = (mod(inp) - target).square().mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(b < a for a, b in zip(losses, losses[1:])), losses  # every step of this small problem is a descent step
    w_before = gnn.trainable_variables[0].value.clone()
    opt.zero_grad()
    (mod(inp) - target).square().mean().backward()
    opt.step()
    assert not torch.equal(w_before, gnn.trainable_variables[0].value)


def test_message_passing_pooling_and_task_model_wrappers(dev):
    from tf2_gnn_amd import TorchGraphTaskModel, TorchMessagePassing, TorchNodesToGraphRepresentation
    from tf2_gnn_amd.layers import (MessagePassingInput, NodesToGraphRepresentationInput, RGCN,
                                    WeightedSumGraphRepresentation)
    from tf2_gnn_amd.layers.message_passing import set_seed
    from tf2_gnn_amd.tasks import NodeMulticlassTask

    V, L, H, G = 240, 2, 32, 6
    adjs = to_dev(random_graph(V, 2000, L, seed=1), dev)
    gen = torch.Generator().manual_seed(4)
    X = torch.randn((V, H), generator=gen).to(dev).requires_grad_(True)
    n2g = torch.arange(V, dtype=torch.int32, device=dev) // (V // G)
    set_seed(2)
    p = RGCN.get_default_hyperparameters()
    p["hidden_dim"] = H
    layer = RGCN(p)
    mp = TorchMessagePassing(layer).eval()  # (eval: the pooling MLPs would draw fresh dropout masks on the second run)
    pool_layer = WeightedSumGraphRepresentation(graph_representation_size=8, num_heads=2, weighting_fun="softmax",
                                                scoring_mlp_layers=[16], transformation_mlp_layers=[16])
    pool = TorchNodesToGraphRepresentation(pool_layer).eval()
    h = mp(MessagePassingInput(X, adjs))
    z = pool(NodesToGraphRepresentationInput(h, n2g, G))
    loss = z.square().sum()
    loss.backward()
    # by hand
    h2 = layer(MessagePassingInput(X.detach(), adjs), training=mp.training)
    z2 = pool_layer(NodesToGraphRepresentationInput(h2, n2g, G), training=pool.training)
    dh = pool_layer.backward((2.0 * z2).contiguous())
    dX = layer.backward(dh)
    assert torch.equal(X.grad, dX)
    for m, lay in ((mp, layer), (pool, pool_layer)):
        for prm, v in zip(m.parameters(), lay.trainable_variables):
            assert torch.equal(prm.grad, v.grad.reshape(prm.grad.shape)), v.name
    # a task model: logits out, any torch loss on top
    params = NodeMulticlassTask.get_default_hyperparameters("rgcn")
    params.update({"gnn_hidden_dim": H, "gnn_num_layers": 2})
    model = NodeMulticlassTask(params, num_edge_types=L, num_node_target_labels=5)
    tm = TorchGraphTaskModel(model).eval()
    feats = {"node_features": X.detach(), "node_to_graph_map": n2g, "num_graphs_in_batch": G,
             **{f"adjacency_list_{i}": a for i, a in enumerate(adjs)}}
    logits = tm(feats)
    labels = (torch.rand((V, 5), generator=gen) > 0.5).float().to(dev)
    torch.nn.functional.binary_cross_entropy_with_logits(logits, labels, reduction="sum").div(V).backward()
    got = [prm.grad.clone() for prm in tm.parameters()]
    out = model(feats, training=False)
    model.compute_task_metrics(feats, out, {"node_labels": labels})  # the reference's loss: sum over labels, mean over nodes
    for (v, g), mine in zip(model.backward(), got):
        scale = max(float(g.abs().max()), 1e-30)
        assert float((mine.reshape(g.shape) - g).abs().max()) / scale <= 1e-5, v.name


def test_backward_after_another_forward_fails_loudly(dev):
    """the reverse pass keeps its context in the layer; a second forward - also a validation forward under
    torch.no_grad() - replaces it.  The backward pass of the FIRST output must raise instead of returning the gradients of the
    wrong batch; the latest recorded forward stays differentiable."""
    from tf2_gnn_amd import TorchGNN

    gnn, inp, V, H = _stack(dev, "rgcn", 32, rate=0.0)
    mod = TorchGNN(gnn).train()
    other = inp._replace(node_features=inp.node_features * 2.0)

    loss_a = mod(inp).square().sum()
    with torch.no_grad():
        mod.eval()
        mod(other)  # validation pass between forward and backward
        mod.train()
    with pytest.raises(RuntimeError, match="another forward pass"):
        loss_a.backward()

    loss_a = mod(inp).square().sum()
    loss_b = mod(other).square().sum()
    with pytest.raises(RuntimeError, match="another forward pass"):
        loss_a.backward()
    loss_b.backward()  # the state in the layer is this pass's
    got = [p.grad.clone() for p in mod.parameters()]
    for p in mod.parameters():
        p.grad = None
    mod(other).square().sum().backward()
    for g, p in zip(got, mod.parameters()):
        assert torch.equal(g, p.grad)
    assert not mod.pending_backward
    #  a forward under torch.no_grad() is not a recorded pass: nothing is pending after it, and a recorded forward
    # that a no_grad forward has overwritten is no longer pending either (its backward can only raise)
    with torch.no_grad():
        mod(inp)
    assert not mod.pending_backward
    loss_c = mod(inp).square().sum()
    assert mod.pending_backward
    with torch.no_grad():
        mod(other)
    assert not mod.pending_backward
    with pytest.raises(RuntimeError, match="another forward pass"):
        loss_c.backward()
