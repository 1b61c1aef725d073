"""Layer-level C entry points (include/tfgnn.h tfgnn_mp_forward / tfgnn_mp_backward, round 6): one library call per
message-passing layer and pass must be the op-level route it replaces - the same kernels with the same arguments - BIT FOR
BIT: output, d(node states), every kernel gradient, also with the epilogues a layer stack hangs on the products (the next
layer's dropout, the gradient factors of the op below, accumulation into another term) and inside a whole training step."""
import pytest
import torch

from tests.helpers import KernelsUsed, random_graph, to_dev

pytestmark = pytest.mark.gpu


def _layer(cls_name, H, L, over=()):
    import tf2_gnn_amd.layers.message_passing as mp

    cls = getattr(mp, cls_name)
    p = cls.get_default_hyperparameters()
    p.update({"hidden_dim": H})
    p.update(dict(over))
    mp.set_seed(21)
    layer = cls(p)
    layer.build(mp.MessagePassingInput((None, H), tuple((None, 2) for _ in range(L))))
    return layer


@pytest.mark.parametrize("cls_name,H,L,V,E", [("RGCN", 320, 4, 1500, 40000), ("RGCN", 128, 3, 700, 9000), ("GGNN", 128, 2, 900, 12000),
                                              ("GNN_Edge_MLP", 256, 5, 400, 6000)])
def test_one_call_per_pass_equals_the_op_level_route(dev, monkeypatch, cls_name, H, L, V, E):
    from tf2_gnn_amd import ops
    from tf2_gnn_amd.layers import MessagePassingInput

    over = {"use_target_state_as_input": False, "num_edge_MLP_hidden_layers": 0} if cls_name == "GNN_Edge_MLP" else {}
    layer = _layer(cls_name, H, L, over)
    gen = torch.Generator().manual_seed(3)
    X = torch.randn((V, H), generator=gen).to(dev)
    dOut = torch.randn((V, H), generator=gen).to(dev)
    adj = to_dev(random_graph(V, E, L, seed=5, hub=(3, 200)), dev)  # hubs: long buckets -> the deferred combine pass runs
    results = {}
    for entry in ("1", "0"):
        monkeypatch.setenv("TFGNN_MP_ENTRY", entry)
        ops.clear_weight_operand_cache()
        with KernelsUsed() as k:
            out = layer(MessagePassingInput(X, adj), training=True)
            dX = layer.backward(dOut)
        torch.cuda.synchronize()
        assert layer._ctx.get("f16x2"), "the layer did not take the split-operand route"
        assert k.delta["sp_nt"] >= 2 and k.delta["sp_tn"] >= 1 and k.delta["gather_sp"] >= 2, k.delta
        results[entry] = (out.clone(), dX.clone(), [v.grad.clone() for v in layer.trainable_variables], dict(k.delta))
    a, b = results["1"], results["0"]
    assert a[3] == b[3], (a[3], b[3])  # the same launches
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for v, ga, gb in zip(layer.trainable_variables, a[2], b[2]):
        assert torch.equal(ga, gb), v.name


@pytest.mark.parametrize("mp_style,H", [("rgcn", 320), ("ggnn", 128)])
def test_training_step_of_a_stack_is_bit_equal_with_and_without_the_layer_entry(dev, monkeypatch, mp_style, H):
    """The benchmarked stack settings (dropout 0.1 fused into the producers' epilogues, Dense after layer 0, cross-layer
    gradient epilogues): every epilogue variant the stack asks of the products goes through the entry's argument struct."""
    from bench import model_params
    from tf2_gnn_amd import ops
    from tf2_gnn_amd.layers import GNN, GNNInput
    from tf2_gnn_amd.layers.message_passing import set_seed

    V, E, L = 2500, 60000, 4
    gen = torch.Generator().manual_seed(7)
    X = torch.randn((V, H), generator=gen).to(dev)
    dOut = torch.randn((V, H), generator=gen).to(dev)
    adj = to_dev(random_graph(V, E, L, seed=11, hub=(5, 300)), dev)
    outs = {}
    for entry in ("1", "0"):
        monkeypatch.setenv("TFGNN_MP_ENTRY", entry)
        set_seed(13)
        gnn = GNN(model_params(mp_style, H, 3))
        ops.clear_weight_operand_cache()
        for _ in range(2):
            gnn._dropout_calls = 0
            out = gnn(GNNInput(X, adj, torch.zeros(V, dtype=torch.int32, device=dev), 1), training=True)
            dX = gnn.backward(dOut, need_input_grad=True)
        torch.cuda.synchronize()
        assert ops.get_gemm_mode() == ops.GEMM_F16X2
        outs[entry] = (out.clone(), dX.clone(), [v.grad.clone() for v in gnn.trainable_variables], [v.name for v in gnn.trainable_variables])
    a, b = outs["1"], outs["0"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for name, ga, gb in zip(a[3], a[2], b[2]):
        assert torch.equal(ga, gb), name


def test_a_stale_weight_operand_is_rebuilt_inside_the_call(dev):
    """The entry receives the kernels only when the cached split form is stale (once per weight value): an in-place update
    followed by mark_updated() must show in the next forward pass."""
    from tf2_gnn_amd import ops
    from tf2_gnn_amd.layers import MessagePassingInput

    layer = _layer("RGCN", 128, 3)
    V = 600
    X = torch.randn((V, 128), generator=torch.Generator().manual_seed(1)).to(dev)
    adj = to_dev(random_graph(V, 7000, 3, seed=2), dev)
    out1 = layer(MessagePassingInput(X, adj), training=False).clone()
    out1b = layer(MessagePassingInput(X, adj), training=False).clone()  # cached operand
    assert torch.equal(out1, out1b)
    for v in layer.trainable_variables:
        v.value.mul_(2.0)
        v.mark_updated()
    out2 = layer(MessagePassingInput(X, adj), training=False)
    torch.cuda.synchronize()
    assert not torch.equal(out2, out1)
    assert float((out2 - 2.0 * out1).abs().max()) <= 1e-5 * float(out2.abs().max())  # relu(2 a) = 2 relu(a)
