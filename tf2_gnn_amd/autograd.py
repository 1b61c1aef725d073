"""torch.autograd integration of the explicit backward passes.

The reference's ``call()`` outputs are differentiable by ``tf.GradientTape`` (tf2_gnn/models/graph_task_model.py:347-357:
``tape.gradient(loss, trainable_variables)``).  The layers of this package carry hand-derived reverse passes instead
(``layer.backward(grad_output)``); the modules below put them behind ``torch.autograd.Function`` so that a model
written in torch on top of them trains with ``loss.backward()``:

    gnn = TorchGNN(GNN(params))
    head = torch.nn.Linear(hidden, 1).cuda()
    gnn.build(GNNInput(features, adjacency_lists, node_to_graph_map, num_graphs))       # weights exist from here on
    opt = torch.optim.Adam(list(gnn.parameters()) + list(head.parameters()))
    out = gnn(GNNInput(features, adjacency_lists, node_to_graph_map, num_graphs))      # training mode = gnn.training
    loss = head(out).square().mean()
    loss.backward()          # runs GNN.backward on the HIP kernels; fills p.grad of every parameter (and Variable.grad)
    opt.step()

Every ``Variable`` of the wrapped layer is exposed as a ``torch.nn.Parameter`` that ALIASES its storage: an optimizer step
is an in-place update of the layer's own weights (several layers keep their kernels as views into one stacked buffer; the
parameters are those views).  The forward pass runs outside the autograd tape - no torch op on the path records anything -
and the saved state lives in the layer (``layer._ctx``), so ONE backward per forward, like the layers' own ``backward``;
forward passes are numbered, and ``backward`` raises when the layer no longer holds the state of the pass it belongs to
(a second forward - also one under ``torch.no_grad()`` - before ``loss.backward()``).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import ops
from .layers.gnn import GNN, GNNInput
from .layers.message_passing.message_passing import MessagePassing, MessagePassingInput, Variable
from .layers.nodes_to_graph_representation import NodesToGraphRepresentation, NodesToGraphRepresentationInput


class _LayerFunction(torch.autograd.Function):
    """forward: runner.run(x) -> tuple of tensors; backward: runner.back(grads) -> (d x | None, [d parameter | None])."""

    @staticmethod
    def forward(ctx, runner, x, *params):
        ctx.runner = runner
        ctx.set_materialize_grads(False)
        outs = runner.run(x)
        # every forward pass REPLACES the state the layer keeps for its reverse pass (layer._ctx, runner._cur): stamp it, so
        # that a backward pass can tell whether the state in the layer is still the one of ITS forward pass
        runner._gen += 1
        runner._taped_gen = runner._gen
        ctx.gen = runner._gen
        return outs

    @staticmethod
    def backward(ctx, *grads):
        runner = ctx.runner
        if ctx.gen != runner._gen:
            raise RuntimeError("another forward pass of this module ran after the one being differentiated (forward pass "
                               f"{ctx.gen}, the layer now holds the state of pass {runner._gen} - e.g. a validation forward "
                               "between forward and loss.backward()): the explicit reverse pass keeps its context in the layer, "
                               "not in the autograd graph, so the gradients would be those of the wrong batch")
        if runner._consumed_gen == ctx.gen:
            raise RuntimeError("the layer's saved state has been consumed: one backward pass per forward pass "
                               "(the explicit reverse pass keeps its context in the layer, not in the autograd graph)")
        runner._consumed_gen = ctx.gen
        dx, pgrads = runner.back(grads)
        return (None, dx, *pgrads)


class _AutogradModule(torch.nn.Module):
    """Base: parameters aliasing the Variables of ``self.layer``; subclasses implement _run / _back."""

    def __init__(self, layer):
        super().__init__()
        self.layer = layer
        self._vars: List[Variable] = []
        self._params = torch.nn.ParameterList()
        self._versions: List[int] = []
        self._gen = 0            # forward passes run so far (each replaces the layer's saved state)
        self._consumed_gen = -1  # the pass whose state a backward pass has used up
        self._taped_gen = -1     # the latest pass autograd recorded (a forward under torch.no_grad() is not one: ADVICE r5)
        self._cur = None
        if getattr(layer, "built", False):
            self._adopt_variables()

    # ---- parameters ------------------------------------------------------------------------------------------------
    def _adopt_variables(self):
        if self._vars:
            return
        self._vars = list(self.layer.trainable_variables)
        for v in self._vars:
            self._params.append(torch.nn.Parameter(v.value, requires_grad=True))  # same storage: updates are the layer's
        self._versions = [p._version for p in self._params]

    def _sync_versions(self):
        """an optimizer step updated the weights in place: drop the operand forms derived from the old values"""
        for i, (p, v) in enumerate(zip(self._params, self._vars)):
            if p._version != self._versions[i] or p.data_ptr() != v.value.data_ptr():
                if p.data_ptr() != v.value.data_ptr():
                    raise RuntimeError(f"parameter {v.name} no longer aliases the layer's weight (was it re-assigned? use .data.copy_())")
                ops.notify_weights_changed(v.value)
                self._versions[i] = p._version

    def build(self, example_inputs) -> "_AutogradModule":
        """Create the layer's weights from the shapes of ``example_inputs`` (the layers build lazily, at their first call)
        and register them as parameters - call this before handing ``module.parameters()`` to an optimizer."""
        if not getattr(self.layer, "built", False):
            self.layer(example_inputs, training=False)
        self._adopt_variables()
        return self

    # ---- autograd plumbing -----------------------------------------------------------------------------------------
    def _apply(self, x: torch.Tensor, state) -> Sequence[torch.Tensor]:
        self._cur = state
        self._sync_versions()
        if not torch.is_grad_enabled():
            # no tape (validation under torch.no_grad()): nothing will ask for this pass's gradients, but the layer's saved
            # state is replaced all the same - a backward pass of an EARLIER forward must fail loudly, not use it
            self._gen += 1
            return self.run(x)
        return _LayerFunction.apply(self, x, *self._params)

    @property
    def pending_backward(self) -> bool:
        """True while the latest forward pass has been recorded by autograd and not differentiated yet."""
        return self._taped_gen == self._gen and self._consumed_gen != self._gen

    def run(self, x):
        raise NotImplementedError

    def back(self, grads):
        raise NotImplementedError

    def _param_grads(self):
        out = []
        for v in self._vars:
            g = v.grad
            out.append(None if g is None else g.reshape(v.value.shape))
        return out


class TorchGNN(_AutogradModule):
    """``tf2_gnn_amd.layers.GNN`` as a torch module: ``module(GNNInput, return_all_representations=False)`` is differentiable
    with respect to the stack's weights and to ``inputs.node_features`` (when that tensor requires grad).  Training mode
    (layer-input dropout) follows ``module.training``."""

    def __init__(self, gnn: GNN):
        super().__init__(gnn)

    def forward(self, inputs: GNNInput, return_all_representations: bool = False):
        if not self.layer.built:
            self.layer(inputs, training=False)  # builds the stack (shapes from the inputs); result discarded
        self._adopt_variables()
        outs = self._apply(inputs.node_features, (inputs, bool(return_all_representations)))
        return (outs[0], tuple(outs[1:])) if return_all_representations else outs[0]

    def run(self, x):
        inputs, want_all = self._cur
        self._x_needs_grad = bool(x.requires_grad)
        res = self.layer(inputs._replace(node_features=x.detach()), training=self.training, return_all_representations=want_all)
        return (res[0], *res[1]) if want_all else (res,)

    def back(self, grads):
        _, want_all = self._cur
        g_out = None if grads[0] is None else grads[0].contiguous()
        g_all = None
        if want_all and any(g is not None for g in grads[1:]):
            g_all = [None if g is None else g.contiguous() for g in grads[1:]]
        if g_out is None and g_all is None:
            return None, [None] * len(self._vars)
        dx = self.layer.backward(g_out, need_input_grad=self._x_needs_grad, grad_all_representations=g_all)
        return (dx if self._x_needs_grad else None), self._param_grads()


class TorchMessagePassing(_AutogradModule):
    """One message passing layer (RGCN, RGAT, RGIN, GGNN, GNN_Edge_MLP, GNN_FiLM or a user subclass of MessagePassing)."""

    def __init__(self, layer: MessagePassing):
        super().__init__(layer)

    def forward(self, inputs: MessagePassingInput):
        if not self.layer.built:
            self.layer(inputs, training=False)
        self._adopt_variables()
        return self._apply(inputs.node_embeddings, inputs)[0]

    def run(self, x):
        inputs = self._cur
        return (self.layer(inputs._replace(node_embeddings=x.detach()), training=self.training),)

    def back(self, grads):
        if grads[0] is None:
            return None, [None] * len(self._vars)
        dx = self.layer.backward(grads[0].contiguous())
        return dx, self._param_grads()


class TorchNodesToGraphRepresentation(_AutogradModule):
    """A node -> graph pooling layer (WeightedSumGraphRepresentation, WASGraphRepresentation)."""

    def __init__(self, layer: NodesToGraphRepresentation):
        super().__init__(layer)

    def forward(self, inputs: NodesToGraphRepresentationInput):
        if not self.layer.built:
            self.layer(inputs, training=False)
        self._adopt_variables()
        return self._apply(inputs.node_embeddings, inputs)[0]

    def run(self, x):
        inputs = self._cur
        return (self.layer(inputs._replace(node_embeddings=x.detach()), training=self.training),)

    def back(self, grads):
        if grads[0] is None:
            return None, [None] * len(self._vars)
        dx = self.layer.backward(grads[0].contiguous())
        return dx, self._param_grads()


class TorchGraphTaskModel(_AutogradModule):
    """A task model of ``tf2_gnn_amd.tasks`` (NodeMulticlassTask, QM9RegressionTask, GraphRegressionTask):
    ``module(batch_features)`` returns the task output (per-node logits / per-graph predictions), differentiable with
    respect to every weight of the GNN and of the head - compute any torch loss on it and call ``loss.backward()``.  (The
    models' own ``compute_task_metrics`` + ``backward()`` stay available: they fuse the reference's losses with their gradients.)"""

    def __init__(self, model):
        super().__init__(model)

    def forward(self, batch_features):
        if not self.layer.built:
            self.layer(batch_features, training=False)
        self._adopt_variables()
        return self._apply(batch_features["node_features"], batch_features)[0]

    def run(self, x):
        out = self.layer(self._cur, training=self.training)
        self._tuple_out = isinstance(out, tuple)
        return (out[0] if self._tuple_out else out,)

    def back(self, grads):
        if grads[0] is None:
            return None, [None] * len(self._vars)
        step = self.layer._step
        if step is None:
            raise RuntimeError("the task model's saved state is gone (another forward pass ran in between)")
        step["dloss"] = grads[0].contiguous()
        self.layer.backward()
        return None, self._param_grads()
