"""tf2_gnn_amd: MI355X-native (gfx950) implementation of the message-passing hot path of
microsoft/tf2-gnn behind the reference's ``tf2_gnn.layers`` interface.

Hand-written HIP kernels in ``libtfgnn.so`` (C ABI: include/tfgnn.h) do all the work; PyTorch-ROCm
tensors provide device memory and streams.  There is no CPU fallback: ops raise if the library is
missing or a tensor is not on a ROCm device.
"""
from . import ops  # noqa: F401
from .layers import (  # noqa: F401
    GGNN,
    GNN,
    GNN_Edge_MLP,
    GNNInput,
    MessagePassing,
    MessagePassingInput,
    NodesToGraphRepresentationInput,
    RGAT,
    RGCN,
    RGIN,
    WASGraphRepresentation,
    WeightedSumGraphRepresentation,
)

from .capture import CapturedStep  # noqa: F401,E402
from .autograd import TorchGNN, TorchGraphTaskModel, TorchMessagePassing, TorchNodesToGraphRepresentation  # noqa: F401,E402

__version__ = "0.2.0"
