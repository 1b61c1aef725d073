from .constants import LAYER_NORM_EPSILON, LEAKY_RELU_ALPHA, SEGMENT_SOFTMAX_EPSILON, SMALL_NUMBER
from .param_helpers import get_activation_function, get_aggregation_function

__all__ = [
    "LAYER_NORM_EPSILON",
    "LEAKY_RELU_ALPHA",
    "SEGMENT_SOFTMAX_EPSILON",
    "SMALL_NUMBER",
    "get_activation_function",
    "get_aggregation_function",
]
