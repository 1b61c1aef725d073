from .constants import SMALL_NUMBER
from .param_helpers import get_activation_function, get_aggregation_function
