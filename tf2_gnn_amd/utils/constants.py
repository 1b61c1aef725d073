"""Numeric constants of the path.  SMALL_NUMBER is the reference's only project-wide constant
(tf2_gnn/utils/constants.py:2: the in-degree normalisation divides by count + SMALL_NUMBER); the others are the
[ext] TensorFlow / Keras defaults the kernels hard-code (csrc/common.hpp, csrc/elementwise.hip), named here so that
the Python layers, the tests and the documentation quote one source."""
SMALL_NUMBER = 1e-7
LEAKY_RELU_ALPHA = 0.2  # tf.nn.leaky_relu default
LAYER_NORM_EPSILON = 1e-3  # tf.keras.layers.LayerNormalization default
SEGMENT_SOFTMAX_EPSILON = 1e-7  # dpu_utils unsorted_segment_softmax
