"""tf.contrib.rnn of the TF-1.3 test shim: GRUCell is the same class as tf.nn.rnn_cell.GRUCell (dense:88)."""
from tensorflow import BasicRNNCell, DropoutWrapper, GRUCell  # noqa: F401
