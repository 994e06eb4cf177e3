"""tf.contrib namespace of the TF-1.3 test shim (see ../__init__.py; test infrastructure only)."""
from . import cudnn_rnn, rnn  # noqa: F401
