"""tensorflow.contrib.cudnn_rnn of the TF-1.3 test shim (sparse:107-108)."""
from tensorflow import CudnnCompatibleGRUCell  # noqa: F401
