"""Host-side mirror of the reference's utils.py pieces the hot path touches:
SMALL_NUMBER (utils.py:8), glorot_init (:11-13), ThreadedIterator (:16-36) and MLP (:39-70).
"""
from __future__ import annotations

import collections
import queue
import threading

import numpy as np
import torch

SMALL_NUMBER = 1e-7


class ThreadedIterator:
    """utils.py:16-36: a producer thread runs the wrapped iterator `max_queue_size` elements ahead of the consumer.

    The reference uses it to pack the next minibatch on the host while `sess.run` works on the current one
    (chem_tensorflow.py:219).  Here the elements are batches assembled ON THE GPU (data_device.pack_batch_device: ~80 small
    launches and two device->host reads per batch, 1 ms of mostly host time), so the producer works on its own HIP stream:
    its launches and its host reads no longer sit in front of the training step's kernels in the consumer's stream.  An
    element is handed over with an event (the consumer's stream waits for it, not the host), and the consumer's reference
    to an element is kept until the work the consumer queued on it has completed -- its memory belongs to the producer
    stream's allocator pool and must not be recycled under kernels that still read it.

    device: a torch.device of type "cuda" to get the side stream; None / CPU: a plain prefetch thread.
    stream: the side stream to use (a model keeps ONE packer stream for all its epochs: a fresh stream per epoch leaves every
    epoch's batch memory cached in its own allocator pool); default: a new stream.  Whatever the constructing thread has queued on
    ITS current stream so far (the resident dataset and its tables) is ordered before the producer's first launch."""

    def __init__(self, original_iterator, max_queue_size: int = 2, device=None, stream=None):
        self.__queue = queue.Queue(maxsize=max_queue_size)
        self.__cuda = device is not None and torch.device(device).type == "cuda" and torch.cuda.is_available()
        self.__device = torch.device(device) if self.__cuda else None
        self.__stream = (stream if stream is not None else torch.cuda.Stream(self.__device)) if self.__cuda else None
        self.__start_after = None
        if self.__cuda:
            with torch.cuda.device(self.__device):
                self.__start_after = torch.cuda.Event()
                self.__start_after.record()
        self.__stop = threading.Event()
        self.__retired = collections.deque()
        self.__thread = threading.Thread(target=self.__worker, args=(original_iterator,), daemon=True)
        self.__thread.start()

    def __put(self, item) -> bool:
        while not self.__stop.is_set():
            try:
                self.__queue.put(item, timeout=0.05)
                return True
            except queue.Full:
                pass
        return False

    def __worker(self, original_iterator):
        try:
            if self.__cuda:
                torch.cuda.set_device(self.__device)
                with torch.cuda.stream(self.__stream):
                    self.__stream.wait_event(self.__start_after)
                    for element in original_iterator:
                        ev = torch.cuda.Event()
                        ev.record()
                        if not self.__put((element, ev, None)):
                            return
            else:
                for element in original_iterator:
                    if not self.__put((element, None, None)):
                        return
            self.__put((None, None, None))
        except BaseException as exc:                                  # surfaces in the consumer, like the reference's would
            self.__put((None, None, exc))

    def __retire(self, element):
        if self.__cuda and element is not None:
            ev = torch.cuda.Event()
            ev.record()                                               # after everything the consumer queued on `element`
            self.__retired.append((ev, element))
        while self.__retired and self.__retired[0][0].query():
            self.__retired.popleft()

    def __iter__(self):
        prev = None
        try:
            while True:
                element, ev, exc = self.__queue.get()
                self.__retire(prev)
                prev = None
                if exc is not None:
                    raise exc
                if element is None:
                    break
                if ev is not None:
                    torch.cuda.current_stream().wait_event(ev)
                prev = element
                yield element
        finally:
            self.__stop.set()                                         # consumer left early: let the producer go
            self.__retire(prev)
            for ev, _ in self.__retired:
                ev.synchronize()
            self.__retired.clear()
            self.__thread.join(timeout=5.0)


class StreamPrefetcher:
    """Software pipeline over HIP streams, on ONE host thread: the next `depth` elements of the wrapped iterator (batches assembled
    on the GPU, data_device.pack_batches_device) are produced on side streams while the consumer's kernels for the current element
    run; the consumer's work goes round-robin onto `consumer_streams` (independent batches: the tail of one batch's kernels is
    back-filled by the next batch's).  Yields (element, stream): the stream already waits for the element's packing; the caller
    queues its work on it (`with torch.cuda.stream(stream)`).  An element is kept alive until the work queued on its stream up
    to the next hand-out has completed (its memory belongs to a packing stream's pool).

    Packing a batch is a CHAIN of ~15 small dependent launches.  Next to a forward pass whose persistent workgroups own every CU
    for the length of a launch, each link waits for a launch boundary, so one chain takes longer than a forward pass: the
    packing streams are created with high priority (their kernels are dispatched first when workgroup slots free up) and
    `depth` chains are kept in flight on `pack_streams` separate streams, so that a batch has `depth` forward passes' time to
    come together.

    The reference overlaps host-side packing with sess.run through a producer thread (utils.py:16-36, ThreadedIterator above);
    with packing on the device a thread is not needed to overlap it -- only more streams -- and the launching thread keeps the
    interpreter lock to itself."""

    def __init__(self, original_iterator, device, consumer_streams=None, pack_streams=2, depth=3, priority=-1):
        self._it = original_iterator
        self._device = torch.device(device)
        if isinstance(pack_streams, int):
            pack_streams = [torch.cuda.Stream(self._device, priority=priority) for _ in range(max(1, pack_streams))]
        self._packs = list(pack_streams)
        self._depth = max(1, int(depth))
        self._streams = list(consumer_streams) if consumer_streams else [None]
        self._retired = collections.deque()

    def _retire(self, element, stream):
        if element is not None:
            ev = torch.cuda.Event()
            ev.record(stream)
            self._retired.append((ev, element))
        while self._retired and self._retired[0][0].query():
            self._retired.popleft()

    def __iter__(self):
        it = iter(self._it)
        queue = collections.deque()
        prev, prev_stream, k, produced, exhausted = None, None, 0, 0, False
        start = torch.cuda.Event()
        start.record()                                   # whatever built the resident dataset is ordered before the first pack
        for ps in self._packs:
            ps.wait_event(start)
        def produce(n):
            nonlocal produced, exhausted
            for _ in range(n):
                if exhausted:
                    return
                with torch.cuda.stream(self._packs[produced % len(self._packs)]):
                    try:
                        element = next(it)
                    except StopIteration:
                        exhausted = True
                        return
                    ready = torch.cuda.Event()
                    ready.record()
                queue.append((element, ready))
                produced += 1

        try:
            while True:
                # An element is handed out as soon as one is ready; the queue is topped up (two elements per hand-out, up to
                # `depth`) AFTER the consumer has queued its kernels for it -- packing `depth` batches up front, or setting up the
                # next epoch (graph order upload, batch boundaries) in front of a hand-out, leaves the GPU idle for as long.
                if not queue:
                    produce(1)
                    if not queue:
                        break
                element, ready = queue.popleft()
                self._retire(prev, prev_stream)
                stream = self._streams[k % len(self._streams)] or torch.cuda.current_stream(self._device)
                stream.wait_event(ready)
                prev, prev_stream, k = element, stream, k + 1
                yield element, stream
                produce(min(2, self._depth - len(queue)))
        finally:
            self._retire(prev, prev_stream)
            for ev, _ in self._retired:
                ev.synchronize()
            self._retired.clear()


def glorot_init(shape):
    """utils.py:11-13 -- uniform(+-sqrt(6/(shape[-2]+shape[-1]))) from the global NumPy RNG."""
    initialization_range = np.sqrt(6.0 / (shape[-2] + shape[-1]))
    return np.random.uniform(low=-initialization_range, high=initialization_range, size=shape).astype(np.float32)


def tf_glorot_uniform(shape, generator: torch.Generator) -> torch.Tensor:
    """The kernels of TF-1.3's RNN cells (sparse:104-110, dense:88) come from `tf.get_variable`'s default
    glorot_uniform_initializer, i.e. from TensorFlow's graph-seeded generator (chem_tensorflow.py:85) and NOT from
    np.random: they must not consume the NumPy stream, or every np.random-drawn variable after the first cell
    (edge weights of later layers, the readout MLPs; utils.py:11-13,64-65) would differ from the reference's for
    the same `random_seed`.  TF's Philox stream itself is not reproducible; a torch CPU generator seeded with the
    same seed stands in for it."""
    limit = float(np.sqrt(6.0 / (shape[-2] + shape[-1])))
    return (torch.rand(tuple(shape), generator=generator, dtype=torch.float32) * 2 - 1) * limit


def dropout_seed(random_seed: int, step: int, *site) -> int:
    """64-bit Philox key of one dropout site: a hash of (random_seed, optimisation step, site) -- the same on every rank of a
    data-parallel job and on every platform (no generator state anywhere)."""
    import hashlib
    text = repr((int(random_seed), int(step)) + tuple(site)).encode()
    return int.from_bytes(hashlib.blake2b(text, digest_size=8).digest(), "little")


class _CounterDropout(torch.autograd.Function):
    """x / keep * floor(keep + U) with the mask re-derived from (seed, row key, column) in the backward pass."""

    @staticmethod
    def forward(ctx, x, keep_prob, seed, row_key):
        from . import ops
        ctx.args = (float(keep_prob), int(seed), row_key)
        return ops.dropout(x.contiguous(), keep_prob, seed, row_key)

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        keep_prob, seed, row_key = ctx.args
        return ops.dropout(dy.contiguous(), keep_prob, seed, row_key), None, None, None


def tf_dropout(x: torch.Tensor, keep_prob: float, seed=None, row_key=None) -> torch.Tensor:
    """tf.nn.dropout: x / keep * floor(keep + U[0,1)); identity at keep_prob == 1.  U is counter-based (ggnn_dropout_f32): a
    function of (seed, row key, column) -- see dropout_seed.  Like every op of the path it runs on the GPU only."""
    if keep_prob >= 1.0:
        return x
    if seed is None:
        raise ValueError("dropout with keep_prob < 1 needs a seed (utils.dropout_seed)")
    return _CounterDropout.apply(x, float(keep_prob), int(seed), row_key)


class _TallLinear(torch.autograd.Function):
    """x @ W + b for tall x ([V,K], V ~ 1e5) on the package's own kernels -- the op-by-op readout of chem_tensorflow_sparse.py:220-231
    / utils.py:64-70, used only where the fused readout + loss kernels do not apply (an unsorted graph_nodes_list, widths beyond
    256): forward and d x on the tiled GEMM (ggnn_gemm_f32: the weight matrix padded to 4 output columns, x as column segments
    of a width the kernel takes), the weight gradient x^T dy on the row-split kernel ggnn_gemm_tn_f32, the bias gradient by
    ggnn_colsum_f32.  No vendor BLAS."""

    @staticmethod
    def _segments(x):
        """x [V,K] as equal column segments of a width the GEMM kernel takes (a multiple of 100, 64 or 32)."""
        K = x.shape[1]
        for seg in (K, K // 2, K // 3, K // 4):
            if seg and K % seg == 0 and K // seg <= 4 and (seg % 100 == 0 or seg % 32 == 0):
                return [x[:, i * seg:(i + 1) * seg] for i in range(K // seg)]
        raise NotImplementedError("the op-by-op readout needs an input width whose halves / thirds / quarters are multiples of "
                                  "32 or 100 (got %d); the fused readout (sorted graph_nodes_list, width <= 256) has no such limit" % K)

    @staticmethod
    def _pad4(W):
        n = W.shape[1]
        if n % 4 == 0:
            return W.contiguous(), n
        Wp = W.new_zeros((W.shape[0], (n + 3) // 4 * 4))
        Wp[:, :n] = W
        return Wp, n

    @staticmethod
    def forward(ctx, x, W, b):
        from . import ops
        x = x.contiguous()
        ctx.save_for_backward(x, W)
        Wp, n = _TallLinear._pad4(W)
        return ops.gemm(_TallLinear._segments(x), Wp)[:, :n] + b

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, W = ctx.saved_tensors
        n = W.shape[1]
        dyp = dy.new_zeros((dy.shape[0], (n + 3) // 4 * 4))
        dyp[:, :n] = dy
        dx = None
        if ctx.needs_input_grad[0]:
            WTp = W.new_zeros((dyp.shape[1], (W.shape[0] + 3) // 4 * 4))          # W^T, rows and columns padded to the kernel's granules
            WTp[:n, :W.shape[0]] = W.t()
            if dyp.shape[1] % 32 == 0:
                dx = ops.gemm([dyp], WTp)[:, :W.shape[0]]
            else:                                                                    # n <= 4 columns: d x = sum_j dy[:, j] W[:, j]^T
                dx = sum(dyp[:, j:j + 1] * WTp[j:j + 1, :W.shape[0]] for j in range(n))
        dW = ops.gemm_tn(x, dyp)[:, :n]
        db = ops.colsum(dyp)[:n]
        return dx, dW, db


class MLP(object):
    """utils.py:39-70.  With hid_sizes=[] (the only use, chem_tensorflow.py:153-157) the call returns
    the PRE-activation of the single layer: inputs @ dropout(W) + b  (utils.py:64-70)."""

    def __init__(self, in_size, out_size, hid_sizes, dropout_keep_prob, device="cpu"):
        self.in_size = in_size
        self.out_size = out_size
        self.hid_sizes = list(hid_sizes)
        self.dropout_keep_prob = dropout_keep_prob
        self.device = device
        self.dropout_seed = None      # callable(layer) -> 64-bit seed of this layer's weight mask (set by the model)
        self.params = self.make_network_params()

    def make_network_params(self):
        dims = [self.in_size] + self.hid_sizes + [self.out_size]
        weight_sizes = list(zip(dims[:-1], dims[1:]))
        weights = [torch.from_numpy(self.init_weights(s)).to(self.device) for s in weight_sizes]
        biases = [torch.zeros(s[-1], dtype=torch.float32, device=self.device) for s in weight_sizes]
        return {"weights": weights, "biases": biases}

    def init_weights(self, shape):
        # numpy 1.13 (the reference's pin) keeps float32 here (value-based scalar casting); numpy 2 would
        # promote to float64 through the np.float64 scalar, so cast explicitly
        return (np.sqrt(6.0 / (shape[-2] + shape[-1])) * (2 * np.random.rand(*shape).astype(np.float32) - 1)).astype(np.float32)

    def dropped_weight(self, layer: int) -> torch.Tensor:
        """utils.py:68: tf.nn.dropout on the layer's weight matrix (identity at keep_prob 1)."""
        keep = self.dropout_keep_prob() if callable(self.dropout_keep_prob) else self.dropout_keep_prob
        W = self.params["weights"][layer]
        if keep >= 1.0:
            return W
        return tf_dropout(W, keep, self.dropout_seed(layer) if self.dropout_seed is not None else None)

    def __call__(self, inputs):
        acts = inputs
        hid = acts
        for i, b in enumerate(self.params["biases"]):
            hid = _TallLinear.apply(acts, self.dropped_weight(i), b)
            acts = torch.relu(hid)
        return hid
