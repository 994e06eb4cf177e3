"""TEST INFRASTRUCTURE (part of oracle/, NOT product code): a deferred-execution stand-in for the slice of
the ``tensorflow==1.3.0`` Python API that the reference's model files touch, evaluated with torch-CPU fp32.

Why it exists: the reference's arithmetic lives in TensorFlow 1.3 (requirements.txt:2), which is absent from
/root/reference and cannot be installed here.  With this package first on ``sys.path`` the reference's OWN
source files (chem_tensorflow.py, chem_tensorflow_sparse.py, chem_tensorflow_dense.py, utils.py) import and
run unmodified: their graph construction, variable creation order, numpy-side initialisation, batch packer,
loss, per-variable clipping and training loop are the reference's code, executed; only the TF ops underneath
are restated here from TF-1.3's published semantics (each op cites the TF-1.3 source it follows).
``tests/golden/make_reference_golden.py`` uses it to produce ``tests/golden/reference_*.npz``.

What this pins and what it does not: a wiring / op-order / shape / initialisation mistake in oracle/ggnn_oracle*.py
or in the package shows up against these vectors; a misreading of a TF kernel's arithmetic that is repeated
here (GRUCell gate order, Adam's epsilon placement, ...) does not.  DESIGN.md §3 says the same.

Semantics: ``tf.*`` calls build a DAG of ``Tensor`` nodes; ``Session.run`` evaluates the fetched nodes once
per call (memoised) on torch-CPU tensors, floats in fp32, integers in int64.  Gradients come from torch
autograd over the same evaluation.  Ops that TF runs in index order on CPU (unsorted_segment_sum) use
``index_add_`` (serial over the index on CPU).  Never imported by the package or by anything on the GPU box.
"""
from __future__ import annotations

import contextlib
import sys
import types

import numpy as np
import torch

__version__ = "1.3.0-shim"

# ------------------------------------------------------------------------------------------------
# dtypes
# ------------------------------------------------------------------------------------------------


class DType:
    def __init__(self, name, torch_dtype, np_dtype):
        self.name, self.torch, self.as_numpy_dtype = name, torch_dtype, np_dtype

    def __repr__(self):
        return "tf." + self.name


float32 = DType("float32", torch.float32, np.float32)
float64 = DType("float64", torch.float64, np.float64)
int32 = DType("int32", torch.int64, np.int32)      # evaluated as int64 (torch indexing), reported as int32
int64 = DType("int64", torch.int64, np.int64)


def _dtype_of(value: torch.Tensor) -> DType:
    if value.dtype == torch.float64:
        return float64
    return float32 if value.dtype.is_floating_point else int32


# ------------------------------------------------------------------------------------------------
# graph, scopes, collections
# ------------------------------------------------------------------------------------------------
class GraphKeys:
    TRAINABLE_VARIABLES = "trainable_variables"
    GLOBAL_VARIABLES = "variables"
    LOCAL_VARIABLES = "local_variables"


class Graph:
    def __init__(self):
        self.collections = {GraphKeys.TRAINABLE_VARIABLES: [], GraphKeys.GLOBAL_VARIABLES: [],
                            GraphKeys.LOCAL_VARIABLES: []}
        self.names = {}
        self.scope = []
        self.seed = None

    @contextlib.contextmanager
    def as_default(self):
        _graph_stack.append(self)
        try:
            yield self
        finally:
            _graph_stack.pop()

    def get_collection(self, key, scope=None):
        items = list(self.collections.get(key, []))
        if scope is not None:
            items = [v for v in items if v.name.startswith(scope)]       # TF: re.match(scope, name)
        return items

    def unique_name(self, name):
        """TF's name uniquification: 'x', 'x_1', 'x_2', ... per fully-scoped name."""
        full = "/".join(self.scope + [name])
        n = self.names.get(full, 0)
        self.names[full] = n + 1
        return full if n == 0 else "%s_%d" % (full, n)


_graph_stack = [Graph()]


def get_default_graph():
    return _graph_stack[-1]


class _Scope:
    def __init__(self, name):
        self.name = name

    def reuse_variables(self):           # dense:102 -- the shim's cells own their variables, nothing to do
        pass


@contextlib.contextmanager
def variable_scope(name, reuse=None):
    g = get_default_graph()
    if isinstance(name, _Scope):
        name = name.name
    g.scope.append(name)
    try:
        yield _Scope("/".join(g.scope))
    finally:
        g.scope.pop()


@contextlib.contextmanager
def name_scope(name):
    yield name


def get_variable_scope():
    return _Scope("/".join(get_default_graph().scope))


def set_random_seed(seed):
    get_default_graph().seed = seed
    torch.manual_seed(seed)


# ------------------------------------------------------------------------------------------------
# tensors
# ------------------------------------------------------------------------------------------------
class _RunContext:
    def __init__(self, feed):
        self.feed = feed
        self.cache = {}


def _walk(obj):
    """All Tensor nodes inside a nested argument structure."""
    if isinstance(obj, Tensor):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _walk(o)
    elif isinstance(obj, dict):
        for o in obj.values():
            yield from _walk(o)
    elif isinstance(obj, slice):
        yield from _walk([obj.start, obj.stop, obj.step])


def _resolve(obj, ctx, as_python=False):
    """Replace Tensor nodes in a nested argument structure by their values."""
    if isinstance(obj, Tensor):
        v = obj._eval(ctx)
        if as_python and isinstance(v, torch.Tensor) and v.dim() == 0 and not v.dtype.is_floating_point:
            return int(v)
        return v
    if isinstance(obj, list):
        return [_resolve(o, ctx, as_python) for o in obj]
    if isinstance(obj, tuple):
        return tuple(_resolve(o, ctx, as_python) for o in obj)
    if isinstance(obj, slice):
        return slice(_resolve(obj.start, ctx, True), _resolve(obj.stop, ctx, True), _resolve(obj.step, ctx, True))
    return obj


def _const(x, like=None):
    """Python / numpy value -> torch value with TF's weak typing (a python scalar takes the other operand's dtype)."""
    if isinstance(x, torch.Tensor):
        return x
    if isinstance(x, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(x))
        if t.dtype == torch.int32:
            t = t.long()
        return t
    if isinstance(x, np.generic):
        return _const(np.asarray(x))
    if like is not None and isinstance(like, torch.Tensor):
        return torch.tensor(x, dtype=like.dtype)
    return torch.tensor(x, dtype=torch.float32 if isinstance(x, float) else torch.int64)


class Tensor:
    def __init__(self, fn, args=(), kwargs=None, name=None, op="op"):
        self._fn, self._args, self._kwargs = fn, args, kwargs or {}
        self.op_type = op
        self.name = name or get_default_graph().unique_name(op)
        self.inputs = list(_walk([args, self._kwargs]))

    # evaluation ------------------------------------------------------------------------------
    def _eval(self, ctx):
        key = id(self)
        if key not in ctx.cache:
            # iterative post-order so an 8-step x 4-type graph never hits the recursion limit
            stack = [self]
            while stack:
                node = stack[-1]
                pending = [i for i in node.inputs if id(i) not in ctx.cache]
                if pending:
                    stack.extend(pending)
                    continue
                stack.pop()
                if id(node) not in ctx.cache:
                    ctx.cache[id(node)] = node._compute(ctx)
        return ctx.cache[key]

    def _compute(self, ctx):
        args = [_resolve(a, ctx) for a in self._args]
        kwargs = {k: _resolve(v, ctx) for k, v in self._kwargs.items()}
        return self._fn(*args, **kwargs)

    def __hash__(self):
        return id(self)

    def __eq__(self, other):
        return self is other

    def __repr__(self):
        return "<shim tf.Tensor %s>" % self.name

    # python operators (TF overloads: math_ops.add / subtract / multiply / truediv / neg) ---------
    def __add__(self, o):
        return _binary(torch.add, self, o, "add")

    def __radd__(self, o):
        return _binary(torch.add, o, self, "add")

    def __sub__(self, o):
        return _binary(torch.sub, self, o, "sub")

    def __rsub__(self, o):
        return _binary(torch.sub, o, self, "sub")

    def __mul__(self, o):
        return _binary(torch.mul, self, o, "mul")

    def __rmul__(self, o):
        return _binary(torch.mul, o, self, "mul")

    def __truediv__(self, o):
        return _binary(torch.div, self, o, "truediv")

    def __rtruediv__(self, o):
        return _binary(torch.div, o, self, "truediv")

    def __neg__(self):
        return Tensor(torch.neg, (self,), op="neg")

    def __getitem__(self, idx):
        return _IndexTensor(None, (self, idx), op="strided_slice")


class _IndexTensor(Tensor):
    def _compute(self, ctx):
        x = _resolve(self._args[0], ctx)
        idx = _resolve(self._args[1], ctx, as_python=True)
        return x[idx]


def _binary(fn, a, b, op):
    def run(x, y):
        if not isinstance(x, torch.Tensor):
            x = _const(x, like=y)
        if not isinstance(y, torch.Tensor):
            y = _const(y, like=x)
        if x.dtype != y.dtype and x.dtype.is_floating_point != y.dtype.is_floating_point:
            raise TypeError("shim: %s of %s and %s (TF would refuse the implicit cast)" % (op, x.dtype, y.dtype))
        return fn(x, y)
    return Tensor(run, (a, b), op=op)


class Placeholder(Tensor):
    def __init__(self, dtype, shape=None, name=None):
        super().__init__(None, (), name=get_default_graph().unique_name(name or "Placeholder") + ":0", op="placeholder")
        self.dtype, self.shape = dtype, shape

    def _compute(self, ctx):
        if self not in ctx.feed:
            raise ValueError("You must feed a value for placeholder tensor '%s'" % self.name)
        v = ctx.feed[self]
        return torch.as_tensor(np.asarray(v), dtype=self.dtype.torch)


def placeholder(dtype, shape=None, name=None):
    return Placeholder(dtype, shape, name)


class Variable(Tensor):
    """tf.Variable(initial_value, name=...).  Initialised at construction from the numpy initial value (the
    reference draws those with np.random at graph-build time, utils.py:29-32,64-65, so the values are what
    ``global_variables_initializer`` would assign)."""

    def __init__(self, initial_value, name=None, dtype=None, trainable=True):
        g = get_default_graph()
        super().__init__(None, (), name=g.unique_name(name or "Variable") + ":0", op="variable")
        v = _const(initial_value) if not isinstance(initial_value, torch.Tensor) else initial_value
        if v.dtype == torch.float64:
            v = v.float() if dtype in (None, float32) else v
        self.value = v.clone().requires_grad_(v.dtype.is_floating_point)
        self.dtype = _dtype_of(self.value)
        g.collections[GraphKeys.GLOBAL_VARIABLES].append(self)
        if trainable:
            g.collections[GraphKeys.TRAINABLE_VARIABLES].append(self)

    def _compute(self, ctx):
        return self.value

    def assign(self, value):
        def run(v):
            with torch.no_grad():
                self.value.copy_(torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v,
                                                 dtype=self.value.dtype).reshape(self.value.shape))
            return None
        return Tensor(run, (value,), op="assign")

    def get_shape(self):
        return tuple(self.value.shape)


def _noop(*deps):
    return Tensor(lambda *a: None, tuple(deps), op="no_op")


def global_variables_initializer():
    return _noop()


def local_variables_initializer():
    return _noop()


def variables_initializer(var_list, name=None):
    return _noop()


def group(*ops):
    return _noop(*ops)


# ------------------------------------------------------------------------------------------------
# ops (tensorflow/python/ops/{array_ops,math_ops,nn_ops}.py @ r1.3)
# ------------------------------------------------------------------------------------------------
def _f(x):
    return x if isinstance(x, torch.Tensor) else _const(x)


def shape(x, out_type=int32, name=None):
    return Tensor(lambda v: torch.tensor(list(v.shape), dtype=torch.int64), (x,), op="shape")


def reshape(x, shp, name=None):
    t = Tensor(None, (x, shp), op="reshape")

    def compute(ctx):
        v = _resolve(x, ctx) if isinstance(x, Tensor) else _f(x)
        s = _resolve(shp, ctx, as_python=True)
        if isinstance(s, torch.Tensor):
            s = [int(i) for i in s]
        return v.reshape([int(i) for i in s])
    t._compute = compute
    return t


def concat(values, axis, name=None):
    t = Tensor(lambda vs: torch.cat([_f(v) for v in vs], dim=axis), (list(values),), op="concat")
    t._axis = axis
    return t


def transpose(x, perm=None, name=None):
    return Tensor(lambda v: v.permute(*perm) if perm is not None else v.t(), (x,), op="transpose")


def matmul(a, b, name=None):
    return Tensor(lambda x, y: torch.matmul(_f(x), _f(y)), (a, b), op="MatMul")


def ones_like(x, dtype=None, name=None):
    return Tensor(lambda v: torch.ones_like(v, dtype=dtype.torch if dtype else v.dtype), (x,), op="ones_like")


def zeros_like(x, dtype=None, name=None):
    return Tensor(lambda v: torch.zeros_like(v, dtype=dtype.torch if dtype else v.dtype), (x,), op="zeros_like")


def expand_dims(x, axis, name=None):
    return Tensor(lambda v: v.unsqueeze(axis), (x,), op="ExpandDims")


def squeeze(x, axis=None, name=None):
    return Tensor(lambda v: v.squeeze() if axis is None else v.squeeze(axis), (x,), op="Squeeze")


def reduce_sum(x, axis=None, keep_dims=False, name=None):
    def run(v):
        if isinstance(v, (list, tuple)):                 # tf.reduce_sum(list of scalars) packs them first
            v = torch.stack([_f(i) for i in v])
        return v.sum() if axis is None else v.sum(dim=axis, keepdim=keep_dims)
    return Tensor(run, (x,), op="Sum")


def identity(x, name=None):
    return Tensor(lambda v: v, (x,), op="Identity")


def _unary(fn, op):
    def build(x, name=None):
        return Tensor(lambda v: fn(_f(v)), (x,), op=op)
    return build


abs = _unary(torch.abs, "Abs")                           # noqa: A001 (mirrors tf.abs)
square = _unary(torch.square, "Square")
exp = _unary(torch.exp, "Exp")
sqrt = _unary(torch.sqrt, "Sqrt")
sigmoid = _unary(torch.sigmoid, "Sigmoid")
tanh = _unary(torch.tanh, "Tanh")


def einsum(equation, *inputs):
    return Tensor(lambda *v: torch.einsum(equation, *v), tuple(inputs), op="einsum")


def gather(params, indices, name=None):
    return Tensor(lambda p, i: p[i], (params, indices), op="Gather")


def unsorted_segment_sum(data, segment_ids, num_segments, name=None):
    """core/kernels/segment_reduction_ops.cc (UnsortedSegmentSumFunctor<CPUDevice>): output zeroed, then
    ``output[segment_ids[i]] += data[i]`` for i ascending."""
    t = Tensor(None, (data, segment_ids, num_segments), op="UnsortedSegmentSum")

    def compute(ctx):
        d, ids = _resolve(data, ctx), _resolve(segment_ids, ctx)
        n = _resolve(num_segments, ctx, as_python=True)
        n = int(n)
        out = torch.zeros((n,) + tuple(d.shape[1:]), dtype=d.dtype)
        return out.index_add(0, ids, d)
    t._compute = compute
    return t


def unsorted_segment_max(data, segment_ids, num_segments, name=None):
    """UnsortedSegmentMax: segments without entries get the dtype's lowest value."""
    t = Tensor(None, (data, segment_ids, num_segments), op="UnsortedSegmentMax")

    def compute(ctx):
        d, ids = _resolve(data, ctx), _resolve(segment_ids, ctx)
        n = int(_resolve(num_segments, ctx, as_python=True))
        out = torch.full((n,) + tuple(d.shape[1:]), torch.finfo(d.dtype).min, dtype=d.dtype)
        idx = ids.reshape([-1] + [1] * (d.dim() - 1)).expand_as(d) if d.dim() > 1 else ids
        return out.scatter_reduce(0, idx, d, reduce="amax", include_self=True)
    t._compute = compute
    return t


def clip_by_norm(t, clip_norm, axes=None, name=None):
    """clip_ops.py:84-94 @ r1.3: t * clip_norm * min(rsqrt(sum(t*t)), 1/clip_norm)."""
    def run(v):
        l2norm_inv = torch.rsqrt((v * v).sum())
        c = torch.tensor(clip_norm, dtype=v.dtype)
        return (v * c) * torch.minimum(l2norm_inv, torch.tensor(1.0, dtype=v.dtype) / c)
    return Tensor(run, (t,), op="clip_by_norm")


def _dropout(x, keep_prob, noise_shape=None, seed=None, name=None):
    """nn_ops.py:1887-1954 @ r1.3: ``x / keep_prob * floor(keep_prob + uniform[0,1))`` -- evaluated even for
    keep_prob == 1 (a placeholder is never short-circuited), where it is the exact identity."""
    def run(v, keep):
        keep = _f(keep).to(v.dtype)
        binary = torch.floor(keep + torch.rand(v.shape, dtype=v.dtype))
        return v / keep * binary
    return Tensor(run, (x, keep_prob), op="dropout")


def _glorot_uniform(shp, dtype=torch.float32):
    """init_ops.py glorot_uniform_initializer = VarianceScaling(1.0, fan_avg, uniform): U(-l, l), l = sqrt(6/(fi+fo)).
    (The default initializer of ``tf.get_variable``, which GRUCell's _linear kernels use.)  The draw comes from
    torch's generator -- TF's Philox stream is not reproducible here; golden files store these weights."""
    fan_in, fan_out = shp[0], shp[1]
    limit = float(np.sqrt(6.0 / (fan_in + fan_out)))
    return (torch.rand(shp, dtype=dtype) * 2 - 1) * limit


class _LayerRNNCell:
    """rnn_cell_impl.RNNCell @ r1.3 is a tf.layers.Layer: the variable scope is captured at the FIRST call
    (``<current scope>/<cell name>``) and every later call reuses those variables."""
    cell_scope_name = "rnn_cell"

    def __init__(self):
        self._vars = None

    def _build_once(self, input_size):
        if self._vars is None:
            with variable_scope(self.cell_scope_name):
                self._vars = self.build(input_size)
        return self._vars


class GRUCell(_LayerRNNCell):
    """rnn_cell_impl.py:263-310 @ r1.3 (tf.nn.rnn_cell.GRUCell is tf.contrib.rnn.GRUCell):
        gates:     [r, u] = split(sigmoid(_linear([x, h], 2n, bias=1.0)), 2)      (r FIRST, then u)
        candidate: c = act(_linear([x, r*h], n, bias=0.0))
        new_h = u*h + (1-u)*c
    _linear (rnn_cell_impl.py:1024-1080): concat(args, 1) @ kernel, then bias_add."""
    cell_scope_name = "gru_cell"

    def __init__(self, num_units, activation=None, reuse=None, kernel_initializer=None, bias_initializer=None):
        super().__init__()
        self.n, self.act = num_units, activation or nn.tanh

    def build(self, input_size):
        n = self.n
        with variable_scope("gates"):
            kg = Variable(_glorot_uniform((input_size + n, 2 * n)), name="kernel")
            bg = Variable(np.ones(2 * n, dtype=np.float32), name="bias")
        with variable_scope("candidate"):
            kc = Variable(_glorot_uniform((input_size + n, n)), name="kernel")
            bc = Variable(np.zeros(n, dtype=np.float32), name="bias")
        return kg, bg, kc, bc

    def __call__(self, inputs, state, scope=None):
        # variables have to exist at graph-construction time (collections, optimizer): the input width is taken from
        # the static shapes the reference passes around -- see _static_width below.
        self._build_once(_static_width(inputs))
        kg, bg, kc, bc = self._vars
        n = self.n
        value = Tensor(lambda x, h, k, b: torch.sigmoid(torch.cat([x, h], dim=1).matmul(k) + b),
                       (inputs, state, kg, bg), op="gru_gates")
        r = Tensor(lambda v: v[:, :n], (value,), op="split_r")
        u = Tensor(lambda v: v[:, n:], (value,), op="split_u")
        rh = r * state
        pre = Tensor(lambda x, s, k, b: torch.cat([x, s], dim=1).matmul(k) + b, (inputs, rh, kc, bc), op="gru_candidate")
        c = self.act(pre)
        new_h = u * state + (1 - u) * c
        _set_width(new_h, n)
        return new_h, new_h


class BasicRNNCell(_LayerRNNCell):
    """rnn_cell_impl.py:232-260 @ r1.3: output = act(_linear([x, h], n, bias=0))."""
    cell_scope_name = "basic_rnn_cell"

    def __init__(self, num_units, activation=None, reuse=None):
        super().__init__()
        self.n, self.act = num_units, activation or nn.tanh

    def build(self, input_size):
        return (Variable(_glorot_uniform((input_size + self.n, self.n)), name="kernel"),
                Variable(np.zeros(self.n, dtype=np.float32), name="bias"))

    def __call__(self, inputs, state, scope=None):
        self._build_once(_static_width(inputs))
        k, b = self._vars
        out = self.act(Tensor(lambda x, h, kk, bb: torch.cat([x, h], dim=1).matmul(kk) + bb, (inputs, state, k, b),
                              op="basic_rnn"))
        _set_width(out, self.n)
        return out, out


class CudnnCompatibleGRUCell(_LayerRNNCell):
    """contrib/cudnn_rnn/python/ops/cudnn_rnn_ops.py:93-150 @ r1.3: gates as GRUCell; candidate
    c = tanh(x·W_i + b_i + r * (h·W_h + b_h))."""
    cell_scope_name = "cudnn_compatible_gru_cell"

    def __init__(self, num_units, reuse=None, kernel_initializer=None):
        super().__init__()
        self.n = num_units

    def build(self, input_size):
        n = self.n
        with variable_scope("gates"):
            kg = Variable(_glorot_uniform((input_size + n, 2 * n)), name="kernel")
            bg = Variable(np.ones(2 * n, dtype=np.float32), name="bias")
        with variable_scope("candidate"):
            with variable_scope("input_projection"):
                ki = Variable(_glorot_uniform((input_size, n)), name="kernel")
                bi = Variable(np.zeros(n, dtype=np.float32), name="bias")
            with variable_scope("hidden_projection"):
                kh = Variable(_glorot_uniform((n, n)), name="kernel")
                bh = Variable(np.zeros(n, dtype=np.float32), name="bias")
        return kg, bg, ki, bi, kh, bh

    def __call__(self, inputs, state, scope=None):
        self._build_once(_static_width(inputs))
        kg, bg, ki, bi, kh, bh = self._vars
        n = self.n
        value = Tensor(lambda x, h, k, b: torch.sigmoid(torch.cat([x, h], dim=1).matmul(k) + b),
                       (inputs, state, kg, bg), op="gru_gates")
        r = Tensor(lambda v: v[:, :n], (value,), op="split_r")
        u = Tensor(lambda v: v[:, n:], (value,), op="split_u")
        c = Tensor(lambda x, h, rr, a, b, cc, d: torch.tanh(x.matmul(a) + b + rr * (h.matmul(cc) + d)),
                   (inputs, state, r, ki, bi, kh, bh), op="cudnn_candidate")
        new_h = u * state + (1 - u) * c
        _set_width(new_h, n)
        return new_h, new_h


class DropoutWrapper:
    """rnn_cell_impl.py:585-760 @ r1.3: output / state dropout applied unless the keep_prob is the python
    float 1.0 (a placeholder always goes through nn.dropout)."""

    def __init__(self, cell, input_keep_prob=1.0, output_keep_prob=1.0, state_keep_prob=1.0, **kw):
        self.cell, self.state_keep_prob = cell, state_keep_prob

    def __call__(self, inputs, state, scope=None):
        output, new_state = self.cell(inputs, state)
        if not (isinstance(self.state_keep_prob, float) and self.state_keep_prob >= 1.0):
            w = _static_width(new_state)
            new_state = _dropout(new_state, self.state_keep_prob)
            _set_width(new_state, w)
        return output, new_state


# static last-dimension bookkeeping (TF knows static shapes; the shim only needs the feature width of cell inputs)
def _set_width(t, w):
    t._width = w
    return t


def _static_width(t):
    w = getattr(t, "_width", None)
    if w is not None:
        return w
    if isinstance(t, Placeholder) and t.shape and t.shape[-1] is not None:
        return t.shape[-1]
    if isinstance(t, Variable):
        return t.value.shape[-1]
    if t.op_type == "concat":
        if t._axis == 0:
            return _static_width(t._args[0][0])
        return sum(_static_width(v) for v in t._args[0])          # the reference only concatenates on axis 0 or the last
    if t.op_type == "MatMul":
        return _static_width(t._args[1])
    if t.op_type in ("UnsortedSegmentSum", "Gather"):
        return _static_width(t._args[0])
    if t.op_type == "reshape":
        s = t._args[1]
        if isinstance(s, (list, tuple)) and isinstance(s[-1], int) and s[-1] > 0:
            return s[-1]
    if t.op_type == "strided_slice" and isinstance(t._args[0], Variable):
        return t._args[0].value.shape[-1]
    for i in t.inputs:                       # elementwise ops: the width of the first operand that has one
        try:
            return _static_width(i)
        except ValueError:
            continue
    raise ValueError("shim: static width of %r unknown" % t)


# ------------------------------------------------------------------------------------------------
# training (python/training/{optimizer,adam}.py, core/kernels/training_ops.cc @ r1.3)
# ------------------------------------------------------------------------------------------------
def _reachable_variables(t):
    seen, out, stack = set(), set(), [t]
    while stack:
        n = stack.pop()
        if id(n) in seen:
            continue
        seen.add(id(n))
        if isinstance(n, Variable):
            out.add(n)
        stack.extend(n.inputs)
    return out


class AdamOptimizer:
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, name="Adam"):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon
        self.slots = {}
        self.b1_power = self.b2_power = None

    def compute_gradients(self, loss, var_list=None):
        g = get_default_graph()
        var_list = list(var_list if var_list is not None else g.get_collection(GraphKeys.TRAINABLE_VARIABLES))
        connected = _reachable_variables(loss)
        live = [v for v in var_list if v in connected]
        all_grads = Tensor(lambda l, *vs: torch.autograd.grad(l, [v.value for v in live], allow_unused=True),
                           (loss,) + tuple(live), op="gradients")
        out = []
        for v in var_list:
            if v not in connected:
                out.append((None, v))                                # TF: None for variables the loss does not reach
                continue
            i = live.index(v)
            out.append((Tensor(lambda gs, i=i: gs[i].detach(), (all_grads,), op="grad"), v))
        return out

    def apply_gradients(self, grads_and_vars, global_step=None, name=None):
        pairs = [(g_, v) for g_, v in grads_and_vars if g_ is not None]
        first = min((v for _, v in pairs), key=lambda v: v.name)
        base = first.name[:-2]
        # slot variables (m: '<var>/Adam', v: '<var>/Adam_1') and the beta powers are GLOBAL variables, like in TF
        saved_scope = get_default_graph().scope
        get_default_graph().scope = []
        try:
            self.b1_power = Variable(np.float32(self.b1), name="beta1_power", trainable=False)
            self.b2_power = Variable(np.float32(self.b2), name="beta2_power", trainable=False)
            for _, v in pairs:
                self.slots[v] = (Variable(torch.zeros_like(v.value), name=v.name[:-2] + "/Adam", trainable=False),
                                 Variable(torch.zeros_like(v.value), name=v.name[:-2] + "/Adam_1", trainable=False))
        finally:
            get_default_graph().scope = saved_scope

        def run(*grads):
            with torch.no_grad():
                b1p, b2p = self.b1_power.value, self.b2_power.value
                # training_ops.cc ApplyAdam: alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1); v += (g²-v)(1-b2);
                # var -= alpha*m/(sqrt(v)+eps)
                alpha = torch.tensor(self.lr, dtype=torch.float32) * torch.sqrt(1 - b2p) / (1 - b1p)
                for (_, var), g_ in zip(pairs, grads):
                    m, v = self.slots[var]
                    m.value.add_((g_ - m.value) * (1 - self.b1))
                    v.value.add_((g_ * g_ - v.value) * (1 - self.b2))
                    var.value.sub_((m.value * alpha) / (torch.sqrt(v.value) + self.eps))
                b1p.mul_(self.b1)
                b2p.mul_(self.b2)
            return None
        return Tensor(run, tuple(g_ for g_, _ in pairs), op="Adam")


# ------------------------------------------------------------------------------------------------
# session
# ------------------------------------------------------------------------------------------------
class _GpuOptions:
    allow_growth = False


class ConfigProto:
    def __init__(self, **kw):
        self.gpu_options = _GpuOptions()


def _to_numpy(v):
    if v is None:
        return None
    if isinstance(v, torch.Tensor):
        a = v.detach().numpy().copy()
        if a.dtype == np.int64:
            a = a.astype(np.int32)
        return a[()] if a.ndim == 0 else a
    return v


class Session:
    def __init__(self, graph=None, config=None):
        self.graph = graph or get_default_graph()

    def run(self, fetches, feed_dict=None):
        ctx = _RunContext(feed_dict or {})

        def fetch(f):
            if f is None:
                return None
            if isinstance(f, (list, tuple)):
                return [fetch(i) for i in f]
            if isinstance(f, Tensor):
                return _to_numpy(f._eval(ctx))
            raise TypeError("shim: cannot fetch %r" % (f,))
        # TF evaluates every fetched tensor BEFORE state-changing ops they do not depend on become visible only if
        # there is a data dependency; the reference fetches [loss, accuracy, summary, train_step] and the shim's
        # in-order evaluation gives loss/accuracy at the pre-update weights, which is what TF returns as well
        # (the forward values feed the gradient that the update consumes).
        return fetch(fetches)

    def close(self):
        pass


# ------------------------------------------------------------------------------------------------
# namespaces: tf.nn, tf.nn.rnn_cell, tf.train, tf.summary, tf.contrib
# ------------------------------------------------------------------------------------------------
def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _embedding_lookup(params, ids, name=None):
    """embedding_ops.py: a single-shard lookup is array_ops.gather(params, ids)."""
    return gather(params, ids)


rnn_cell = _module(__name__ + ".nn.rnn_cell", GRUCell=GRUCell, BasicRNNCell=BasicRNNCell, DropoutWrapper=DropoutWrapper)
nn = _module(__name__ + ".nn", tanh=tanh, sigmoid=sigmoid, relu=_unary(torch.relu, "Relu"), dropout=_dropout,
             embedding_lookup=_embedding_lookup, rnn_cell=rnn_cell)
train = _module(__name__ + ".train", AdamOptimizer=AdamOptimizer)


class _FileWriter:
    def __init__(self, logdir, graph=None):
        self.logdir = logdir

    def add_summary(self, summary, global_step=None):
        pass

    def close(self):
        pass


summary = _module(__name__ + ".summary", scalar=lambda name, t: _noop(t), merge_all=lambda: _noop(),
                  FileWriter=_FileWriter)


from . import contrib  # noqa: E402,F401  (tf.contrib.rnn.GRUCell, tensorflow.contrib.cudnn_rnn)
