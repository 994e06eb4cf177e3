"""Test oracle for the non-default switches of chem_tensorflow_sparse.py (attention :147-149,170-196; BasicRNNCell / CudnnCompatibleGRUCell
:105-110): ONE timestep restated in differentiable torch ops (reference op order) and differentiated by torch autograd.
`install(variants)` routes VariantStepFn.backward through it (variants.BACKWARD_ORACLE); the product package holds no torch
restatement of the path."""
import torch

SMALL_NUMBER = 1e-7


def _activation(name: str):
    return torch.tanh if name.lower() == "tanh" else torch.relu


def step_torch(h, index, nin, edge_weights, edge_biases, attention_weights, use_avg, residuals, cell_type, cell, activation):
    """One timestep of chem_tensorflow_sparse.py:153-216 in differentiable torch ops (reference op order)."""
    V, D = h.shape
    T = edge_weights.shape[0]
    src, dst = index.adj[:, 0].long(), index.adj[:, 1].long()
    off = index.type_off
    etype = torch.cat([torch.full((off[t + 1] - off[t],), t, dtype=torch.long, device=h.device) for t in range(T)]) \
        if index.num_messages else torch.zeros(0, dtype=torch.long, device=h.device)
    H = torch.einsum('vd,tde->vte', h, edge_weights)                      # :160-164 for every type at once
    messages = H[src, etype]                                              # [M, D], type-major like :168
    if attention_weights is not None:                                     # :147-149, 170-196
        scores = (h[src] * h[dst]).sum(-1) * attention_weights[etype]
        smax = torch.full((V,), torch.finfo(h.dtype).min, dtype=h.dtype, device=h.device)
        smax = smax.scatter_reduce(0, dst, scores, reduce="amax", include_self=True)
        exped = torch.exp(scores - smax[dst])
        ssum = torch.zeros(V, dtype=h.dtype, device=h.device).index_add(0, dst, exped)
        messages = messages * (exped / (ssum[dst] + SMALL_NUMBER)).unsqueeze(-1)
    incoming = torch.zeros_like(h).index_add(0, dst, messages)            # :198-200
    if edge_biases is not None:
        incoming = incoming + nin.matmul(edge_biases)                     # :202-204
    if use_avg:
        incoming = incoming / (nin.sum(dim=-1, keepdim=True) + SMALL_NUMBER)   # :206-209
    x = torch.cat(list(residuals) + [incoming], dim=-1)                   # :211-212
    if cell_type == 'rnn':                                                # BasicRNNCell: act([x,h] W + b)
        kernel, bias = cell
        return _activation(activation)(torch.cat([x, h], dim=1).matmul(kernel) + bias)
    gates = torch.sigmoid(torch.cat([x, h], dim=1).matmul(cell[0]) + cell[1])
    r, u = gates[:, :D], gates[:, D:]                                     # r first, then u
    if cell_type == 'gru':
        c = _activation(activation)(torch.cat([x, r * h], dim=1).matmul(cell[2]) + cell[3])
    else:                                                                 # CudnnCompatibleGRUCell
        c = torch.tanh(x.matmul(cell[2]) + cell[3] + r * (h.matmul(cell[4]) + cell[5]))
    return u * h + (1 - u) * c


def autograd_backward(ctx, g, h, nin, W, bias, attn, cell, residuals):
    """VariantStepFn.backward derived by torch autograd from step_torch (same return tuple)."""
    leaves = [t.detach().requires_grad_(True) for t in [h, W] + ([bias] if bias is not None else []) +
              ([attn] if attn is not None else []) + list(cell) + list(residuals)]
    it = iter(leaves)
    h_, W_ = next(it), next(it)
    bias_ = next(it) if bias is not None else None
    attn_ = next(it) if attn is not None else None
    cell_ = [next(it) for _ in range(ctx.num_cell)]
    res_ = [next(it) for _ in range(ctx.num_res)]
    with torch.enable_grad():
        out = step_torch(h_, ctx.index, nin, W_, bias_, attn_, ctx.use_avg, res_, ctx.cell_type, cell_, ctx.activation)
    grads = list(torch.autograd.grad(out, leaves, g, allow_unused=True))
    it = iter(grads)
    dh, dW = next(it), next(it)
    dbias = next(it) if bias is not None else None
    dattn = next(it) if attn is not None else None
    dcell = [next(it) for _ in range(ctx.num_cell)]
    dres = [next(it) for _ in range(ctx.num_res)]
    return (dh, None, None, None, None, None, None, None, dW, dbias, dattn, *dcell, *dres)
