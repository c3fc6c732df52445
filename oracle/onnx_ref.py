"""TEST INFRASTRUCTURE (oracle): numpy evaluation of the ONNX operators flowmirror_hydravox_amd/onnx_graph.py executes on the device.

SURVEY.md §8(f) N2 — the reference runs `speech_tokenizer_v3.onnx` and `campplus.onnx` with onnxruntime
(server/model_utils/cosyvoice/cli/frontend.py:92-115).  onnxruntime is a third-party dependency that is neither under /root/reference nor installed here
and the two graphs are assets of the weights repository, so this restates the PUBLISHED operator semantics (github.com/onnx/onnx docs/Operators.md,
default domain, opsets 11-17) one function per operator, in float64 where it matters, and is held against hand-computed cases in
tests/test_oracle_golden.py.  **Parity unpinned** against onnxruntime and against the real graphs.

`run(graph, feeds)` takes any object with .nodes (op, inputs, outputs, attrs), .initializers, .inputs, .outputs, .opset.
"""
import math

import numpy as np


def _axes(node, v, idx):
    if len(v) > idx and v[idx] is not None:
        return tuple(int(a) for a in np.asarray(v[idx]).reshape(-1))
    a = node.attrs.get('axes')
    return None if a is None else tuple(int(x) for x in a)


def _conv(x, w, b, strides, pads, dilations, group):
    """ONNX Conv (Operators.md#Conv): cross-correlation, N-d (1 or 2 spatial axes here), zero padding `pads` = [begins..., ends...]"""
    nd = x.ndim - 2
    xp = np.pad(x.astype(np.float64), [(0, 0), (0, 0)] + [(pads[i], pads[nd + i]) for i in range(nd)])
    B, Cin = x.shape[:2]
    Cout, cg = w.shape[:2]
    ks = w.shape[2:]
    out_sp = [(xp.shape[2 + i] - dilations[i] * (ks[i] - 1) - 1) // strides[i] + 1 for i in range(nd)]
    y = np.zeros([B, Cout] + out_sp, np.float64)
    og = Cout // group
    for g in range(group):
        xs = xp[:, g * cg:(g + 1) * cg]
        ws = w[g * og:(g + 1) * og].astype(np.float64)
        for idx in np.ndindex(*ks):
            sl = tuple(slice(idx[i] * dilations[i], idx[i] * dilations[i] + (out_sp[i] - 1) * strides[i] + 1, strides[i]) for i in range(nd))
            patch = xs[(slice(None), slice(None)) + sl]                     # [B][cg][out...]
            y[:, g * og:(g + 1) * og] += np.einsum('bc...,oc->bo...', patch, ws[(slice(None), slice(None)) + idx])
    if b is not None:
        y += b.reshape([1, -1] + [1] * nd)
    return y.astype(np.float32)


def _avgpool1d(x, k, s, p, ceil_mode, count_include_pad):
    """ONNX AveragePool over the last axis (Operators.md#AveragePool)"""
    T = x.shape[-1]
    num = T + 2 * p - k
    t_out = (math.ceil(num / s) if ceil_mode else num // s) + 1
    if ceil_mode and (t_out - 1) * s >= T + p:
        t_out -= 1
    y = np.zeros(x.shape[:-1] + (t_out,), np.float64)
    for o in range(t_out):
        lo, hi = o * s - p, o * s - p + k
        a, b = max(lo, 0), min(hi, T)
        n = (min(hi, T + p) - lo) if count_include_pad else (b - a)
        y[..., o] = x[..., a:b].sum(-1, dtype=np.float64) / max(n, 1)
    return y.astype(np.float32)


def _erf(x):
    return np.vectorize(math.erf, otypes=[np.float64])(x.astype(np.float64)).astype(np.float32)


def _reshape(x, target):
    dims = [int(d) for d in np.asarray(target).reshape(-1)]
    dims = [x.shape[i] if d == 0 else d for i, d in enumerate(dims)]
    return x.reshape(dims)


def _node(node, v, opset):
    op, at = node.op, node.attrs
    f32 = lambda a: np.asarray(a, np.float32) if np.issubdtype(np.asarray(a).dtype, np.floating) else np.asarray(a)   # noqa: E731
    if op == 'Constant':
        for k in ('value', 'value_float', 'value_int', 'value_ints', 'value_floats'):
            if k in at:
                return f32(at[k])
    if op == 'Shape':
        return np.asarray(v[0].shape, np.int64)
    if op == 'Size':
        return np.asarray(v[0].size, np.int64)
    if op in ('Identity', 'Dropout'):
        return v[0]
    if op == 'Cast':
        to = {1: np.float32, 6: np.int32, 7: np.int64, 9: np.bool_, 11: np.float64, 10: np.float16}[int(at['to'])]
        return (np.trunc(v[0]) if np.issubdtype(to, np.integer) and np.issubdtype(v[0].dtype, np.floating) else v[0]).astype(to)
    two = {'Add': np.add, 'Sub': np.subtract, 'Mul': np.multiply, 'Pow': np.power, 'Max': np.maximum, 'Min': np.minimum, 'Equal': np.equal,
           'Less': np.less, 'Greater': np.greater, 'And': np.logical_and, 'Or': np.logical_or}
    if op in ('Max', 'Min') and len(v) != 2:
        out = v[0]
        for t in v[1:]:
            out = two[op](out, t)
        return out
    if op in two:
        return two[op](v[0], v[1])
    if op == 'Div':
        if np.issubdtype(v[0].dtype, np.integer):
            return np.trunc(v[0] / v[1]).astype(v[0].dtype)
        return (v[0] / v[1]).astype(np.float32)
    one = {'Relu': lambda x: np.maximum(x, 0), 'Sigmoid': lambda x: (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32), 'Tanh': np.tanh, 'Erf': _erf,
           'Sqrt': np.sqrt, 'Exp': np.exp, 'Log': np.log, 'Neg': np.negative, 'Abs': np.abs, 'Round': np.round, 'Floor': np.floor, 'Ceil': np.ceil,
           'Reciprocal': lambda x: 1.0 / x, 'Softplus': lambda x: np.log1p(np.exp(x.astype(np.float64))).astype(np.float32), 'Sin': np.sin, 'Cos': np.cos,
           'Not': np.logical_not}
    if op in one:
        return one[op](v[0])
    if op == 'Xor':
        return np.logical_xor(v[0], v[1])
    if op == 'Mod':                                            # ONNX Mod, fmod = 0 (integer operands): the sign follows the divisor; fmod = 1: C fmod
        return np.fmod(v[0], v[1]) if int(at.get('fmod', 0)) else np.mod(v[0], v[1])
    if op in ('Sum', 'Mean'):
        acc = sum(x.astype(np.float64) for x in v)
        return (acc / len(v) if op == 'Mean' else acc).astype(np.float32)
    if op == 'PRelu':
        return np.where(v[0] > 0, v[0], v[0] * v[1]).astype(np.float32)
    if op == 'Elu':
        return np.where(v[0] > 0, v[0], np.float32(at.get('alpha', 1.0)) * (np.exp(np.minimum(v[0], 0)) - 1)).astype(np.float32)
    if op == 'HardSigmoid':
        return np.clip(np.float32(at.get('alpha', 0.2)) * v[0] + np.float32(at.get('beta', 0.5)), 0, 1).astype(np.float32)
    if op == 'Sign':
        return np.sign(v[0])
    if op == 'LogSoftmax':
        x = v[0].astype(np.float64)
        ax = int(at.get('axis', -1))
        z = x - x.max(ax, keepdims=True)
        return (z - np.log(np.exp(z).sum(ax, keepdims=True))).astype(np.float32)
    if op in ('ArgMax', 'ArgMin'):
        fn = np.argmax if op == 'ArgMax' else np.argmin
        r = fn(v[0], axis=int(at.get('axis', 0)))
        return (np.expand_dims(r, int(at.get('axis', 0))) if int(at.get('keepdims', 1)) else r).astype(np.int64)
    if op == 'Split':
        ax = int(at.get('axis', 0))
        sizes = [int(t) for t in (v[1] if len(v) > 1 and v[1] is not None else at.get('split', []))]
        if not sizes:
            k = len(node.outputs)
            sizes = [-(-v[0].shape[ax] // k)] * k
            sizes[-1] = v[0].shape[ax] - sum(sizes[:-1])
        return list(np.split(v[0], np.cumsum(sizes)[:-1], axis=ax))
    if op == 'Gelu':
        return (0.5 * v[0] * (1.0 + _erf(v[0] / np.float32(math.sqrt(2.0))))).astype(np.float32)
    if op == 'Clip':
        lo = v[1] if len(v) > 1 and v[1] is not None else at.get('min', -np.inf)
        hi = v[2] if len(v) > 2 and v[2] is not None else at.get('max', np.inf)
        return np.clip(v[0], np.asarray(lo).reshape(-1)[0], np.asarray(hi).reshape(-1)[0])
    if op == 'LeakyRelu':
        return np.where(v[0] > 0, v[0], v[0] * np.float32(at.get('alpha', 0.01)))
    if op == 'Where':
        return np.where(v[0], v[1], v[2])
    if op == 'Reshape':
        return _reshape(v[0], v[1])
    if op == 'Flatten':
        ax = int(at.get('axis', 1))
        return v[0].reshape(int(np.prod(v[0].shape[:ax])) if ax else 1, -1)
    if op == 'Transpose':
        return np.transpose(v[0], at.get('perm'))
    if op == 'Unsqueeze':
        out = v[0]
        for a in sorted(_axes(node, v, 1)):
            out = np.expand_dims(out, a if a >= 0 else a + out.ndim + 1)
        return out
    if op == 'Squeeze':
        ax = _axes(node, v, 1)
        return np.squeeze(v[0], axis=ax)
    if op == 'Concat':
        return np.concatenate([np.atleast_1d(x) for x in v], axis=int(at['axis']))
    if op == 'Slice':
        x = v[0]
        if len(v) > 1:
            starts, ends = np.asarray(v[1]).reshape(-1), np.asarray(v[2]).reshape(-1)
            axes = np.asarray(v[3]).reshape(-1) if len(v) > 3 and v[3] is not None else np.arange(len(starts))
            steps = np.asarray(v[4]).reshape(-1) if len(v) > 4 and v[4] is not None else np.ones(len(starts), np.int64)
        else:
            starts, ends = np.asarray(at['starts']), np.asarray(at['ends'])
            axes, steps = np.asarray(at.get('axes', list(range(len(starts))))), np.ones(len(starts), np.int64)
        sl = [slice(None)] * x.ndim
        for s, e, a, st in zip(starts, ends, axes, steps):
            s, e, st = int(s), int(e), int(st)
            sl[int(a)] = slice(s, None if (st < 0 and e < -x.shape[int(a)]) else e, st)
        return x[tuple(sl)]
    if op == 'Gather':
        return np.take(v[0], np.asarray(v[1]).astype(np.int64), axis=int(at.get('axis', 0)))
    if op == 'Expand':
        return v[0] * np.ones([int(s) for s in v[1]], dtype=v[0].dtype)
    if op == 'Tile':
        return np.tile(v[0], [int(r) for r in v[1]])
    if op == 'Range':
        return np.arange(v[0].item(), v[1].item(), v[2].item()).astype(v[0].dtype)
    if op == 'ConstantOfShape':
        val = at.get('value')
        val = np.zeros(1, np.float32) if val is None else np.asarray(val).reshape(-1)
        return np.full([int(x) for x in v[0]], val[0], dtype=val.dtype)
    if op == 'Pad':
        pads = [int(p) for p in (v[1] if len(v) > 1 and v[1] is not None else at['pads'])]
        val = np.asarray(v[2]).reshape(-1)[0] if len(v) > 2 and v[2] is not None else at.get('value', 0.0)
        r = v[0].ndim
        return np.pad(v[0], [(pads[i], pads[r + i]) for i in range(r)], constant_values=val)
    red = {'ReduceMean': np.mean, 'ReduceSum': np.sum, 'ReduceMax': np.max, 'ReduceMin': np.min, 'ReduceProd': np.prod}
    if op in red:
        ax = _axes(node, v, 1)
        x = v[0].astype(np.float64) if np.issubdtype(v[0].dtype, np.floating) else v[0]
        return red[op](x, axis=ax, keepdims=bool(at.get('keepdims', 1))).astype(v[0].dtype)
    if op in ('ReduceL2', 'ReduceSumSquare'):
        ax = _axes(node, v, 1)
        s = np.sum(v[0].astype(np.float64) ** 2, axis=ax, keepdims=bool(at.get('keepdims', 1)))
        return (np.sqrt(s) if op == 'ReduceL2' else s).astype(np.float32)
    if op == 'GlobalAveragePool':
        return v[0].mean(axis=tuple(range(2, v[0].ndim)), keepdims=True, dtype=np.float64).astype(np.float32)
    if op == 'Softmax':
        x = v[0].astype(np.float64)
        ax = int(at.get('axis', 1 if opset < 13 else -1)) % x.ndim          # (ONNX: the default axis is 1 before opset 13)
        if opset < 13 and ax != x.ndim - 1:
            shp = x.shape
            x2 = x.reshape(int(np.prod(shp[:ax])), -1)
            e = np.exp(x2 - x2.max(-1, keepdims=True))
            return (e / e.sum(-1, keepdims=True)).reshape(shp).astype(np.float32)
        e = np.exp(x - x.max(ax, keepdims=True))
        return (e / e.sum(ax, keepdims=True)).astype(np.float32)
    if op == 'LayerNormalization':
        x = v[0].astype(np.float64)
        ax = tuple(range(int(at.get('axis', -1)) % x.ndim, x.ndim))
        mu = x.mean(ax, keepdims=True)
        var = ((x - mu) ** 2).mean(ax, keepdims=True)
        y = (x - mu) / np.sqrt(var + float(at.get('epsilon', 1e-5))) * v[1]
        return (y + v[2] if len(v) > 2 and v[2] is not None else y).astype(np.float32)
    if op == 'BatchNormalization':
        x = v[0].astype(np.float64)
        shp = [1, -1] + [1] * (x.ndim - 2)
        sc, bi, mu, var = [p.astype(np.float64).reshape(shp) for p in v[1:5]]
        return ((x - mu) / np.sqrt(var + float(at.get('epsilon', 1e-5))) * sc + bi).astype(np.float32)
    if op == 'AveragePool':
        k, st = at['kernel_shape'], at.get('strides', [1])
        pads = _pads(at, v[0].shape[2:], k, st, [1])
        return _avgpool1d(v[0], int(k[0]), int(st[0]), int(pads[0]), int(at.get('ceil_mode', 0)), int(at.get('count_include_pad', 0)))
    if op == 'MatMul':
        return np.matmul(v[0].astype(np.float64), v[1].astype(np.float64)).astype(np.float32)
    if op == 'Gemm':
        a = v[0].T if int(at.get('transA', 0)) else v[0]
        b = v[1].T if int(at.get('transB', 0)) else v[1]
        y = float(at.get('alpha', 1.0)) * (a.astype(np.float64) @ b.astype(np.float64))
        if len(v) > 2 and v[2] is not None:
            y = y + float(at.get('beta', 1.0)) * v[2]
        return y.astype(np.float32)
    if op == 'Conv':
        nd = v[0].ndim - 2
        st, dil = at.get('strides', [1] * nd), at.get('dilations', [1] * nd)
        return _conv(v[0], v[1], v[2] if len(v) > 2 else None, st, _pads(at, v[0].shape[2:], v[1].shape[2:], st, dil), dil, int(at.get('group', 1)))
    raise NotImplementedError('oracle: ONNX operator %s' % op)


def _pads(at, sizes, kernel, strides, dilations):
    """ONNX auto_pad (operator spec: Conv / AveragePool): NOTSET -> `pads`; VALID -> 0; SAME_*: out = ceil(in / stride), odd element at the end (UPPER) / start (LOWER)"""
    mode = at.get('auto_pad', b'NOTSET')
    mode = mode.decode() if isinstance(mode, (bytes, bytearray)) else str(mode)
    n = len(kernel)
    if mode in ('', 'NOTSET'):
        return list(at.get('pads', [0] * (2 * n)))
    if mode == 'VALID':
        return [0] * (2 * n)
    lo, hi = [], []
    for size, k, st, d in zip(sizes, kernel, strides, dilations):
        total = max((-(-size // st) - 1) * st + (k - 1) * d + 1 - size, 0)
        a, b = total // 2, total - total // 2
        lo.append(a if mode == 'SAME_UPPER' else b)
        hi.append(b if mode == 'SAME_UPPER' else a)
    return lo + hi


def run(graph, feeds):
    vals = dict(graph.initializers)
    for name in graph.inputs:
        vals[name] = np.asarray(feeds[name])
    for node in graph.nodes:
        ins = [vals[i] if i else None for i in node.inputs]
        out = _node(node, ins, graph.opset)
        outs = out if isinstance(out, (list, tuple)) else [out]
        for name, o in zip(node.outputs, outs):
            if name:
                vals[name] = np.asarray(o)
    return {o: vals[o] for o in graph.outputs}
