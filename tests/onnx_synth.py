"""Synthetic ONNX graphs shaped like the two frontend assets of the reference (server/model_utils/cosyvoice/cli/frontend.py:92-115), built with the
product's own container writer.  The real `campplus.onnx` / `speech_tokenizer_v3.onnx` are not in the tree; these carry the same operator mix at toy
widths with seeded weights:

* `campplus_like`  — kaldi fbank [1][T][80] -> FCM head (2-D convolutions + BatchNorm + ReLU over [C][80][T]) -> reshape by shape arithmetic -> strided
  TDNN Conv1d -> a CAM dense layer (1x1 / dilated Conv1d, context = global mean + segment AveragePool(ceil_mode) expanded back, sigmoid gate) ->
  statistics pooling (mean, sqrt of clipped variance) -> Gemm + BatchNorm -> embedding.
* `tokenizer_like` — log-mel [1][n_mels][T] -> two Conv1d + GELU (Erf form), stride 2 -> + sliced positional table -> two transformer blocks (decomposed and
  fused LayerNorm, batched MatMul attention with Softmax, FFN) -> FSQ head (Tanh, Round, base-3 digits through a MatMul) -> int64 token ids.
"""
import math

import numpy as np

from flowmirror_hydravox_amd.onnx_graph import Graph, Node


class _B:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.nodes, self.inits, self.n = [], {}, 0

    def name(self, p='t'):
        self.n += 1
        return '%s%d' % (p, self.n)

    def const(self, arr, p='c'):
        n = self.name(p)
        self.inits[n] = np.asarray(arr)
        return n

    def w(self, *shape, scale=None):
        fan = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        return self.const((self.rng.standard_normal(shape) * (scale or 1.0 / math.sqrt(fan))).astype(np.float32), 'w')

    def op(self, op, ins, n_out=1, **attrs):
        outs = [self.name(op.lower()) for _ in range(n_out)]
        self.nodes.append(Node(op, ins, outs, attrs, name=self.name('n')))
        return outs[0] if n_out == 1 else outs

    def i64(self, *vals):
        return self.const(np.asarray(vals, np.int64), 'i')

    def bn(self, x, c):
        r = self.rng
        return self.op('BatchNormalization', [x, self.const((1 + 0.1 * r.standard_normal(c)).astype(np.float32)), self.const((0.1 * r.standard_normal(c)).astype(np.float32)),
                                              self.const((0.1 * r.standard_normal(c)).astype(np.float32)), self.const((0.5 + r.random(c)).astype(np.float32))], epsilon=1e-5)

    def graph(self, inputs, outputs, opset=17):
        return Graph(self.nodes, self.inits, inputs, outputs, opset)


def campplus_like(seed=0, emb=24):
    b = _B(seed)
    x = b.op('Transpose', ['fbank'], perm=[0, 2, 1])                                   # [1][80][T]
    x = b.op('Unsqueeze', [x, b.i64(1)])                                               # [1][1][80][T]
    x = b.op('Relu', [b.bn(b.op('Conv', [x, b.w(8, 1, 3, 3), b.w(8)], kernel_shape=[3, 3], strides=[2, 1], pads=[1, 1, 1, 1]), 8)])       # [1][8][40][T]
    r = b.op('Relu', [b.bn(b.op('Conv', [x, b.w(8, 8, 3, 3)], kernel_shape=[3, 3], strides=[1, 1], pads=[1, 1, 1, 1]), 8)])
    x = b.op('Relu', [b.op('Add', [r, x])])                                            # residual block of the FCM head
    x = b.op('Relu', [b.bn(b.op('Conv', [x, b.w(8, 8, 3, 3), b.w(8)], kernel_shape=[3, 3], strides=[2, 1], pads=[1, 1, 1, 1]), 8)])       # [1][8][20][T]
    t = b.op('Unsqueeze', [b.op('Gather', [b.op('Shape', [x]), b.const(np.asarray(3, np.int64), 'i')], axis=0), b.i64(0)])
    x = b.op('Reshape', [x, b.op('Concat', [b.i64(1), b.i64(-1), t], axis=0)])          # [1][160][T] through shape arithmetic on the host
    x = b.op('Relu', [b.bn(b.op('Conv', [x, b.w(64, 160, 5), b.w(64)], kernel_shape=[5], strides=[2], pads=[2, 2]), 64)])                # [1][64][T/2]
    # CAM dense layer
    h = b.op('Relu', [b.bn(x, 64)])
    h = b.op('Relu', [b.bn(b.op('Conv', [h, b.w(32, 64, 1)], kernel_shape=[1]), 32)])
    y = b.op('Conv', [h, b.w(16, 32, 3)], kernel_shape=[3], dilations=[2], pads=[2, 2])   # local branch
    g = b.op('ReduceMean', [h], axes=[2], keepdims=1)
    seg = b.op('AveragePool', [h], kernel_shape=[10], strides=[10], ceil_mode=1)       # [1][32][ceil(T'/10)]
    shp = b.op('Shape', [seg])
    e = b.op('Expand', [b.op('Unsqueeze', [seg, b.i64(-1)]), b.op('Concat', [shp, b.i64(10)], axis=0)])
    e = b.op('Reshape', [e, b.op('Concat', [b.op('Slice', [shp, b.i64(0), b.i64(2), b.i64(0)]), b.i64(-1)], axis=0)])
    tl = b.op('Slice', [b.op('Shape', [h]), b.i64(2), b.i64(3), b.i64(0)])
    e = b.op('Slice', [e, b.i64(0), tl, b.i64(2)])                                     # seg-pooled context cut back to T'
    ctx = b.op('Add', [g, e])
    m = b.op('Relu', [b.op('Conv', [ctx, b.w(8, 32, 1), b.w(8)], kernel_shape=[1])])
    m = b.op('Sigmoid', [b.op('Conv', [m, b.w(16, 8, 1), b.w(16)], kernel_shape=[1])])
    x = b.op('Concat', [x, b.op('Mul', [y, m])], axis=1)                               # [1][80][T']
    # statistics pooling
    mean = b.op('ReduceMean', [x], axes=[2], keepdims=0)
    var = b.op('Sub', [b.op('ReduceMean', [b.op('Mul', [x, x])], axes=[2], keepdims=0), b.op('Mul', [mean, mean])])
    std = b.op('Sqrt', [b.op('Clip', [var, b.const(np.asarray(1e-4, np.float32)), b.const(np.asarray(1e4, np.float32))])])
    s = b.op('Concat', [mean, std], axis=1)                                            # [1][160]
    o = b.bn(b.op('Gemm', [s, b.w(emb, 160), b.w(emb)], transB=1), emb)
    b.nodes.append(Node('Identity', [o], ['embedding'], {}))
    return b.graph(['fbank'], ['embedding'])


def tokenizer_like(seed=1, n_mels=16, d=32, heads=4, with_length=False):
    """with_length: a second input `mel_len` (int32 [1], as `speech_tokenizer_session.get_inputs()[1]` of the reference) masks the attention keys past
    ceil(mel_len / 2) — Range / Less / Where on the host-resident integer side of the graph"""
    b = _B(seed)
    hd = d // heads

    def gelu(x):
        e = b.op('Erf', [b.op('Div', [x, b.const(np.asarray(math.sqrt(2.0), np.float32))])])
        return b.op('Mul', [b.op('Mul', [x, b.op('Add', [e, b.const(np.asarray(1.0, np.float32))])]), b.const(np.asarray(0.5, np.float32))])

    x = gelu(b.op('Conv', ['mel', b.w(d, n_mels, 3), b.w(d)], kernel_shape=[3], pads=[1, 1]))
    x = gelu(b.op('Conv', [x, b.w(d, d, 3), b.w(d)], kernel_shape=[3], strides=[2], pads=[1, 1]))
    x = b.op('Transpose', [x], perm=[0, 2, 1])                                          # [1][T2][d]
    t2 = b.op('Slice', [b.op('Shape', [x]), b.i64(1), b.i64(2), b.i64(0)])
    x = b.op('Add', [x, b.op('Slice', [b.w(128, d, scale=0.1), b.i64(0), t2, b.i64(0)])])
    for blk in range(2):
        if blk == 0:                                                                    # LayerNorm as exporters of older opsets decompose it
            mu = b.op('ReduceMean', [x], axes=[-1], keepdims=1)
            dx = b.op('Sub', [x, mu])
            var = b.op('ReduceMean', [b.op('Pow', [dx, b.const(np.asarray(2.0, np.float32))])], axes=[-1], keepdims=1)
            n = b.op('Div', [dx, b.op('Sqrt', [b.op('Add', [var, b.const(np.asarray(1e-5, np.float32))])])])
            n = b.op('Add', [b.op('Mul', [n, b.w(d, scale=1.0)]), b.w(d, scale=0.1)])
        else:
            n = b.op('LayerNormalization', [x, b.w(d, scale=1.0), b.w(d, scale=0.1)], axis=-1, epsilon=1e-5)

        def proj(inp):
            p = b.op('Add', [b.op('MatMul', [inp, b.w(d, d)]), b.w(d, scale=0.1)])
            return b.op('Reshape', [p, b.i64(1, -1, heads, hd)])
        q = b.op('Transpose', [proj(n)], perm=[0, 2, 1, 3])
        k = b.op('Transpose', [proj(n)], perm=[0, 2, 3, 1])
        v = b.op('Transpose', [proj(n)], perm=[0, 2, 1, 3])
        sc = b.op('Mul', [b.op('MatMul', [q, k]), b.const(np.asarray(1.0 / math.sqrt(hd), np.float32))])
        if with_length:
            if blk == 0:
                half = b.op('Div', [b.op('Add', [b.op('Cast', ['mel_len'], to=7), b.i64(1)]), b.i64(2)])            # ceil(len / 2) frames after the stride-2 stem
                frames = b.op('Squeeze', [b.op('Slice', [b.op('Shape', [x]), b.i64(1), b.i64(2), b.i64(0)]), b.i64(0)])
                pos = b.op('Range', [b.const(np.asarray(0, np.int64), 'i'), frames, b.const(np.asarray(1, np.int64), 'i')])
                keymask = b.op('Unsqueeze', [b.op('Less', [pos, half]), b.i64(0, 1, 2)])                              # [1][1][1][T2] bool, on the host
            sc = b.op('Where', [keymask, sc, b.const(np.asarray(-1e9, np.float32))])
        a = b.op('MatMul', [b.op('Softmax', [sc], axis=-1), v])                        # [1][heads][T2][hd]
        a = b.op('Reshape', [b.op('Transpose', [a], perm=[0, 2, 1, 3]), b.i64(1, -1, d)])
        x = b.op('Add', [x, b.op('Add', [b.op('MatMul', [a, b.w(d, d)]), b.w(d, scale=0.1)])])
        n2 = b.op('LayerNormalization', [x, b.w(d, scale=1.0), b.w(d, scale=0.1)], axis=-1, epsilon=1e-5)
        f = gelu(b.op('Add', [b.op('MatMul', [n2, b.w(d, 2 * d)]), b.w(2 * d, scale=0.1)]))
        x = b.op('Add', [x, b.op('Add', [b.op('MatMul', [f, b.w(2 * d, d)]), b.w(d, scale=0.1)])])
    # FSQ head: 4 ternary digits per frame -> ids in [0, 81)
    z = b.op('Tanh', [b.op('MatMul', [x, b.w(d, 4, scale=0.5)])])
    dig = b.op('Add', [b.op('Round', [b.op('Mul', [z, b.const(np.asarray(0.999, np.float32))])]), b.const(np.asarray(1.0, np.float32))])
    ids = b.op('MatMul', [dig, b.const(np.asarray([[1.0], [3.0], [9.0], [27.0]], np.float32))])
    ids = b.op('Squeeze', [b.op('Cast', [ids], to=7), b.i64(-1)])
    b.nodes.append(Node('Identity', [ids], ['tokens'], {}))
    b.nodes.append(Node('Identity', [z], ['latent'], {}))
    return b.graph(['mel', 'mel_len'] if with_length else ['mel'], ['tokens', 'latent'])


def per_operator_cases(seed=11):
    """(operator, inputs, attributes) — one node each, evaluated on the device against oracle/onnx_ref.py by tests/test_gpu_ops.py::
    test_onnx_per_operator_cases_vs_oracle: the operators / attributes the two synthetic graphs above do not reach.  float32 and bool inputs are device
    values there, integer inputs stay host values (as shape arithmetic does in a real graph).  Together with the graphs this list IS the executor's covered
    set (onnx_graph.COVERED; tests/test_host_cpu.py checks the two are equal): OnnxRunner refuses a graph that steps outside it."""
    rng = np.random.default_rng(seed)
    xs = rng.standard_normal((3, 5, 17)).astype(np.float32)
    pos = (np.abs(xs) + 0.1).astype(np.float32)
    slope = rng.standard_normal((5, 1)).astype(np.float32)
    small = (1.0 + 0.1 * rng.standard_normal((2, 4, 3))).astype(np.float32)
    i64 = lambda *v: np.asarray(v, np.int64)                                                                             # noqa: E731
    x1, w1, b1 = rng.standard_normal((2, 6, 37)).astype(np.float32), rng.standard_normal((8, 3, 5)).astype(np.float32), rng.standard_normal(8).astype(np.float32)
    x4, w4, b4 = rng.standard_normal((2, 3, 11, 9)).astype(np.float32), rng.standard_normal((4, 3, 3, 3)).astype(np.float32), rng.standard_normal(4).astype(np.float32)
    xa1, wa1 = rng.standard_normal((2, 4, 23)).astype(np.float32), rng.standard_normal((6, 4, 4)).astype(np.float32)
    return [
        ('Exp', [xs], {}), ('Log', [pos], {}), ('Neg', [xs], {}), ('Abs', [xs], {}), ('Floor', [3 * xs], {}), ('Ceil', [3 * xs], {}), ('Reciprocal', [pos], {}),
        ('Softplus', [xs], {}), ('Sin', [xs], {}), ('Cos', [xs], {}), ('Min', [xs, slope], {}), ('Max', [xs, slope, -xs], {}), ('Max', [xs, slope], {}),
        ('Equal', [np.round(xs), np.round(slope)], {}), ('Greater', [xs, slope], {}), ('Not', [xs > 0], {}), ('And', [xs > 0, xs < 0.5], {}), ('Or', [xs > 0.5, xs < -0.5], {}),
        ('Xor', [xs > 0, xs > 0.5], {}), ('LeakyRelu', [xs], {'alpha': 0.2}), ('Gelu', [xs], {}), ('PRelu', [xs, slope], {}), ('Elu', [xs], {'alpha': 0.7}),
        ('HardSigmoid', [xs], {'alpha': 0.3, 'beta': 0.4}), ('Sign', [xs], {}), ('LogSoftmax', [xs], {'axis': 1}), ('Sum', [xs, slope, xs], {}), ('Mean', [xs, slope], {}),
        ('ArgMax', [xs], {'axis': 2, 'keepdims': 0}), ('ArgMin', [xs], {'axis': 1, 'keepdims': 1}),
        ('ReduceSum', [xs, i64(1)], {'keepdims': 0}), ('ReduceMax', [xs], {'axes': [2], 'keepdims': 1}), ('ReduceMin', [xs], {'axes': [0, 2], 'keepdims': 0}),
        ('ReduceL2', [xs], {'axes': [1], 'keepdims': 1}), ('ReduceSumSquare', [xs], {'axes': [2], 'keepdims': 0}), ('ReduceProd', [i64(2, 3, 4)], {'keepdims': 0}),             # (shape arithmetic: an integer, host-resident operand)
        ('GlobalAveragePool', [xs], {}), ('Flatten', [xs], {'axis': 2}), ('Tile', [xs, i64(2, 1, 3)], {}), ('Pad', [xs, i64(0, 1, 2, 0, 0, 3), np.asarray(0.5, np.float32)], {}),
        ('Size', [xs], {}), ('Constant', [], {'value': small}), ('ConstantOfShape', [i64(2, 3)], {'value': np.asarray([1.5], np.float32)}), ('Dropout', [xs], {}),
        ('Mod', [i64(7, -7, 9), i64(3, 3, 4)], {}), ('Split', [xs, i64(4, 6, 7)], {'axis': 2}), ('Cast', [3000.0 * xs], {'to': 10}),
        ('Conv', [x1, w1, b1], dict(kernel_shape=[5], strides=[2], dilations=[3], pads=[4, 7], group=2)),
        ('Conv', [x4, w4, b4], dict(kernel_shape=[3, 3], strides=[1, 1], auto_pad=b'SAME_UPPER')),
        ('Conv', [xa1, wa1], dict(kernel_shape=[4], strides=[2], auto_pad=b'SAME_LOWER')),
        ('AveragePool', [xa1], dict(kernel_shape=[3], strides=[1], auto_pad=b'SAME_UPPER', count_include_pad=0)),
        ('AveragePool', [xa1], dict(kernel_shape=[4], strides=[3], pads=[1, 1], ceil_mode=1, count_include_pad=1)),
    ]


def covered_set():
    """operator -> attribute names reached by the two synthetic graphs and per_operator_cases(): what the device executor is tested on against the oracle"""
    cov = {}
    for g in (campplus_like(), tokenizer_like(), tokenizer_like(with_length=True)):
        for n in g.nodes:
            cov.setdefault(n.op, set()).update(n.attrs.keys())
    for op, _, attrs in per_operator_cases():
        cov.setdefault(op, set()).update(attrs.keys())
    return cov
