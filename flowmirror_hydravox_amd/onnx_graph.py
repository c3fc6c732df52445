"""ONNX graphs on the device.

The zero-shot frontend of the reference feeds the prompt audio through two ONNX graphs with onnxruntime on the CPU
(server/model_utils/cosyvoice/cli/frontend.py:92-115): `speech_tokenizer_v3.onnx` (whisper log-mel -> speech tokens) and `campplus.onnx`
(kaldi fbank -> 192-d speaker embedding).  The graphs are assets of the weights repository and are not in the tree; this module is what can
be built without them — a reader for the ONNX container (protobuf wire format, no `onnx` package needed) and an executor that runs a graph's
floating-point operators in libhvx:

    Conv (1-D), MatMul, Gemm                       -> hvx_op_gemm, the exact-fp32 MFMA GEMM / implicit-GEMM convolution of the hot path
    Conv (2-D)                                     -> hvx_conv2d
    Add Sub Mul Div Pow Max Min Relu Sigmoid Tanh Erf Sqrt Exp Log Neg Abs Round Floor Ceil Reciprocal Clip LeakyRelu Softplus Sin Cos
    Equal Less Greater Where Not, Transpose Slice Expand Tile and the pieces of Concat / Pad
                                                   -> hvx_nd_elementwise (strided, broadcasting; data movement is its COPY operator)
    ReduceMean ReduceSum ReduceMax ReduceMin ReduceL2 GlobalAveragePool, Softmax, LayerNormalization, BatchNormalization
                                                   -> hvx_rows_reduce / hvx_rows_softmax (+ elementwise)
    AveragePool (1-D, ceil_mode)                   -> hvx_avgpool_rows

Integer tensors (shapes, axes, indices: Shape, Gather, Unsqueeze, Concat, Range, ConstantOfShape ... on int64) never leave the host: a node whose
inputs are all host values is evaluated with numpy, as shape arithmetic always is.  torch is used for device memory only (allocation, views,
host <-> device copies and the destination-strided copies of Concat / Pad).

Parity: `oracle/onnx_ref.py` evaluates the same operator set with numpy following the ONNX operator specification; tests/ hold the executor to it on
synthetic graphs shaped like the two assets (tests/onnx_synth.py).  Against the REAL graphs parity is unpinned: they are not available here.
"""
import ctypes as C
import struct

import numpy as np

# ---------------------------------------------------------------------------------------------------------------------------------
# protobuf wire format (https://protobuf.dev/programming-guides/encoding/) — just enough for onnx.proto3
# ---------------------------------------------------------------------------------------------------------------------------------


def _varint(buf, i):
    v, s = 0, 0
    while True:
        b = buf[i]
        i += 1
        v |= (b & 0x7F) << s
        if b < 0x80:
            return v, i
        s += 7


def _fields(buf):
    """[(field number, wire type, value)]: varint -> int, 64-bit / 32-bit -> bytes, length-delimited -> memoryview"""
    buf = memoryview(buf)
    i, n, out = 0, len(buf), []
    while i < n:
        key, i = _varint(buf, i)
        f, w = key >> 3, key & 7
        if w == 0:
            v, i = _varint(buf, i)
        elif w == 1:
            v, i = bytes(buf[i:i + 8]), i + 8
        elif w == 2:
            ln, i = _varint(buf, i)
            v, i = buf[i:i + ln], i + ln
        elif w == 5:
            v, i = bytes(buf[i:i + 4]), i + 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % w)
        out.append((f, w, v))
    return out


def _sint64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_varints(v):
    out, i, b = [], 0, bytes(v)
    while i < len(b):
        x, i = _varint(b, i)
        out.append(_sint64(x))
    return out


_DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 5: np.int16, 6: np.int32, 7: np.int64, 9: np.bool_, 10: np.float16, 11: np.float64}


def _tensor(buf):
    dims, dt, name, raw = [], 1, '', None
    f32, i32, i64, f64 = [], [], [], []
    for f, w, v in _fields(buf):
        if f == 1:
            dims += _packed_varints(v) if w == 2 else [_sint64(v)]
        elif f == 2:
            dt = v
        elif f == 4:
            f32 += list(np.frombuffer(bytes(v), '<f4')) if w == 2 else [struct.unpack('<f', v)[0]]
        elif f == 5:
            i32 += _packed_varints(v) if w == 2 else [_sint64(v)]
        elif f == 7:
            i64 += _packed_varints(v) if w == 2 else [_sint64(v)]
        elif f == 8:
            name = bytes(v).decode()
        elif f == 9:
            raw = bytes(v)
        elif f == 10:
            f64 += list(np.frombuffer(bytes(v), '<f8')) if w == 2 else [struct.unpack('<d', v)[0]]
    if dt not in _DTYPES:
        raise ValueError('tensor %r: unsupported ONNX data type %d' % (name, dt))
    npdt = _DTYPES[dt]
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np.dtype(npdt).newbyteorder('<')).astype(npdt)
    elif dt == 1:
        arr = np.asarray(f32, np.float32)
    elif dt == 11:
        arr = np.asarray(f64, np.float64)
    elif dt == 7:
        arr = np.asarray(i64, np.int64)
    elif dt == 10:
        arr = np.asarray(i32, np.uint16).view(np.float16)
    else:
        arr = np.asarray(i32).astype(npdt)
    return name, arr.reshape(dims)


def _attribute(buf):
    name, val, typ = '', None, 0
    floats, ints, strings = [], [], []
    for f, w, v in _fields(buf):
        if f == 1:
            name = bytes(v).decode()
        elif f == 2:
            val = struct.unpack('<f', v)[0]
        elif f == 3:
            val = _sint64(v)
        elif f == 4:
            val = bytes(v)
        elif f == 5:
            val = _tensor(v)[1]
        elif f == 7:
            floats += list(np.frombuffer(bytes(v), '<f4')) if w == 2 else [struct.unpack('<f', v)[0]]
        elif f == 8:
            ints += _packed_varints(v) if w == 2 else [_sint64(v)]
        elif f == 9:
            strings.append(bytes(v))
        elif f == 20:
            typ = v
    if typ == 6 or (val is None and floats):
        val = [float(x) for x in floats]
    elif typ == 7 or (val is None and ints):
        val = [int(x) for x in ints]
    elif typ == 8:
        val = strings
    elif typ == 7 and val is None:
        val = []
    return name, val


class Node:
    def __init__(self, op, inputs, outputs, attrs, name=''):
        self.op, self.inputs, self.outputs, self.attrs, self.name = op, list(inputs), list(outputs), dict(attrs), name

    def __repr__(self):
        return 'Node(%s %s -> %s)' % (self.op, self.inputs, self.outputs)


class Graph:
    def __init__(self, nodes, initializers, inputs, outputs, opset=17):
        self.nodes, self.initializers, self.inputs, self.outputs, self.opset = nodes, initializers, inputs, outputs, opset


def _node(buf):
    ins, outs, name, op, attrs = [], [], '', '', {}
    for f, w, v in _fields(buf):
        if f == 1:
            ins.append(bytes(v).decode())
        elif f == 2:
            outs.append(bytes(v).decode())
        elif f == 3:
            name = bytes(v).decode()
        elif f == 4:
            op = bytes(v).decode()
        elif f == 5:
            k, val = _attribute(v)
            attrs[k] = val
    return Node(op, ins, outs, attrs, name)


def _value_name(buf):
    for f, w, v in _fields(buf):
        if f == 1:
            return bytes(v).decode()
    return ''


def load_onnx(data):
    """ONNX ModelProto (bytes or a path) -> Graph"""
    if isinstance(data, str):
        with open(data, 'rb') as fh:
            data = fh.read()
    graph, opset = None, 17
    for f, w, v in _fields(data):
        if f == 7:
            graph = v
        elif f == 8:
            dom, ver = '', None
            for f2, w2, v2 in _fields(v):
                if f2 == 1:
                    dom = bytes(v2).decode()
                elif f2 == 2:
                    ver = v2
            if dom in ('', 'ai.onnx') and ver is not None:
                opset = ver
    if graph is None:
        raise ValueError('not an ONNX model: no graph')
    nodes, inits, inputs, outputs = [], {}, [], []
    for f, w, v in _fields(graph):
        if f == 1:
            nodes.append(_node(v))
        elif f == 5:
            name, arr = _tensor(v)
            inits[name] = arr
        elif f == 11:
            inputs.append(_value_name(v))
        elif f == 12:
            outputs.append(_value_name(v))
    return Graph(nodes, inits, [i for i in inputs if i not in inits], outputs, opset)


# ---- writer (tests build synthetic graphs with it; also lets a graph be saved back) ---------------------------------------------------------

def _key(f, w):
    return _enc_varint((f << 3) | w)


def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(f, payload):
    return _key(f, 2) + _enc_varint(len(payload)) + payload


def _enc_tensor(name, arr):
    arr = np.asarray(arr)
    dt = {np.dtype(np.float32): 1, np.dtype(np.int64): 7, np.dtype(np.int32): 6, np.dtype(np.bool_): 9, np.dtype(np.float64): 11}[arr.dtype]
    out = b''.join(_key(1, 0) + _enc_varint(int(d)) for d in arr.shape)
    out += _key(2, 0) + _enc_varint(dt) + _ld(8, name.encode()) + _ld(9, np.ascontiguousarray(arr).astype(arr.dtype.newbyteorder('<')).tobytes())
    return out


def _enc_attr(name, val):
    out = _ld(1, name.encode())
    if isinstance(val, float):
        out += _key(2, 5) + struct.pack('<f', val) + _key(20, 0) + _enc_varint(1)
    elif isinstance(val, (int, np.integer)):
        out += _key(3, 0) + _enc_varint(int(val)) + _key(20, 0) + _enc_varint(2)
    elif isinstance(val, (bytes, str)):
        out += _ld(4, val.encode() if isinstance(val, str) else val) + _key(20, 0) + _enc_varint(3)
    elif isinstance(val, np.ndarray):
        out += _ld(5, _enc_tensor('', val)) + _key(20, 0) + _enc_varint(4)
    elif isinstance(val, (list, tuple)) and val and isinstance(val[0], float):
        out += _ld(7, b''.join(struct.pack('<f', x) for x in val)) + _key(20, 0) + _enc_varint(6)
    elif isinstance(val, (list, tuple)):
        out += _ld(8, b''.join(_enc_varint(int(x)) for x in val)) + _key(20, 0) + _enc_varint(7)
    else:
        raise TypeError('attribute %s: %r' % (name, type(val)))
    return out


def save_onnx(graph):
    """Graph -> ONNX ModelProto bytes (ir_version 8, default-domain opset graph.opset)"""
    g = b''
    for n in graph.nodes:
        nb = b''.join(_ld(1, i.encode()) for i in n.inputs) + b''.join(_ld(2, o.encode()) for o in n.outputs)
        nb += _ld(3, n.name.encode()) + _ld(4, n.op.encode()) + b''.join(_ld(5, _enc_attr(k, v)) for k, v in n.attrs.items())
        g += _ld(1, nb)
    g += _ld(2, b'hvx')
    for name, arr in graph.initializers.items():
        g += _ld(5, _enc_tensor(name, arr))
    for name in graph.inputs:
        g += _ld(11, _ld(1, name.encode()))
    for name in graph.outputs:
        g += _ld(12, _ld(1, name.encode()))
    return _key(1, 0) + _enc_varint(8) + _ld(8, _ld(1, b'') + _key(2, 0) + _enc_varint(graph.opset)) + _ld(7, g)


# ---------------------------------------------------------------------------------------------------------------------------------
# host side: integer / shape arithmetic with numpy
# ---------------------------------------------------------------------------------------------------------------------------------

def _axes_arg(node, vals, idx, name='axes'):
    """`axes`-like argument: an input (opset >= 13) or an attribute"""
    if len(vals) > idx and vals[idx] is not None:
        return [int(a) for a in np.asarray(vals[idx]).reshape(-1)]
    a = node.attrs.get(name)
    return None if a is None else [int(x) for x in a]


def _slice_args(node, vals, rank):
    if len(vals) > 1:
        starts, ends = np.asarray(vals[1]).reshape(-1), np.asarray(vals[2]).reshape(-1)
        axes = np.asarray(vals[3]).reshape(-1) if len(vals) > 3 and vals[3] is not None else np.arange(len(starts))
        steps = np.asarray(vals[4]).reshape(-1) if len(vals) > 4 and vals[4] is not None else np.ones(len(starts), np.int64)
    else:
        starts, ends = np.asarray(node.attrs['starts']), np.asarray(node.attrs['ends'])
        axes = np.asarray(node.attrs.get('axes', list(range(len(starts)))))
        steps = np.ones(len(starts), np.int64)
    return [int(s) for s in starts], [int(e) for e in ends], [int(a) % rank for a in axes], [int(s) for s in steps]


def _resolve_pads(at, sizes, kernel, strides, dilations):
    """ONNX `pads` ([begin..., end...]) of a Conv / pooling node after `auto_pad` (NOTSET: the attribute; VALID: none; SAME_UPPER / SAME_LOWER: output
    = ceil(input / stride), the odd element at the end / at the beginning)"""
    n = len(kernel)
    mode = at.get('auto_pad', 'NOTSET')
    mode = mode.decode() if isinstance(mode, (bytes, bytearray)) else str(mode)
    if mode in ('', 'NOTSET'):
        return [int(p) for p in at.get('pads', [0] * (2 * n))]
    if mode == 'VALID':
        return [0] * (2 * n)
    if mode not in ('SAME_UPPER', 'SAME_LOWER'):
        raise NotImplementedError('auto_pad=%s' % mode)
    begin, end = [], []
    for size, k, st, d in zip(sizes, kernel, strides, dilations):
        out = -(-int(size) // int(st))
        total = max((out - 1) * int(st) + (int(k) - 1) * int(d) + 1 - int(size), 0)
        small, big = total // 2, total - total // 2
        begin.append(small if mode == 'SAME_UPPER' else big)
        end.append(big if mode == 'SAME_UPPER' else small)
    return begin + end


def _host_eval(node, v):
    """numpy evaluation of a node whose inputs are all host arrays (shape arithmetic and constants)"""
    op, at = node.op, node.attrs
    if op == 'Identity':
        return v[0]
    if op == 'Cast':
        return v[0].astype(_DTYPES[int(at['to'])])
    if op in ('Add', 'Sub', 'Mul', 'Div', 'Pow', 'Max', 'Min', 'Equal', 'Less', 'Greater', 'And', 'Or', 'Mod'):
        a, b = v[0], v[1]
        if op == 'Div' and np.issubdtype(a.dtype, np.integer):
            return (np.trunc(a / b)).astype(a.dtype)
        return {'Add': np.add, 'Sub': np.subtract, 'Mul': np.multiply, 'Div': np.divide, 'Pow': np.power, 'Max': np.maximum, 'Min': np.minimum,
                'Equal': np.equal, 'Less': np.less, 'Greater': np.greater, 'And': np.logical_and, 'Or': np.logical_or, 'Mod': np.mod}[op](a, b)
    if op in ('Neg', 'Abs', 'Sqrt', 'Floor', 'Ceil', 'Not'):
        return {'Neg': np.negative, 'Abs': np.abs, 'Sqrt': np.sqrt, 'Floor': np.floor, 'Ceil': np.ceil, 'Not': np.logical_not}[op](v[0])
    if op == 'Where':
        return np.where(v[0], v[1], v[2])
    if op == 'Gather':
        return np.take(v[0], v[1].astype(np.int64), axis=int(at.get('axis', 0)))
    if op == 'Unsqueeze':
        out = v[0]
        for a in sorted(_axes_arg(node, v, 1)):
            out = np.expand_dims(out, a if a >= 0 else a + out.ndim + 1)
        return out
    if op == 'Squeeze':
        ax = _axes_arg(node, v, 1)
        return np.squeeze(v[0], axis=None if ax is None else tuple(ax))
    if op == 'Concat':
        return np.concatenate([np.atleast_1d(x) for x in v], axis=int(at['axis']))
    if op == 'Reshape':
        return v[0].reshape(_reshape_dims(v[0].shape, v[1]))
    if op == 'Slice':
        starts, ends, axes, steps = _slice_args(node, v, v[0].ndim)
        sl = [slice(None)] * v[0].ndim
        for s, e, a, st in zip(starts, ends, axes, steps):
            sl[a] = slice(s, None if (st < 0 and e < -v[0].shape[a]) else e, st)
        return v[0][tuple(sl)]
    if op == 'Range':
        return np.arange(v[0].item(), v[1].item(), v[2].item()).astype(v[0].dtype)
    if op == 'ConstantOfShape':
        val = at.get('value')
        val = np.zeros(1, np.float32) if val is None else np.asarray(val).reshape(-1)
        return np.full([int(x) for x in v[0]], val[0], dtype=val.dtype)
    if op == 'Expand':
        return v[0] * np.ones([int(x) for x in v[1]], dtype=v[0].dtype)
    if op == 'Tile':
        return np.tile(v[0], [int(x) for x in v[1]])
    if op == 'ReduceProd':
        ax = _axes_arg(node, v, 1)
        return np.prod(v[0], axis=None if ax is None else tuple(ax), keepdims=bool(at.get('keepdims', 1)))
    if op in ('ReduceSum', 'ReduceMax', 'ReduceMin'):
        ax = _axes_arg(node, v, 1)
        fn = {'ReduceSum': np.sum, 'ReduceMax': np.max, 'ReduceMin': np.min}[op]
        return fn(v[0], axis=None if ax is None else tuple(ax), keepdims=bool(at.get('keepdims', 1)))
    if op == 'Transpose':
        return np.transpose(v[0], at.get('perm'))
    raise NotImplementedError('ONNX operator %s on host (integer) values' % op)


def _reshape_dims(in_shape, target):
    dims = [int(x) for x in np.asarray(target).reshape(-1)]
    dims = [in_shape[i] if d == 0 else d for i, d in enumerate(dims)]
    if -1 in dims:
        known = int(np.prod([d for d in dims if d != -1])) if len(dims) > 1 else 1
        dims[dims.index(-1)] = int(np.prod(in_shape)) // max(known, 1)
    return dims


# ---------------------------------------------------------------------------------------------------------------------------------
# device side
# ---------------------------------------------------------------------------------------------------------------------------------

_UNARY = {'Relu': 'RELU', 'Sigmoid': 'SIGMOID', 'Tanh': 'TANH', 'Erf': 'ERF', 'Sqrt': 'SQRT', 'Exp': 'EXP', 'Log': 'LOG', 'Neg': 'NEG', 'Abs': 'ABS',
          'Round': 'ROUND', 'Floor': 'FLOOR', 'Ceil': 'CEIL', 'Reciprocal': 'RECIP', 'Softplus': 'SOFTPLUS', 'Sin': 'SIN', 'Cos': 'COS', 'Identity': 'COPY'}
_BINARY = {'Add': 'ADD', 'Sub': 'SUB', 'Mul': 'MUL', 'Div': 'DIV', 'Pow': 'POW', 'Max': 'MAX', 'Min': 'MIN', 'Equal': 'EQUAL', 'Less': 'LESS', 'Greater': 'GREATER'}
_EW = dict(COPY=0, RELU=1, SIGMOID=2, TANH=3, ERF=4, SQRT=5, EXP=6, LOG=7, NEG=8, ABS=9, ROUND=10, FLOOR=11, CEIL=12, RECIP=13, CLIP=14, LEAKY_RELU=15, SOFTPLUS=16, SIN=17,
           COS=18, ADD=32, SUB=33, MUL=34, DIV=35, POW=36, MAX=37, MIN=38, EQUAL=39, LESS=40, GREATER=41, WHERE=48)
_RED = dict(SUM=0, MEAN=1, MAX=2, MIN=3, SUMSQ=4)


# every operator the executor accepts (default ONNX domain); anything else raises NotImplementedError naming the operator
OPERATORS = ('Constant Shape Size Reshape Flatten Unsqueeze Squeeze Dropout Cast Identity Transpose Slice Expand Tile Concat Pad Gather Split '
             'Relu Sigmoid Tanh Erf Sqrt Exp Log Neg Abs Round Floor Ceil Reciprocal Softplus Sin Cos Add Sub Mul Div Pow Max Min Equal Less Greater '
             'Clip LeakyRelu Not And Or Xor Sum Mean PRelu Elu HardSigmoid Sign LogSoftmax ArgMax ArgMin Gelu Where '
             'ReduceMean ReduceSum ReduceMax ReduceMin ReduceL2 ReduceSumSquare GlobalAveragePool Softmax LayerNormalization BatchNormalization AveragePool '
             'MatMul Gemm Conv Range ConstantOfShape ReduceProd Mod').split()


# operator -> attribute names on which the device executor is held to the oracle (oracle/onnx_ref.py) by the tests: the two synthetic frontend graphs and the
# one-node cases of tests/onnx_synth.py (tests/test_host_cpu.py::test_onnx_covered_set_is_what_the_tests_reach keeps this table equal to what they reach).
# onnxruntime and the real campplus.onnx / speech_tokenizer_v3.onnx are absent from this image, so everything outside the table is UNTESTED arithmetic:
# OnnxRunner refuses such a graph instead of running it (allow_uncovered=True overrides, for bring-up against a real asset).
COVERED = {'Abs': (), 'Add': (), 'And': (), 'ArgMax': ('axis', 'keepdims'), 'ArgMin': ('axis', 'keepdims'),
           'AveragePool': ('auto_pad', 'ceil_mode', 'count_include_pad', 'kernel_shape', 'pads', 'strides'), 'BatchNormalization': ('epsilon',), 'Cast': ('to',), 'Ceil': (),
           'Clip': (), 'Concat': ('axis',), 'Constant': ('value',), 'ConstantOfShape': ('value',), 'Conv': ('auto_pad', 'dilations', 'group', 'kernel_shape', 'pads', 'strides'),
           'Cos': (), 'Div': (), 'Dropout': (), 'Elu': ('alpha',), 'Equal': (), 'Erf': (), 'Exp': (), 'Expand': (), 'Flatten': ('axis',), 'Floor': (), 'Gather': ('axis',),
           'Gelu': (), 'Gemm': ('transB',), 'GlobalAveragePool': (), 'Greater': (), 'HardSigmoid': ('alpha', 'beta'), 'Identity': (), 'LayerNormalization': ('axis', 'epsilon'),
           'LeakyRelu': ('alpha',), 'Less': (), 'Log': (), 'LogSoftmax': ('axis',), 'MatMul': (), 'Max': (), 'Mean': (), 'Min': (), 'Mod': (), 'Mul': (), 'Neg': (), 'Not': (),
           'Or': (), 'PRelu': (), 'Pad': (), 'Pow': (), 'Range': (), 'Reciprocal': (), 'ReduceL2': ('axes', 'keepdims'), 'ReduceMax': ('axes', 'keepdims'),
           'ReduceMean': ('axes', 'keepdims'), 'ReduceMin': ('axes', 'keepdims'), 'ReduceProd': ('keepdims',), 'ReduceSum': ('keepdims',),
           'ReduceSumSquare': ('axes', 'keepdims'), 'Relu': (), 'Reshape': (), 'Round': (), 'Shape': (), 'Sigmoid': (), 'Sign': (), 'Sin': (), 'Size': (), 'Slice': (),
           'Softmax': ('axis',), 'Softplus': (), 'Split': ('axis',), 'Sqrt': (), 'Squeeze': (), 'Sub': (), 'Sum': (), 'Tanh': (), 'Tile': (), 'Transpose': ('perm',),
           'Unsqueeze': (), 'Where': (), 'Xor': ()}


def uncovered(graph):
    """[(node name, operator, attribute | None)] of a graph that lie outside COVERED (None: the operator itself)"""
    out = []
    for n in graph.nodes:
        if n.op not in COVERED:
            out.append((n.name, n.op, None))
            continue
        out += [(n.name, n.op, a) for a in n.attrs if a not in COVERED[n.op]]
    return out


class OnnxRunner:
    """run(feeds) -> {output name: numpy array}.  Floating-point tensors live on the device (fp32, contiguous), integer tensors on the host.
    A graph that uses an operator or an attribute outside COVERED is refused at construction (NotImplementedError naming every such node)."""

    def __init__(self, graph, device='cuda', allow_uncovered=False):
        import torch
        from . import _lib
        _lib.require_gpu()
        self.torch, self._lib, self.lib = torch, _lib, _lib.load()
        self.g = graph if isinstance(graph, Graph) else load_onnx(graph)
        bad = [] if allow_uncovered else uncovered(self.g)
        if bad:
            raise NotImplementedError('ONNX graph steps outside the executor\'s tested operator / attribute set (onnx_graph.COVERED): '
                                      + ', '.join('%s%s' % (op, '' if a is None else '.' + a) for _, op, a in bad[:12]) + (' ...' if len(bad) > 12 else '')
                                      + '; pass allow_uncovered=True to run it anyway (parity of those nodes is unpinned)')
        self.device = torch.device(device)
        self.consts = {}
        for name, arr in self.g.initializers.items():
            self.consts[name] = self._up(arr) if (np.issubdtype(arr.dtype, np.floating) and arr.ndim > 0) else arr      # (scalars and integers stay host values)
        self._wcache = {}
        self.op_counts = {}

    # ---- plumbing -----------------------------------------------------------------------------------------------------------------
    def _up(self, arr):
        return self.torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(self.device)

    def _is_dev(self, x):
        return self.torch.is_tensor(x)

    def _host(self, x):
        return x.detach().cpu().numpy() if self._is_dev(x) else np.asarray(x)

    def _dev(self, x):
        return x if self._is_dev(x) else self._up(np.asarray(x))

    def _new(self, shape):
        return self.torch.empty([int(s) for s in shape], dtype=self.torch.float32, device=self.device)

    def _ew(self, op, shape, a, sa, b=None, sb=None, c=None, sc=None, p0=0.0, p1=0.0, a_off=0):
        """out[shape] = op(a, b, c) with element strides sa / sb / sc over the output axes (0 broadcasts); a_off: element offset into a"""
        shape = [int(s) for s in shape]
        out = self._new(shape)
        if out.numel() == 0:
            return out
        # fold to at most 6 axes: merge trailing axes that are jointly contiguous in every operand
        dims = [(shape[i], sa[i], (sb[i] if sb else 0), (sc[i] if sc else 0)) for i in range(len(shape)) if shape[i] != 1] or [(1, 0, 0, 0)]
        merged = [dims[-1]]
        for n, x, y, z in reversed(dims[:-1]):
            m = merged[0]
            if x == m[1] * m[0] and y == m[2] * m[0] and z == m[3] * m[0]:
                merged[0] = (n * m[0], m[1], m[2], m[3])
            else:
                merged.insert(0, (n, x, y, z))
        if len(merged) > 6:
            raise NotImplementedError('more than 6 non-mergeable axes in one elementwise operator')
        d = self._lib.NdDesc()
        d.ndim = len(merged)
        for i, (n, x, y, z) in enumerate(merged):
            d.shape[i], d.stride_a[i], d.stride_b[i], d.stride_c[i] = n, x, y, z
        ptr = self._lib.ptr
        pa = C.c_void_p(a.data_ptr() + 4 * a_off)
        self._lib.check(self.lib.hvx_nd_elementwise(_EW[op], C.byref(d), pa, ptr(b), ptr(c), float(p0), float(p1), ptr(out), self._lib.stream_ptr()), 'hvx_nd_elementwise')
        return out

    @staticmethod
    def _strides(shape):
        st, acc = [], 1
        for n in reversed(shape):
            st.insert(0, acc)
            acc *= int(n)
        return st

    def _bstrides(self, shape, out_shape):
        """element strides of a contiguous tensor of `shape` broadcast to out_shape"""
        shape = [1] * (len(out_shape) - len(shape)) + [int(s) for s in shape]
        st = self._strides(shape)
        return [0 if shape[i] == 1 and out_shape[i] != 1 else st[i] for i in range(len(out_shape))]

    def _binary(self, op, a, b):
        a, b = self._dev(a), self._dev(b)
        out_shape = list(np.broadcast_shapes(tuple(a.shape), tuple(b.shape)))
        return self._ew(op, out_shape, a, self._bstrides(a.shape, out_shape), b, self._bstrides(b.shape, out_shape))

    def _unary(self, op, a, p0=0.0, p1=0.0):
        return self._ew(op, list(a.shape), a, self._strides(a.shape), p0=p0, p1=p1)

    def _permute(self, x, perm):
        st = self._strides(x.shape)
        return self._ew('COPY', [x.shape[p] for p in perm], x, [st[p] for p in perm])

    def _reduce(self, op, x, axes, keepdims):
        rank = x.dim()
        axes = sorted(a % rank for a in (axes if axes is not None else range(rank)))
        keep = [i for i in range(rank) if i not in axes]
        xp = x if axes == list(range(rank - len(axes), rank)) else self._permute(x, keep + axes)
        rows = int(np.prod([x.shape[i] for i in keep])) if keep else 1
        cols = int(np.prod([x.shape[i] for i in axes]))
        out = self._new([rows])
        self._lib.check(self.lib.hvx_rows_reduce(_RED[op], self._lib.ptr(xp), rows, cols, self._lib.ptr(out), self._lib.stream_ptr()), 'hvx_rows_reduce')
        shape = [1 if i in axes else x.shape[i] for i in range(rank)] if keepdims else [x.shape[i] for i in keep]
        return out.reshape(shape)

    # ---- GEMM-shaped operators ---------------------------------------------------------------------------------------------------------
    def _gemm_rows(self, a2, w_nk, bias=None):
        """a2 [M][K] (device, contiguous) x w_nk [N][K] (device) -> [M][N] through the exact-fp32 GEMM of the hot path (K padded to 32 with zeros)"""
        from . import ops
        torch = self.torch
        M, K = a2.shape
        N = w_nk.shape[0]
        Kp = (K + 31) // 32 * 32
        if Kp != K:
            ap = torch.zeros(M, Kp, dtype=torch.float32, device=self.device)
            ap[:, :K] = a2
            wp = torch.zeros(N, Kp, dtype=torch.float32, device=self.device)
            wp[:, :K] = w_nk
            a2, w_nk = ap, wp
        out = ops.conv1d(a2.reshape(1, M, Kp), w_nk.contiguous(), bias, n_out=N, taps=1, cin_pad=Kp)
        return out.reshape(M, N)

    def _matmul(self, a, b, b_name=None):
        a, b = self._dev(a), self._dev(b)
        if b.dim() == 2:
            key = ('mm', b_name)
            w = self._wcache.get(key) if b_name in self.consts else None
            if w is None:
                w = self._permute(b, [1, 0])
                if b_name in self.consts:
                    self._wcache[key] = w
            lead = list(a.shape[:-1])
            return self._gemm_rows(a.reshape(-1, a.shape[-1]), w).reshape(lead + [b.shape[1]])
        # batched: broadcast the leading axes, one GEMM per matrix pair
        lead = list(np.broadcast_shapes(tuple(a.shape[:-2]), tuple(b.shape[:-2])))
        M, K, N = a.shape[-2], a.shape[-1], b.shape[-1]
        ae = self._ew('COPY', lead + [M, K], a, self._bstrides(a.shape, lead + [M, K])).reshape(-1, M, K)
        bt = self._ew('COPY', lead + [N, K], b, (lambda s: s[:-2] + [s[-1], s[-2]])(self._bstrides(b.shape, lead + [K, N]))).reshape(-1, N, K)
        out = self._new([ae.shape[0], M, N])
        for i in range(ae.shape[0]):
            out[i] = self._gemm_rows(ae[i], bt[i])
        return out.reshape(lead + [M, N])

    def _conv(self, node, x, w, bias, w_name):
        at = node.attrs
        x = self._dev(x)
        group = int(at.get('group', 1))
        if x.dim() == 4:
            dil, strides = at.get('dilations', [1, 1]), at.get('strides', [1, 1])
            pads = _resolve_pads(at, x.shape[2:], w.shape[2:], strides, dil)
            if group != 1 or list(dil) != [1, 1] or pads[0] != pads[2] or pads[1] != pads[3]:
                raise NotImplementedError('2-D Conv with groups / dilation / asymmetric padding')
            B, Cin, H, W = x.shape
            Cout, _, kh, kw = w.shape
            Ho, Wo = (H + 2 * pads[0] - kh) // strides[0] + 1, (W + 2 * pads[1] - kw) // strides[1] + 1
            y = self._new([B, Cout, Ho, Wo])
            ptr = self._lib.ptr
            wd, bd = self._dev(w), (None if bias is None else self._dev(bias))       # (named: a temporary would be freed — and its block re-used — before the launch)
            self._lib.check(self.lib.hvx_conv2d(ptr(x), ptr(wd), ptr(bd), B, Cin, H, W, Cout, kh, kw, strides[0],
                                                strides[1], pads[0], pads[1], ptr(y), self._lib.stream_ptr()), 'hvx_conv2d')
            return y
        from . import ops, packing
        torch = self.torch
        B, Cin, T = x.shape
        Cout, cg, k = w.shape
        dil, stride = int(at.get('dilations', [1])[0]), int(at.get('strides', [1])[0])
        pads = _resolve_pads(at, [T], [k], [stride], [dil])
        T_out = (T + pads[0] + pads[1] - dil * (k - 1) - 1) // stride + 1
        key = ('conv', w_name)
        packed = self._wcache.get(key) if w_name in self.consts else None
        if packed is None:
            wh = (w.detach().cpu().numpy() if self._is_dev(w) else np.asarray(w, np.float32))
            wt = torch.from_numpy(np.ascontiguousarray(wh))
            packed = packing.conv_weight(wt).to(self.device)      # [Cout][k][cg padded to 32]: rows of group g are its Cout / group output channels
            if w_name in self.consts:
                self._wcache[key] = packed
        cin_pad = packed.shape[-1] // k
        # channel-major [B][C][T] -> time-major rows [B][T][groups * cin_pad] (zero-padded channels)
        if group > 1:
            rows = torch.zeros(B, T, group * cin_pad, dtype=torch.float32, device=self.device)
            xt = self._permute(x, [0, 2, 1])
            rows.view(B, T, group, cin_pad)[..., :cg] = xt.view(B, T, group, cg)
        else:
            rows = torch.zeros(B, T, cin_pad, dtype=torch.float32, device=self.device)
            rows[..., :Cin] = self._permute(x, [0, 2, 1])
        y = ops.conv1d(rows, packed, None if bias is None else self._dev(bias), n_out=Cout // group, taps=k, cin_pad=cin_pad, pad_left=int(pads[0]), dil=dil, stride=stride,
                       m_out=T_out, groups=group)
        return self._permute(y[..., :Cout].contiguous(), [0, 2, 1])

    # ---- the walk -------------------------------------------------------------------------------------------------------------------
    def run(self, feeds):
        torch = self.torch
        vals = dict(self.consts)
        for name in self.g.inputs:
            if torch.is_tensor(feeds[name]) and feeds[name].is_floating_point():           # (a device tensor is taken as it is)
                vals[name] = feeds[name].to(self.device, torch.float32).contiguous()
                continue
            x = np.asarray(feeds[name].cpu() if torch.is_tensor(feeds[name]) else feeds[name])
            vals[name] = self._up(x) if np.issubdtype(x.dtype, np.floating) else x
        for node in self.g.nodes:
            ins = [vals[i] if i else None for i in node.inputs]
            self.op_counts[node.op] = self.op_counts.get(node.op, 0) + 1
            outs = self._node(node, ins)
            if not isinstance(outs, (list, tuple)):
                outs = [outs]
            for name, v in zip(node.outputs, outs):
                if name:
                    vals[name] = v
        torch.cuda.synchronize(self.device)
        return {o: (vals[o].detach().cpu().numpy() if self._is_dev(vals[o]) else np.asarray(vals[o])) for o in self.g.outputs}

    def _node(self, node, v):
        op, at = node.op, node.attrs
        torch = self.torch
        if op == 'Constant':
            val = at['value'] if 'value' in at else (np.asarray(at['value_float'], np.float32) if 'value_float' in at else
                                                     np.asarray(at.get('value_int', at.get('value_ints', at.get('value_floats')))))
            val = np.asarray(val)
            return self._up(val) if np.issubdtype(val.dtype, np.floating) and val.ndim > 0 else val
        if op == 'Shape':
            return np.asarray(list(v[0].shape), np.int64)
        if op == 'Size':
            return np.asarray(int(np.prod(v[0].shape)), np.int64)
        dev = [x for x in v if x is not None and self._is_dev(x)]
        if not dev:
            return _host_eval(node, [None if x is None else np.asarray(x) for x in v])
        x = v[0]
        # ---- pure views (device tensors are always contiguous) -------------------------------------------------------------------------------
        if op == 'Reshape':
            return x.reshape(_reshape_dims(list(x.shape), v[1]))
        if op == 'Flatten':
            ax = int(at.get('axis', 1))
            return x.reshape(int(np.prod(x.shape[:ax])) if ax else 1, -1)
        if op == 'Unsqueeze':
            out = x
            for a in sorted(_axes_arg(node, v, 1)):
                out = out.unsqueeze(a)
            return out
        if op == 'Squeeze':
            ax = _axes_arg(node, v, 1)
            return x.reshape([n for i, n in enumerate(x.shape) if not ((ax is None and n == 1) or (ax is not None and (i in ax or i - x.dim() in ax)))])
        if op in ('Dropout',):
            return x
        if op == 'Cast':
            to = _DTYPES[int(at['to'])]
            if to == np.float16:                                 # device values stay fp32 STORAGE, but a Cast to fp16 rounds (and saturates to inf) like the graph says
                return x.to(self.torch.float16).to(self.torch.float32)
            return x if np.issubdtype(to, np.floating) else x.detach().cpu().numpy().astype(to)   # (an integer result is a host value from here on)
        # ---- data movement ----------------------------------------------------------------------------------------------------------------
        if op == 'Transpose':
            return self._permute(x, at.get('perm') or list(range(x.dim()))[::-1])
        if op == 'Slice':
            starts, ends, axes, steps = _slice_args(node, v, x.dim())
            shape, st, off = list(x.shape), self._strides(x.shape), 0
            base = list(st)
            for s, e, a, sp in zip(starts, ends, axes, steps):
                n = x.shape[a]
                idx = range(n)[slice(s, None if (sp < 0 and e < -n) else e, sp)]
                shape[a] = len(idx)
                off += (idx.start if len(idx) else 0) * base[a]
                st[a] = base[a] * sp
            return self._ew('COPY', shape, x, st, a_off=off)
        if op == 'Expand':
            out_shape = list(np.broadcast_shapes(tuple(x.shape), tuple(int(s) for s in v[1])))
            return self._ew('COPY', out_shape, x, self._bstrides(x.shape, out_shape))
        if op == 'Tile':
            reps = [int(r) for r in v[1]]
            out = x
            for a, r in enumerate(reps):
                if r != 1:
                    shp = list(out.shape)
                    e = self._ew('COPY', shp[:a] + [r] + shp[a:], out, (lambda s: s[:a] + [0] + s[a:])(self._strides(out.shape)))
                    out = e.reshape(shp[:a] + [r * shp[a]] + shp[a + 1:])
            return out
        if op == 'Concat':
            ax = int(at['axis']) % max(d.dim() for d in dev)
            parts = [self._dev(p) for p in v]
            out = self._new([sum(p.shape[i] for p in parts) if i == ax else parts[0].shape[i] for i in range(parts[0].dim())])
            o = 0
            for p in parts:                                        # destination-strided copies: device memory plumbing
                out.narrow(ax, o, p.shape[ax]).copy_(p)
                o += p.shape[ax]
            return out
        if op == 'Pad':
            pads = [int(p) for p in (v[1] if len(v) > 1 and v[1] is not None else at['pads'])]
            val = float(self._host(v[2]).reshape(-1)[0]) if len(v) > 2 and v[2] is not None else float(at.get('value', 0.0))
            if at.get('mode', b'constant') not in (b'constant', 'constant'):
                raise NotImplementedError('Pad mode %r' % at.get('mode'))
            r = x.dim()
            out = torch.full([x.shape[i] + pads[i] + pads[r + i] for i in range(r)], val, dtype=torch.float32, device=self.device)
            sl = out
            for i in range(r):
                sl = sl.narrow(i, pads[i], x.shape[i])
            sl.copy_(x)
            return out
        if op == 'Gather':
            idx = np.asarray(v[1]).astype(np.int64)
            ax = int(at.get('axis', 0)) % x.dim()
            idx = np.where(idx < 0, idx + x.shape[ax], idx)
            if idx.ndim == 0:                                      # a scalar index is a slice
                st = self._strides(x.shape)
                shape = list(x.shape[:ax]) + list(x.shape[ax + 1:])
                return self._ew('COPY', shape or [1], x, (st[:ax] + st[ax + 1:]) or [0], a_off=int(idx) * st[ax]).reshape(shape)
            pieces = [self._ew('COPY', list(x.shape[:ax]) + [1] + list(x.shape[ax + 1:]), x, self._strides(x.shape), a_off=int(i) * self._strides(x.shape)[ax])
                      for i in idx.reshape(-1)]
            out = torch.cat(pieces, dim=ax)
            return out.reshape(list(x.shape[:ax]) + list(idx.shape) + list(x.shape[ax + 1:]))
        # ---- arithmetic ---------------------------------------------------------------------------------------------------------------------
        if op in _UNARY:
            return self._unary(_UNARY[op], x)
        if op in _BINARY and len(v) == 2:
            return self._binary(_BINARY[op], v[0], v[1])
        if op == 'Clip':
            lo = v[1] if len(v) > 1 and v[1] is not None else at.get('min', -3.4e38)
            hi = v[2] if len(v) > 2 and v[2] is not None else at.get('max', 3.4e38)
            return self._unary('CLIP', x, float(self._host(lo).reshape(-1)[0]), float(self._host(hi).reshape(-1)[0]))
        if op == 'LeakyRelu':
            return self._unary('LEAKY_RELU', x, float(at.get('alpha', 0.01)))
        if op == 'Not':
            return self._binary('EQUAL', x, np.zeros(1, np.float32))
        one, zero = np.ones(1, np.float32), np.zeros(1, np.float32)
        if op in ('And', 'Or', 'Xor'):                              # booleans are 0 / 1 floats on the device
            a, b = self._dev(np.asarray(v[0], np.float32) if not self._is_dev(v[0]) else v[0]), self._dev(np.asarray(v[1], np.float32) if not self._is_dev(v[1]) else v[1])
            return self._binary('MUL', a, b) if op == 'And' else (self._binary('MAX', a, b) if op == 'Or' else self._binary('EQUAL', self._binary('EQUAL', a, b), zero))
        if op in ('Sum', 'Mean') or (op in ('Max', 'Min') and len(v) != 2):
            acc = self._dev(v[0])
            for t in v[1:]:
                acc = self._binary({'Sum': 'ADD', 'Mean': 'ADD', 'Max': 'MAX', 'Min': 'MIN'}[op], acc, t)
            return self._binary('DIV', acc, np.asarray([float(len(v))], np.float32)) if op == 'Mean' else acc
        if op == 'PRelu':
            return self._binary('ADD', self._unary('RELU', x), self._binary('MUL', self._binary('MIN', x, zero), v[1]))
        if op == 'Elu':
            neg = self._binary('MUL', self._binary('SUB', self._unary('EXP', self._binary('MIN', x, zero)), one), np.asarray([float(at.get('alpha', 1.0))], np.float32))
            return self._binary('ADD', self._unary('RELU', x), neg)
        if op == 'HardSigmoid':
            y = self._binary('ADD', self._binary('MUL', x, np.asarray([float(at.get('alpha', 0.2))], np.float32)), np.asarray([float(at.get('beta', 0.5))], np.float32))
            return self._unary('CLIP', y, 0.0, 1.0)
        if op == 'Sign':
            return self._binary('SUB', self._binary('GREATER', x, zero), self._binary('LESS', x, zero))
        if op == 'LogSoftmax':
            sm = self._node(Node('Softmax', node.inputs, node.outputs, {'axis': at.get('axis', 1 if self.g.opset < 13 else -1)}), v)
            return self._unary('LOG', sm)
        if op in ('ArgMax', 'ArgMin'):
            # first index of the extremum: min over where(x == extremum, index, n) — three elementwise passes and two reductions, no dedicated kernel
            ax = int(at.get('axis', 0)) % x.dim()
            if int(at.get('select_last_index', 0)):
                raise NotImplementedError('ArgMax / ArgMin with select_last_index')
            ext = self._reduce('MAX' if op == 'ArgMax' else 'MIN', x, [ax], True)
            n = x.shape[ax]
            iota = np.arange(n, dtype=np.float32).reshape([n if i == ax else 1 for i in range(x.dim())])
            hit = self._binary('EQUAL', x, ext)
            c, a, b = hit, self._dev(iota), self._dev(np.asarray([float(n)], np.float32))
            cand = self._ew('WHERE', list(x.shape), c, self._strides(c.shape), a, self._bstrides(a.shape, list(x.shape)), b, self._bstrides(b.shape, list(x.shape)))
            idx = self._reduce('MIN', cand, [ax], bool(at.get('keepdims', 1)))
            return idx.detach().cpu().numpy().astype(np.int64)       # (an integer result is a host value)
        if op == 'Split':
            ax = int(at.get('axis', 0)) % x.dim()
            sizes = [int(t) for t in (v[1] if len(v) > 1 and v[1] is not None else at.get('split', []))]
            if not sizes:
                k = len(node.outputs)
                sizes = [-(-x.shape[ax] // k)] * k
                sizes[-1] = x.shape[ax] - sum(sizes[:-1])
            outs, o, st = [], 0, self._strides(x.shape)
            for n in sizes:
                shape = list(x.shape)
                shape[ax] = n
                outs.append(self._ew('COPY', shape, x, st, a_off=o * st[ax]))
                o += n
            return outs
        if op == 'Gelu':
            h = self._binary('MUL', x, np.asarray([0.7071067811865476], np.float32))
            return self._binary('MUL', self._binary('MUL', x, np.asarray([0.5], np.float32)), self._binary('ADD', self._unary('ERF', h), np.asarray([1.0], np.float32)))
        if op == 'Where':
            c, a, b = self._dev(np.asarray(v[0], np.float32) if not self._is_dev(v[0]) else v[0]), self._dev(v[1]), self._dev(v[2])
            out_shape = list(np.broadcast_shapes(tuple(c.shape), tuple(a.shape), tuple(b.shape)))
            return self._ew('WHERE', out_shape, c, self._bstrides(c.shape, out_shape), a, self._bstrides(a.shape, out_shape), b, self._bstrides(b.shape, out_shape))
        if op in ('ReduceMean', 'ReduceSum', 'ReduceMax', 'ReduceMin', 'ReduceL2', 'ReduceSumSquare'):
            ax = _axes_arg(node, v, 1)
            kd = bool(at.get('keepdims', 1))
            if op == 'ReduceL2':
                return self._unary('SQRT', self._reduce('SUMSQ', x, ax, kd))
            return self._reduce({'ReduceMean': 'MEAN', 'ReduceSum': 'SUM', 'ReduceMax': 'MAX', 'ReduceMin': 'MIN', 'ReduceSumSquare': 'SUMSQ'}[op], x, ax, kd)
        if op == 'GlobalAveragePool':
            return self._reduce('MEAN', x, list(range(2, x.dim())), True)
        if op == 'Softmax':
            ax = int(at.get('axis', 1 if self.g.opset < 13 else -1)) % x.dim()      # (the default axis is 1 before opset 13, the last one from 13 on)
            if self.g.opset < 13 and ax != x.dim() - 1:            # (older opsets flatten from `axis`)
                rows, cols = int(np.prod(x.shape[:ax])), int(np.prod(x.shape[ax:]))
                xp, back = x, None
            else:
                perm = [i for i in range(x.dim()) if i != ax] + [ax]
                xp = x if ax == x.dim() - 1 else self._permute(x, perm)
                back = None if ax == x.dim() - 1 else [perm.index(i) for i in range(x.dim())]
                rows, cols = xp.numel() // xp.shape[-1], xp.shape[-1]
            out = self._new(list(xp.shape))
            self._lib.check(self.lib.hvx_rows_softmax(self._lib.ptr(xp), rows, cols, self._lib.ptr(out), self._lib.stream_ptr()), 'hvx_rows_softmax')
            return out if back is None else self._permute(out, back)
        if op == 'LayerNormalization':
            ax = int(at.get('axis', -1)) % x.dim()
            axes = list(range(ax, x.dim()))
            mean = self._reduce('MEAN', x, axes, True)
            d = self._binary('SUB', x, mean)
            var = self._reduce('MEAN', self._binary('MUL', d, d), axes, True)
            y = self._binary('DIV', d, self._unary('SQRT', self._binary('ADD', var, np.asarray([float(at.get('epsilon', 1e-5))], np.float32))))
            y = self._binary('MUL', y, v[1])
            return self._binary('ADD', y, v[2]) if len(v) > 2 and v[2] is not None else y
        if op == 'BatchNormalization':
            sc, bi, mu, var = [(p.detach().cpu().numpy() if self._is_dev(p) else np.asarray(p)).astype(np.float64) for p in v[1:5]]
            a = sc / np.sqrt(var + float(at.get('epsilon', 1e-5)))
            shp = [1, -1] + [1] * (x.dim() - 2)
            return self._binary('ADD', self._binary('MUL', x, a.astype(np.float32).reshape(shp)), (bi - mu * a).astype(np.float32).reshape(shp))
        if op == 'AveragePool':
            k, st = at['kernel_shape'], at.get('strides', [1] * len(at['kernel_shape']))
            pads = _resolve_pads(at, x.shape[2:], k, st, [1] * len(k))
            if len(k) != 1 or pads[0] != pads[1]:
                raise NotImplementedError('AveragePool beyond 1-D with symmetric padding')
            T = x.shape[-1]
            num = T + 2 * pads[0] - k[0]
            t_out = (-(-num // st[0]) if at.get('ceil_mode', 0) else num // st[0]) + 1
            if at.get('ceil_mode', 0) and (t_out - 1) * st[0] >= T + pads[0]:
                t_out -= 1                                       # (the last window must start inside the input or its left padding)
            out = self._new(list(x.shape[:-1]) + [t_out])
            self._lib.check(self.lib.hvx_avgpool_rows(self._lib.ptr(x), x.numel() // T, T, k[0], st[0], pads[0], int(at.get('count_include_pad', 0)), self._lib.ptr(out), t_out,
                                                      self._lib.stream_ptr()), 'hvx_avgpool_rows')
            return out
        # ---- GEMM-shaped ------------------------------------------------------------------------------------------------------------------------
        if op == 'MatMul':
            return self._matmul(v[0], v[1], node.inputs[1])
        if op == 'Gemm':
            a, b = self._dev(v[0]), self._dev(v[1])
            if int(at.get('transA', 0)):
                a = self._permute(a, [1, 0])
            w = b if int(at.get('transB', 0)) else self._permute(b, [1, 0])
            y = self._gemm_rows(a, w)
            alpha, beta = float(at.get('alpha', 1.0)), float(at.get('beta', 1.0))
            if alpha != 1.0:
                y = self._binary('MUL', y, np.asarray([alpha], np.float32))
            if len(v) > 2 and v[2] is not None:
                c = self._dev(v[2])
                y = self._binary('ADD', y, c if beta == 1.0 else self._binary('MUL', c, np.asarray([beta], np.float32)))
            return y
        if op == 'Conv':
            return self._conv(node, v[0], v[1], v[2] if len(v) > 2 else None, node.inputs[1])
        raise NotImplementedError('ONNX operator %s' % op)
