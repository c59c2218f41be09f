"""The TSN BN-Inception graph as data.

Two sources give the same ``Graph``:

* :func:`parse_prototxt` -- a small reader for Caffe's text format, so the drop-in command line can
  take the very files the reference passes to ``CaffeNet``
  (src/features_GPU_compute/models/ucf101/tsn_bn_inception_{rgb,flow}_deploy.prototxt;
  calcSig_wOF.py:158-161);
* :func:`bn_inception` -- the same topology generated from a compact table, for
  when no prototxt is at hand (bench, smoke, GPU box).

``Graph.plan()`` lowers the layer list to what the device executes: Convolution + frozen BN + ReLU
become one op, Concat disappears (producers write at a channel offset of the concat tensor),
Dropout is the identity in TEST phase, and ``fc-action`` -- computed by the reference but never read
by the feature path (calcSig_wOF.py:95 reads ``global_pool``) -- is dropped.
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple


# ------------------------------------------------------------------------------------------------
# graph model
# ------------------------------------------------------------------------------------------------
@dataclass
class Layer:
    name: str
    type: str                      # Convolution | BN | ReLU | Pooling | Concat | Dropout | InnerProduct
    bottoms: List[str]
    tops: List[str]
    num_output: int = 0
    kernel: int = 1
    stride: int = 1
    pad: int = 0
    pool: str = ""                 # MAX | AVE


@dataclass
class Graph:
    name: str
    input_name: str
    input_shape: Tuple[int, int, int]          # (C, H, W)
    layers: List[Layer] = field(default_factory=list)

    def conv_layers(self) -> List[Layer]:
        return [l for l in self.layers if l.type == "Convolution"]

    def plan(self, feature_blob: str = "global_pool", fuse: bool = True) -> "Plan":
        p = _lower(self, feature_blob)
        return _fuse(p) if fuse else p


@dataclass
class Tensor:
    h: int
    w: int
    c: int
    name: str


@dataclass
class Op:
    kind: str                      # conv | maxpool | avgpool | gavgpool
    name: str                      # name of the producing layer (weights are keyed by it)
    src: int
    dst: int
    src_coff: int
    dst_coff: int
    cin: int
    cout: int
    k: int = 1
    stride: int = 1
    pad: int = 0
    relu: bool = False
    bn: Optional[str] = None       # name of the folded BN layer
    out_blob: str = ""
    bias: bool = True              # conv: add the (folded) bias in the epilogue
    segments: Optional[List["Segment"]] = None   # merged sibling 1x1 convs: one GEMM, several destinations
    bias_from: Optional[Tuple[str, Optional[str]]] = None   # avgpool that finishes a commuted pool_proj: (conv, bn)
    pre_pool: Optional[Tuple[int, int, str]] = None   # conv 1x1 that reads max-pool(src): (window, stride, name of the Pooling layer)


@dataclass
class Segment:
    name: str                      # conv layer whose weights fill these output columns
    bn: Optional[str]
    cout: int
    dst: int
    dst_coff: int
    relu: bool
    bias: bool


@dataclass
class Plan:
    tensors: List[Tensor]
    ops: List[Op]
    feature_slot: int
    feature_dim: int
    blob_loc: Dict[str, Tuple[int, int, int]]     # blob -> (slot, channel offset, channels)

    def macs_per_crop(self) -> int:
        total = 0
        for op in self.ops:
            if op.kind == "conv":
                t = self.tensors[op.segments[0].dst if op.segments else op.dst]
                total += t.h * t.w * op.cout * op.cin * op.k * op.k
        return total


def conv_out(size: int, k: int, s: int, p: int) -> int:
    """Caffe convolution output size: floor((in + 2p - k)/s) + 1."""
    return (size + 2 * p - k) // s + 1


def pool_out(size: int, k: int, s: int, p: int) -> int:
    """Caffe pooling output size: ceil((in + 2p - k)/s) + 1, minus one if the last window would
    start in the padding (pooling_layer.cpp)."""
    out = -(-(size + 2 * p - k) // s) + 1
    if p > 0 and (out - 1) * s >= size + p:
        out -= 1
    return out


def _lower(g: Graph, feature_blob: str) -> Plan:
    c0, h0, w0 = g.input_shape
    shapes: Dict[str, Tuple[int, int, int]] = {g.input_name: (c0, h0, w0)}     # blob -> (C,H,W)
    # pass 1: shapes + which blobs live inside a concat output
    inside: Dict[str, Tuple[str, int]] = {}                                     # blob -> (concat top, channel offset)
    for l in g.layers:
        if l.type == "Convolution":
            c, h, w = shapes[l.bottoms[0]]
            shapes[l.tops[0]] = (l.num_output, conv_out(h, l.kernel, l.stride, l.pad), conv_out(w, l.kernel, l.stride, l.pad))
        elif l.type in ("BN", "ReLU", "Dropout"):
            shapes[l.tops[0]] = shapes[l.bottoms[0]]
        elif l.type == "Pooling":
            c, h, w = shapes[l.bottoms[0]]
            shapes[l.tops[0]] = (c, pool_out(h, l.kernel, l.stride, l.pad), pool_out(w, l.kernel, l.stride, l.pad))
        elif l.type == "Concat":
            off = 0
            hw = None
            for b in l.bottoms:
                c, h, w = shapes[b]
                if hw is None:
                    hw = (h, w)
                if (h, w) != hw:
                    raise ValueError("Concat %s: spatial mismatch" % l.name)
                if b in inside:
                    raise ValueError("blob %s feeds two Concat layers" % b)
                inside[b] = (l.tops[0], off)
                off += c
            shapes[l.tops[0]] = (off, hw[0], hw[1])
        elif l.type == "InnerProduct":
            shapes[l.tops[0]] = (l.num_output, 1, 1)
        else:
            raise ValueError("unsupported layer type %s (%s)" % (l.type, l.name))
    if feature_blob not in shapes:
        raise KeyError("feature blob %r is not produced by this network" % feature_blob)

    tensors: List[Tensor] = []
    loc: Dict[str, Tuple[int, int, int]] = {}

    def slot_for(blob: str) -> Tuple[int, int, int]:
        if blob in loc:
            return loc[blob]
        c, h, w = shapes[blob]
        if blob in inside:
            parent, off = inside[blob]
            ps, pc, _ = slot_for(parent)
            loc[blob] = (ps, pc + off, c)
        else:
            tensors.append(Tensor(h, w, c, blob))
            loc[blob] = (len(tensors) - 1, 0, c)
        return loc[blob]

    slot_for(g.input_name)          # slot 0 = network input
    ops: List[Op] = []
    by_top: Dict[str, Op] = {}      # blob -> op that produced it (for BN / ReLU folding)
    for l in g.layers:
        if l.type == "Convolution":
            src = slot_for(l.bottoms[0])
            # the conv's own top is never materialised if a BN consumes it: decide at BN time
            op = Op("conv", l.name, src[0], -1, src[1], 0, src[2], l.num_output, l.kernel, l.stride, l.pad,
                    out_blob=l.tops[0])
            ops.append(op)
            by_top[l.tops[0]] = op
        elif l.type == "BN":
            op = by_top.get(l.bottoms[0])
            if op is None or op.kind != "conv" or op.bn is not None:
                raise ValueError("BN %s does not follow a Convolution" % l.name)
            op.bn = l.name
            op.out_blob = l.tops[0]
            by_top[l.tops[0]] = op
        elif l.type == "ReLU":
            op = by_top.get(l.bottoms[0])
            if op is None or op.kind != "conv" or l.tops[0] != l.bottoms[0]:
                raise ValueError("ReLU %s is not an in-place activation of a Convolution" % l.name)
            op.relu = True
        elif l.type == "Pooling":
            src = slot_for(l.bottoms[0])
            c, h, w = shapes[l.bottoms[0]]
            is_global = (l.pool == "AVE" and l.kernel == h and l.kernel == w and l.pad == 0)
            kind = "gavgpool" if is_global else ("maxpool" if l.pool == "MAX" else "avgpool")
            op = Op(kind, l.name, src[0], -1, src[1], 0, c, c, l.kernel, l.stride, l.pad, out_blob=l.tops[0])
            ops.append(op)
            by_top[l.tops[0]] = op
        elif l.type == "Dropout":
            if l.tops[0] != l.bottoms[0]:
                loc[l.tops[0]] = slot_for(l.bottoms[0])
        elif l.type in ("Concat", "InnerProduct"):
            pass
    # resolve destinations now that BN renaming is known; drop ops whose output nobody needs
    needed = {feature_blob}
    keep: List[Op] = []
    consumers: Dict[str, int] = {}
    for l in g.layers:
        if l.type in ("Convolution", "Pooling", "Concat"):
            for b in l.bottoms:
                consumers[b] = consumers.get(b, 0) + 1
    for op in ops:
        blob = op.out_blob
        if blob != feature_blob and consumers.get(blob, 0) == 0:
            continue                                             # e.g. nothing: fc-action is not an op
        d = slot_for(blob)
        op.dst, op.dst_coff = d[0], d[1]
        keep.append(op)
    fs = slot_for(feature_blob)
    if fs[1] != 0 or tensors[fs[0]].h != 1 or tensors[fs[0]].w != 1:
        raise ValueError("feature blob must be a 1x1 tensor of its own")
    return Plan(tensors, keep, fs[0], fs[2], dict(loc))


# one column tile of the widest tiling of the pooled-input kernels (64 x 256): every pooling window is then read exactly once
FOLD_POOL_MAX_COUT = 256


def _fuse(plan: Plan) -> Plan:
    """Two graph-level rewrites that change no layer's mathematics:

    * ``pool_proj(avgpool(x))`` -> ``avgpool(pool_proj_linear(x)) + bias, ReLU``: a 3x3/1 average pool and a 1x1
      convolution are both linear and commute (zero padding counted in the divisor either way), so the pool runs
      on the 32..128 projected channels instead of the 192..1056 input channels, and the projection becomes one
      more sibling of the block's other 1x1 convolutions;
    * sibling 1x1/1 convolutions that read the same tensor (3-4 per inception block) become ONE implicit GEMM
      with concatenated output columns; each 32-column group is stored to its own destination;
    * a 3x3 max pool read by exactly one such GEMM of at most FOLD_POOL_MAX_COUT output columns (pool1 ->
      conv2/3x3_reduce; pool2 -> the 224-column sibling group of inception_3a) disappears into it: the GEMM takes the maximum
      over the window while it stages a pixel, and the pooled tensor is never written or read back.  The limit is ONE column
      tile of the widest pooled-input tiling (64 x 256): a wider GEMM would walk its rows once per column tile and take every
      window maximum again each time (pool2 folded into two 128-column tiles: 0.129 ms against 0.060 + 0.070 ms apart; into one
      256-column tile: 0.108-0.113 ms, round 5).
    """
    ops = list(plan.ops)
    tensors = list(plan.tensors)
    loc = dict(plan.blob_loc)
    # --- commute average pool and projection
    out: List[Op] = []
    i = 0
    consumed_by: Dict[Tuple[int, int], List[Op]] = {}
    for op in ops:
        consumed_by.setdefault((op.src, op.src_coff), []).append(op)
    skip = set()
    for op in ops:
        if id(op) in skip:
            continue
        if op.kind == "avgpool" and op.k == 3 and op.stride == 1 and op.pad == 1:
            users = consumed_by.get((op.dst, op.dst_coff), [])
            if len(users) == 1 and users[0].kind == "conv" and users[0].k == 1 and users[0].stride == 1 and users[0].pad == 0 \
                    and users[0].cin == op.cout:
                conv = users[0]
                t = tensors[conv.dst]
                tensors.append(Tensor(t.h, t.w, conv.cout, conv.name + "/linear"))
                tmp = len(tensors) - 1
                out.append(Op("conv", conv.name, op.src, tmp, op.src_coff, 0, conv.cin, conv.cout, 1, 1, 0, relu=False,
                              bn=conv.bn, out_blob=conv.name + "/linear", bias=False))
                out.append(Op("avgpool", op.name, tmp, conv.dst, 0, conv.dst_coff, conv.cout, conv.cout, 3, 1, 1,
                              relu=conv.relu, out_blob=conv.out_blob, bias_from=(conv.name, conv.bn)))
                loc.pop(op.out_blob, None)
                loc[conv.name + "/linear"] = (tmp, 0, conv.cout)
                skip.add(id(conv))
                continue
        out.append(op)
    # --- merge sibling 1x1 convolutions
    merged: List[Op] = []
    done = set()
    for idx, op in enumerate(out):
        if id(op) in done:
            continue
        if op.kind == "conv" and op.k == 1 and op.stride == 1 and op.pad == 0:
            sibs = [o for o in out[idx:] if o.kind == "conv" and o.k == 1 and o.stride == 1 and o.pad == 0
                    and (o.src, o.src_coff, o.cin) == (op.src, op.src_coff, op.cin) and id(o) not in done]
            if len(sibs) > 1:
                segs = [Segment(o.name, o.bn, o.cout, o.dst, o.dst_coff, o.relu, o.bias) for o in sibs]
                m = Op("conv", "+".join(o.name for o in sibs), op.src, sibs[0].dst, op.src_coff, sibs[0].dst_coff, op.cin,
                       sum(o.cout for o in sibs), 1, 1, 0, relu=False, out_blob=sibs[0].out_blob, segments=segs)
                merged.append(m)
                done.update(id(o) for o in sibs)
                continue
        merged.append(op)
    # --- a 3x3 max pool whose only reader is one 1x1 GEMM: the GEMM's loader takes the window maximum itself
    readers: Dict[int, List[Op]] = {}
    for op in merged:
        readers.setdefault(op.src, []).append(op)
    final: List[Op] = []
    gone = set()
    for op in merged:
        if id(op) in gone:
            continue
        if op.kind == "maxpool" and op.k == 3 and op.pad == 0 and op.dst_coff == 0 and tensors[op.dst].c == op.cout:
            users = readers.get(op.dst, [])
            u = users[0] if len(users) == 1 else None
            if (u is not None and u.kind == "conv" and u.k == 1 and u.stride == 1 and u.pad == 0 and u.src_coff == 0
                    and u.cin == op.cout and op.cin % 32 == 0 and op.dst != plan.feature_slot and u.cout <= FOLD_POOL_MAX_COUT):
                u.src, u.src_coff = op.src, op.src_coff
                u.pre_pool = (op.k, op.stride, op.name)
                loc.pop(op.out_blob, None)          # the pooled tensor is never written: its slot shrinks to a stub
                tensors[op.dst] = Tensor(1, 1, 4, op.out_blob + "/unused")
                gone.add(id(op))
                continue
        final.append(op)
    return Plan(tensors, final, plan.feature_slot, plan.feature_dim, loc)


# ------------------------------------------------------------------------------------------------
# prototxt reader (protobuf text format: `key: value` and `key { ... }`, '#' comments)
# ------------------------------------------------------------------------------------------------
_TOKEN = re.compile(r'\s*(?:(#[^\n]*)|([A-Za-z_][\w\-/.]*)|("(?:[^"\\]|\\.)*")|([-+]?[0-9][\w.+-]*)|([{}:]))')


def _tokens(text: str):
    pos = 0
    n = len(text)
    while pos < n:
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() == "":
                return
            raise ValueError("prototxt: cannot tokenise at %r" % text[pos:pos + 30])
        pos = m.end()
        if m.group(1):
            continue
        yield next(g for g in m.groups()[1:] if g is not None)


def _parse_message(toks, top=False):
    msg: List[Tuple[str, object]] = []
    for tok in toks:
        if tok == "}":
            if top:
                raise ValueError("prototxt: unbalanced '}'")
            return msg
        key = tok
        nxt = next(toks)
        if nxt == ":":
            val = next(toks)
            if val == "{":
                msg.append((key, _parse_message(toks)))
            else:
                msg.append((key, val[1:-1] if val.startswith('"') else val))
        elif nxt == "{":
            msg.append((key, _parse_message(toks)))
        else:
            raise ValueError("prototxt: expected ':' or '{' after %s" % key)
    if not top:
        raise ValueError("prototxt: missing '}'")
    return msg


def _get(msg, key, default=None):
    for k, v in msg:
        if k == key:
            return v
    return default


def _all(msg, key):
    return [v for k, v in msg if k == key]


def parse_prototxt(text: str) -> Graph:
    msg = _parse_message(_tokens(text), top=True)
    dims = [int(v) for v in _all(msg, "input_dim")]
    if len(dims) != 4:
        shape = _get(msg, "input_shape")
        dims = [int(v) for v in _all(shape, "dim")] if shape else dims
    if len(dims) != 4:
        raise ValueError("prototxt: need 4 input_dim values")
    g = Graph(_get(msg, "name", "net"), _get(msg, "input", "data"), (dims[1], dims[2], dims[3]))
    for lm in _all(msg, "layer") + _all(msg, "layers"):
        l = Layer(_get(lm, "name"), _get(lm, "type"), _all(lm, "bottom"), _all(lm, "top"))
        if l.type == "Convolution":
            p = _get(lm, "convolution_param", [])
            l.num_output = int(_get(p, "num_output"))
            l.kernel = int(_get(p, "kernel_size", 1))
            l.stride = int(_get(p, "stride", 1))
            l.pad = int(_get(p, "pad", 0))
        elif l.type == "Pooling":
            p = _get(lm, "pooling_param", [])
            l.pool = str(_get(p, "pool", "MAX"))
            l.kernel = int(_get(p, "kernel_size", 1))
            l.stride = int(_get(p, "stride", 1))
            l.pad = int(_get(p, "pad", 0))
        elif l.type == "InnerProduct":
            l.num_output = int(_get(_get(lm, "inner_product_param", []), "num_output"))
        g.layers.append(l)
    return g


def to_prototxt(g: Graph) -> str:
    """The deploy prototxt of a graph in the layout of the reference's files (models/ucf101/tsn_bn_inception_*_deploy.prototxt):
    what ``parse_prototxt`` reads back layer for layer.  For hosts that have the architecture but not the reference checkout
    (bench.py's fresh-process run of the command line)."""
    c, h, w = g.input_shape
    lines = ['name: "%s"' % g.name, 'input: "%s"' % g.input_name, "input_dim: 1", "input_dim: %d" % c, "input_dim: %d" % h, "input_dim: %d" % w]
    for l in g.layers:
        body = 'layer { name: "%s" type: "%s" %s %s' % (l.name, l.type, " ".join('bottom: "%s"' % b for b in l.bottoms),
                                                        " ".join('top: "%s"' % t for t in l.tops))
        if l.type == "Convolution":
            body += " convolution_param { num_output: %d pad: %d kernel_size: %d stride: %d }" % (l.num_output, l.pad, l.kernel, l.stride)
        elif l.type == "Pooling":
            body += " pooling_param { pool: %s kernel_size: %d stride: %d pad: %d }" % (l.pool, l.kernel, l.stride, l.pad)
        elif l.type == "InnerProduct":
            body += " inner_product_param { num_output: %d }" % l.num_output
        lines.append(body + " }")
    return "\n".join(lines) + "\n"


def load_prototxt(path: str) -> Graph:
    with open(path) as f:
        return parse_prototxt(f.read())


# ------------------------------------------------------------------------------------------------
# built-in BN-Inception (TSN deploy topology)
# ------------------------------------------------------------------------------------------------
# block -> (1x1, 3x3_reduce, 3x3, double_reduce, double_3x3 (both), pool kind, pool_proj); reduction
# blocks (stride 2, no 1x1, max-pool passed through un-projected) have 1x1 = 0 and pool_proj = 0.
_BLOCKS = [
    ("3a", 64, 64, 64, 64, 96, "AVE", 32),
    ("3b", 64, 64, 96, 64, 96, "AVE", 64),
    ("3c", 0, 128, 160, 64, 96, "MAX", 0),
    ("4a", 224, 64, 96, 96, 128, "AVE", 128),
    ("4b", 192, 96, 128, 96, 128, "AVE", 128),
    ("4c", 160, 128, 160, 128, 160, "AVE", 128),
    ("4d", 96, 128, 192, 160, 192, "AVE", 128),
    ("4e", 0, 128, 192, 192, 256, "MAX", 0),
    ("5a", 352, 192, 320, 160, 224, "AVE", 128),
    ("5b", 352, 192, 320, 192, 224, "MAX", 128),
]


def bn_inception(in_channels: int = 3, size: int = 224, with_fc: bool = True) -> Graph:
    g = Graph("BN-Inception", "data", (in_channels, size, size))

    def conv(name, bottom, n, k=1, s=1, p=0, relu_name=None):
        g.layers.append(Layer(name, "Convolution", [bottom], [name], n, k, s, p))
        g.layers.append(Layer(name + "_bn", "BN", [name], [name + "_bn"]))
        rn = relu_name or (name.rsplit("/", 1)[0] + "/relu_" + name.rsplit("/", 1)[1])
        g.layers.append(Layer(rn, "ReLU", [name + "_bn"], [name + "_bn"]))
        return name + "_bn"

    def pool(name, bottom, kind, k, s, p=0):
        g.layers.append(Layer(name, "Pooling", [bottom], [name], kernel=k, stride=s, pad=p, pool=kind))
        return name

    x = conv("conv1/7x7_s2", "data", 64, 7, 2, 3, "conv1/relu_7x7")
    x = pool("pool1/3x3_s2", x, "MAX", 3, 2)
    x = conv("conv2/3x3_reduce", x, 64)
    x = conv("conv2/3x3", x, 192, 3, 1, 1)
    x = pool("pool2/3x3_s2", x, "MAX", 3, 2)
    for blk, c1, c3r, c3, cdr, cd, pk, pp in _BLOCKS:
        pre = "inception_%s/" % blk
        reduction = c1 == 0
        s2 = 2 if reduction else 1
        outs = []
        if c1:
            outs.append(conv(pre + "1x1", x, c1))
        t = conv(pre + "3x3_reduce", x, c3r)
        outs.append(conv(pre + "3x3", t, c3, 3, s2, 1))
        t = conv(pre + "double_3x3_reduce", x, cdr)
        t = conv(pre + "double_3x3_1", t, cd, 3, 1, 1)
        outs.append(conv(pre + "double_3x3_2", t, cd, 3, s2, 1))
        if reduction:
            outs.append(pool(pre + "pool", x, "MAX", 3, 2))
        else:
            t = pool(pre + "pool", x, pk, 3, 1, 1)
            outs.append(conv(pre + "pool_proj", t, pp))
        g.layers.append(Layer(pre + "output", "Concat", outs, [pre + "output"]))
        x = pre + "output"
    g.layers.append(Layer("global_pool", "Pooling", [x], ["global_pool"], kernel=7, stride=1, pad=0, pool="AVE"))
    g.layers.append(Layer("dropout", "Dropout", ["global_pool"], ["global_pool"]))
    if with_fc:
        g.layers.append(Layer("fc-action", "InnerProduct", ["global_pool"], ["fc-action"], num_output=101))
    return g
