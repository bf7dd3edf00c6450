"""Layer-by-layer execution of the two-path U-Net with a hand-rolled backward tape: the path for the config branches the
fused RenderPlan does not cover (act = elu, norm = pixel / layer / batch, pool = max / avg and the `upconv` that comes with
pooling -- nlt/networks/elements.py:42-56,69-94,103-121; nlt/networks/convnet.py:50-76).

It follows Model._call statement for statement (nlt/models/nlt.py:141-199: per-layer observation maps, their mean,
concat with the query map, the skip stack, the bottleneck self-concat) on the generic layer objects of
networks/elements.py; every layer is one libnlt_hip.so launch (convs fuse a directly following LeakyReLU / ReLU).
Each executed op leaves a node (value, parents, backward closure); `backward` walks the nodes in reverse and leaves the
weight gradients in the layers' views of the model's flat gradient bucket -- the same contract as RenderPlan.backward.
These branches are about coverage, not speed; the released configs never come here.
"""
import torch

from . import _capi as C
from .networks.elements import Act, ChannelNorm, Conv2D, Identity, PixelNorm, Pool2D, Sequential, UpSample2D


class Node:
    __slots__ = ('value', 'parents', 'back', 'grad')

    def __init__(self, value, parents=(), back=None):
        self.value, self.parents, self.back, self.grad = value, parents, back, None


class Tape:
    def __init__(self, record):
        self.record, self.nodes = record, []

    def leaf(self, value):
        return Node(value)

    def add(self, value, parents, back):
        node = Node(value, parents, back if self.record else None)
        if self.record:
            self.nodes.append(node)
        return node

    def backward(self, out_node, grad):
        out_node.grad = grad
        for node in reversed(self.nodes):
            if node.grad is None or node.back is None:
                continue
            for parent, g in zip(node.parents, node.back(node.grad)):
                if g is None or (parent.back is None and not parent.parents):
                    continue                                        # a network input: nothing upstream
                parent.grad = g if parent.grad is None else parent.grad + g       # (fan-in: the skip stack)
            node.grad = None


def _zero_bias(n, device):
    return torch.zeros(n, device=device, dtype=torch.float32)


def conv(tape, layer, x, act=None):
    """Conv2D / Conv2DTranspose (+ fused LeakyReLU / ReLU).  x: Node."""
    xv = x.value.contiguous()
    n, h, w, cin = xv.shape
    y = layer(xv, act=act)
    cout = layer.n_ch_out

    def back(g):
        g = g.contiguous()
        if act is not None:
            g = act.backward(g, y)                                   # mask from the output: y > 0 <=> pre-activation > 0
        C.conv_backward_weights(layer.mode, xv, cin, cin, None, 0, 0, n, h, w, g, cout, cout, layer.dkernel, layer.dbias)
        if x.back is None and not x.parents:
            return (None,)
        dx = torch.empty_like(xv)
        oh, ow = layer.out_hw(h, w)
        if layer.mode == C.CONV1X1:
            wt = layer.kernel.detach()[0, 0].t().contiguous().view(1, 1, cout, cin)
            C.conv_forward(C.CONV1X1, g, cout, cout, None, 0, 0, n, h, w, wt, None, _zero_bias(cin, g.device), cin, dx, cin,
                           act=False, algo=C.ALGO_DIRECT)
        else:
            mfma = cin % 4 == 0 and cout % 4 == 0
            packed, ks = layer.packed_adjoint(0, cin) if mfma else (None, layer.kernel.detach())
            C.conv_forward(layer.ADJOINT[layer.mode], g, cout, cout, None, 0, 0, n, oh, ow, ks.contiguous(), packed,
                           _zero_bias(cin, g.device), cin, dx, cin, act=False, algo=C.ALGO_AUTO if mfma else C.ALGO_DIRECT)
        return (dx,)
    return tape.add(y, (x,), back)


def unary(tape, layer, x, saved='input'):
    xv = x.value.contiguous()
    y = layer(xv)
    keep = y if saved == 'output' else xv
    return tape.add(y, (x,), lambda g: (layer.backward(g, keep),))


def concat(tape, nodes):
    vals = [nd.value for nd in nodes]
    widths = [v.shape[-1] for v in vals]

    def back(g):
        out, a = [], 0
        for wd in widths:
            out.append(g[..., a:a + wd].contiguous())
            a += wd
        return tuple(out)
    return tape.add(torch.cat(vals, -1), tuple(nodes), back)


def obs_mean(tape, nodes, obs_weights=None):
    """tf.reduce_mean over the observation axis of the (optionally weighted) stack (nlt.py:161-164)."""
    n, h, w, c = nodes[0].value.shape
    k = len(nodes)
    stacked = torch.stack([nd.value for nd in nodes], 1).contiguous()
    out = torch.empty((n, h, w, c), device=stacked.device, dtype=torch.float32)
    C.obs_mean_forward(stacked, obs_weights, n, k, h * w, c, out, c)

    def back(g):
        if obs_weights is not None:
            raise NotImplementedError("training through obs_weights")
        share = C.scale_rows(g.contiguous().view(1, -1), torch.full((1,), 1.0 / k, device=g.device)).view(g.shape)
        return (share,) * k
    return tape.add(out, tuple(nodes), back)


def run_layer(tape, layer, x):
    """One entry of Network.layers (a bare Conv2D or a Sequential, possibly nested: upconv) on node x."""
    if isinstance(layer, Conv2D):
        return conv(tape, layer, x)
    assert isinstance(layer, Sequential), type(layer)
    L, i = layer.layers, 0
    while i < len(L):
        l = L[i]
        if isinstance(l, Conv2D):
            j = i + 1
            while j < len(L) and isinstance(L[j], Identity):
                j += 1
            fuse = j < len(L) and isinstance(L[j], Act) and L[j].kind == 'lrelu'
            x = conv(tape, l, x, act=L[j] if fuse else None)
            i = j + 1 if fuse else i + 1
            continue
        if isinstance(l, Sequential):
            x = run_layer(tape, l, x)
        elif isinstance(l, Act):
            x = unary(tape, l, x, saved='output')
        elif isinstance(l, (PixelNorm, ChannelNorm, Pool2D, UpSample2D)):   # (ChannelNorm.backward also adds dgamma / dbeta)
            x = unary(tape, l, x)
        elif not isinstance(l, Identity):
            raise NotImplementedError(type(l).__name__)
        i += 1
    return x


def forward(model, query_x, obs_xs, obs_weights=None, obs_override=None, record=False):
    """Model._call (nlt.py:141-199).  Returns (output Node, Tape)."""
    q, o = model.net['query'], model.net['obs']
    tape = Tape(record)
    qx = tape.leaf(query_x)
    ox = [tape.leaf(x) for x in obs_xs]
    stack, qy = [], None
    for i, (layer_q, is_c) in enumerate(zip(q.layers, q.is_contracting)):
        if is_c:
            oy = [run_layer(tape, o.layers[i], x) for x in ox]
            ox = oy
            qy = run_layer(tape, layer_q, qx)
            if model.use_obs:
                if obs_override is not None:
                    agg = tape.leaf(obs_override[i].expand(qy.value.shape[0], -1, -1, -1))
                else:
                    agg = obs_mean(tape, oy, obs_weights)
                qx = concat(tape, [qy, agg])
            else:
                qx = qy
            stack.append(qx)
        else:
            if stack:
                qx = concat(tape, [qx, stack.pop()])
            qy = run_layer(tape, layer_q, qx)
            qx = qy
    return qy, tape
