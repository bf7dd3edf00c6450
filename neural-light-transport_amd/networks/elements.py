"""Layer factories with the reference's names and semantics (nlt/networks/elements.py:26-125),
executed by libnlt_hip.so.  Weights live in Keras layouts (conv: (kh,kw,Cin,Cout); deconv:
(kh,kw,Cout,Cin)) as torch CUDA tensors so checkpoints / oracles exchange arrays unchanged.

The released-config branch (conv, deconv, leakyrelu / relu, iden, norm / pool 'none') runs on the fused RenderPlan
(engine.py).  act = elu, norm = pixel, pool = max / avg and `upconv` are stand-alone layers executed layer by layer
(generic.py; csrc/branches.hip), and so are norm = layer / batch (csrc/norms.hip), whose gamma / beta are two more
slots per layer of the model's flat parameter bucket.  norm = instance raises NotImplementedError: it is tf.contrib,
which TF 2.2 does not have -- the reference itself cannot run it.
"""
import math

import torch

from .. import _capi as C

_seed_counter = [0]


def _glorot_uniform(shape, device, seed):
    """Keras default kernel initialiser: U(-l, l), l = sqrt(6 / (fan_in + fan_out)) with the
    receptive field folded into both fans."""
    kh, kw, a, b = shape
    limit = math.sqrt(6.0 / (kh * kw * a + kh * kw * b))
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(shape, generator=g) * 2 - 1) * limit).to(device)


class Layer:
    built = True
    trainable = ()

    def build(self, cin, device):
        return cin

    def variables(self):
        return []


class Conv2D(Layer):
    """tf.keras.layers.Conv2D / Conv2DTranspose(n_ch_out, kernel_size, strides, padding='same')."""

    def __init__(self, n_ch_out, kernel_size, stride, transpose=False):
        if kernel_size not in (1, 2) or stride not in (1, 2) or (kernel_size == 1 and (stride != 1 or transpose)):
            raise NotImplementedError("kernel %d stride %d" % (kernel_size, stride))
        self.n_ch_out, self.kernel_size, self.stride, self.transpose = n_ch_out, kernel_size, stride, transpose
        if kernel_size == 1:
            self.mode = C.CONV1X1
        elif transpose:
            self.mode = C.DECONV_K2S2 if stride == 2 else C.DECONV_K2S1
        else:
            self.mode = C.CONV_K2S2 if stride == 2 else C.CONV_K2S1
        self.built = False
        self.kernel = self.bias = None          # Keras-layout weights (views into the model's flat bucket once bound)
        self.dkernel = self.dbias = None        # matching views into the flat gradient bucket
        self.cin = None
        self._packed = {}
        self._epoch = [0]                       # shared "weights were rewritten" counter (optimizer bumps it)
        self._registry = None                   # PackRegistry of the owning model (one-launch refresh of all fragments)

    def build(self, cin, device='cuda', seed=None):
        if not self.built:
            if seed is None:
                _seed_counter[0] += 1
                seed = _seed_counter[0]
            k, n = self.kernel_size, self.n_ch_out
            shape = (k, k, n, cin) if self.transpose else (k, k, cin, n)
            self.kernel = _glorot_uniform(shape, device, seed)
            self.bias = torch.zeros(n, device=device)                        # Keras: zeros
            self.cin = cin
            self.built = True
        return self.n_ch_out

    def set_weights(self, kernel, bias):
        kernel = torch.as_tensor(kernel, dtype=torch.float32)
        bias = torch.as_tensor(bias, dtype=torch.float32)
        cin = kernel.shape[3] if self.transpose else kernel.shape[2]
        if self.built and tuple(kernel.shape) == tuple(self.kernel.shape):
            with torch.no_grad():               # in place: keeps views into a flat bucket valid
                self.kernel.copy_(kernel)
                self.bias.copy_(bias)
        else:
            dev = self.kernel.device if self.built else 'cuda'
            self.kernel = kernel.to(dev).contiguous()
            self.bias = bias.to(dev).contiguous()
        self.cin, self.built = cin, True
        self._packed = {}
        if self._registry is not None:
            self._registry.drop(self)

    def variables(self):
        return [self.kernel, self.bias]

    def _version(self):
        return (self.kernel.data_ptr(), self.kernel._version, self._epoch[0])

    def _cached_pack(self, key, make, desc):
        """Packed-fragment cache shared by the three layouts below.  A miss allocates + packs (`make`); a STALE entry
        (the kernel was rewritten: optimizer step) is refreshed in place -- together with every other packed buffer of
        the model in ONE launch when the layer belongs to a PackRegistry, by re-running `make` otherwise."""
        ent = self._packed.get(key)
        ver = self._version()
        reg = self._registry
        if reg is not None:
            reg.touch(self, key)
            if ent is not None and reg.is_inactive(self, key):
                # A buffer the census had retired is asked for again (advisor r05): it is stale or about to be -- the one-launch
                # refresh skips inactive buffers -- whatever its version stamp says.  Back into the refresh set, and THIS buffer
                # alone re-packed now, in place (recorded tapes and graphs keep its address), with any open launch tape paused:
                # the pack launch belongs to no step.
                paused = C.tape_pause()
                try:
                    reg.activate(self, key)
                    ent[1].copy_(make())
                finally:
                    C.tape_resume(paused)
                self._packed[key] = (ver, ent[1])
                return ent[1]
        if ent is not None and ent[0] == ver:
            return ent[1]
        if ent is not None and reg is not None and reg.owns(self, key):
            paused = C.tape_pause()                  # (a stale ACTIVE buffer outside the top-of-pass refresh: same rule)
            try:
                reg.refresh()
            finally:
                C.tape_resume(paused)
            return self._packed[key][1]
        buf = make()
        self._packed[key] = (ver, buf)
        if reg is not None:
            reg.add(self, key, buf, desc)
        return buf

    def packed(self, c0, c1):
        """MFMA fragment layout of the kernel for a (c0 | c1) input split; re-packed when the
        kernel tensor has been written (optimizer step / set_weights)."""
        return self._cached_pack((c0, c1), lambda: C.pack_conv_weights(self.mode, self.kernel.detach(), c0, c1, self.n_ch_out),
                                 dict(kind=C.REPACK_MFMA, mode=self.mode, c0=c0, c1=c1, cout=self.n_ch_out, tn=0, lo=0,
                                      full=self.n_ch_out))

    def packed_tile(self, tn):
        """Fragment layout of the kernel for the LDS-tiled conv (csrc/conv_tile.hip), re-made on weight change."""
        return self._cached_pack(('tile', tn),
                                 lambda: C.pack_conv_tile_weights(self.mode, self.kernel.detach(), self.cin, self.n_ch_out, tn),
                                 dict(kind=C.REPACK_TILE, mode=self.mode, c0=self.cin, c1=0, cout=self.n_ch_out, tn=tn, lo=0,
                                      full=self.n_ch_out))

    def packed_wino(self, tn):
        """G g G^T fragments of a stride-1 k2 kernel for the Winograd kernel (csrc/conv_wino.hip), re-made on weight change."""
        return self._cached_pack(('wino', tn),
                                 lambda: C.pack_conv_wino_weights(self.mode, self.kernel.detach(), self.cin, self.n_ch_out, tn),
                                 dict(kind=C.REPACK_WINO, mode=self.mode, c0=self.cin, c1=0, cout=self.n_ch_out, tn=tn, lo=0,
                                      full=self.n_ch_out))

    def packed_adjoint_wino(self, lo, hi, tn):
        """Winograd fragments for backward-data w.r.t. forward input channels [lo, hi): read in place from the layer's own array."""
        adj = self.ADJOINT[self.mode]
        return self._cached_pack(('adjwino', lo, hi, tn),
                                 lambda: C.pack_conv_wino_weights(adj, self.kernel.detach(), self.n_ch_out, hi - lo, tn, self.cin, lo),
                                 dict(kind=C.REPACK_WINO, mode=adj, c0=self.n_ch_out, c1=0, cout=hi - lo, tn=tn, lo=lo, full=self.cin))

    def packed_bf16(self, c0, c1):
        """bf16 MFMA fragments (csrc/conv_bf16.hip) for a (c0 | c1) input split; re-packed when the kernel was rewritten.
        (Inference path: not part of the one-launch PackRegistry refresh.)"""
        cache = self.__dict__.setdefault('_packed_bf', {})
        ent, ver = cache.get((c0, c1)), self._version()
        if ent is None or ent[0] != ver:
            ent = cache[(c0, c1)] = (ver, C.conv_bf16_pack(self.mode, self.kernel.detach(), c0, c1, self.n_ch_out))
        return ent[1]

    def packed_tile3(self, tn):
        """Three-term bf16 fragments of the kernel for csrc/conv_tile3.hip (precision = f32x3); re-packed when the kernel was
        rewritten (a forward-mode cache like packed_bf16: not part of the one-launch PackRegistry refresh)."""
        cache = self.__dict__.setdefault('_packed_t3', {})
        ent, ver = cache.get(tn), self._version()
        if ent is None or ent[0] != ver:
            ent = cache[tn] = (ver, C.pack_conv_tile3_weights(self.mode, self.kernel.detach(), self.cin, self.n_ch_out, tn))
        return ent[1]

    ADJOINT = {C.CONV_K2S2: C.DECONV_K2S2, C.CONV_K2S1: C.DECONV_K2S1,
               C.DECONV_K2S2: C.CONV_K2S2, C.DECONV_K2S1: C.CONV_K2S1}

    def packed_adjoint(self, lo, hi):
        """Fragments for backward-DATA w.r.t. forward input channels [lo, hi): the adjoint conv
        family reads the SAME Keras array (a conv kernel (kh,kw,Cin,Cout) is the transposed
        conv's (kh,kw,Cout',Cin') and vice versa), sliced along the forward-input axis.
        Returns (fragments, slice view); the view is NOT contiguous -- the GPU path reads the fragments only."""
        k = self.kernel.detach()
        ks = k[..., lo:hi] if self.transpose else k[:, :, lo:hi, :]
        adj = self.ADJOINT[self.mode]
        buf = self._cached_pack(('adj', lo, hi), lambda: C.pack_conv_weights(adj, ks.contiguous(), self.n_ch_out, 0, hi - lo),
                                dict(kind=C.REPACK_MFMA, mode=adj, c0=self.n_ch_out, c1=0, cout=hi - lo, tn=0, lo=lo,
                                     full=self.cin))
        return buf, ks

    def packed_adjoint_tile(self, lo, hi, tn):
        """LDS-tile fragments for backward-data w.r.t. forward input channels [lo, hi) (csrc/conv_tile.hip): read in place from
        the layer's own Keras array (no slice copy), refreshed with every other packed buffer in the one-launch registry pass."""
        adj = self.ADJOINT[self.mode]
        return self._cached_pack(('adjtile', lo, hi, tn),
                                 lambda: C.pack_conv_tile_weights_adjoint(adj, self.kernel.detach(), self.n_ch_out, hi - lo, tn, self.cin, lo),
                                 dict(kind=C.REPACK_TILE, mode=adj, c0=self.n_ch_out, c1=0, cout=hi - lo, tn=tn, lo=lo, full=self.cin))

    def out_hw(self, h, w):
        if self.mode == C.CONV_K2S2:
            return h // 2, w // 2
        if self.mode == C.DECONV_K2S2:
            return 2 * h, 2 * w
        return h, w

    def __call__(self, x, act=None):
        """x [N,H,W,Cin] dense -> [N,H',W',Cout]; `act` (an Act layer) is fused when given."""
        n, h, w, cin = x.shape
        self.build(cin, x.device)
        assert cin == self.cin, "layer built for %d input channels, got %d" % (self.cin, cin)
        oh, ow = self.out_hw(h, w)
        out = torch.empty((n, oh, ow, self.n_ch_out), device=x.device, dtype=torch.float32)
        use_mfma = cin % 4 == 0 and self.n_ch_out % 4 == 0
        C.conv_forward(self.mode, x.contiguous(), cin, cin, None, 0, 0, n, h, w,
                       self.kernel.detach(), self.packed(cin, 0) if use_mfma else None, self.bias.detach(),
                       self.n_ch_out, out, self.n_ch_out, act=act is not None,
                       alpha=act.alpha if act is not None else 0.0)
        return out


class PackRegistry:
    """Every packed-fragment buffer of a model's convs, so that ONE launch (nlt_repack_weights) refills them all after
    an optimizer step instead of one allocation + pack launch (+ slice copy) per layer and layout."""

    def __init__(self, state_fn=None):
        self.entries = {}            # (id(layer), key) -> (layer, key, buffer, descriptor fields)
        self.table = None
        self.state_fn = state_fn     # () -> hashable that changes whenever any kernel of the model is rewritten
        self.state = None
        self.version = 0             # bumped when the SET of buffers changes (recorded launch tapes point at them)
        # r05: the plan-time trials pack a fragment layout for every candidate kernel of every layer (201 buffers, 118 MB at
        # depth 256) and the step keeps 60-odd of them: the refresh launch cost 0.12 ms per train step.  After a plan has been
        # (re-)tuned a CENSUS runs over its next passes; buffers nothing asked for go INACTIVE: kept, not refreshed, hence
        # stale -- whoever asks for one later re-activates it (`_cached_pack`), and recorded tapes (which hold buffer
        # addresses and would read a stale one without asking) are invalidated by the version bump.
        self.inactive = set()
        self.used = None             # keys asked for since the census began (None: no census running)
        self.ticks_left = 0
        import threading
        self._rec = threading.local()    # keys touched while THIS host thread records a launch tape

    def owns(self, layer, key):
        return (id(layer), key) in self.entries

    def is_inactive(self, layer, key):
        return (id(layer), key) in self.inactive

    def touch(self, layer, key):
        k = (id(layer), key)
        if self.used is not None:
            self.used.add(k)
        rec = getattr(self._rec, 'keys', None)
        if rec is not None:
            rec.add(k)

    # A recorded launch tape reads its fragment buffers WITHOUT asking for them (`_cached_pack` is not on the replay path), so
    # a census that only sees asks would retire buffers that live tapes -- of this plan or of another plan over the same nets
    # (pipeline lanes) -- still use (advisor r05).  A plan brackets its recording with begin_record / end_record, keeps the
    # keys with the tape and re-touches them on every replay.
    def begin_record(self):
        self._rec.keys = set()

    def end_record(self):
        keys, self._rec.keys = getattr(self._rec, 'keys', None), None
        return frozenset(keys or ())

    def touch_keys(self, keys):
        if self.used is not None and keys:
            self.used.update(keys)

    def begin_census(self, passes=4):
        """Called when a plan's choices have just been made: the next `passes` plan passes (forward / backward, whichever
        come) say which buffers the chosen kernels read."""
        self.used, self.ticks_left = set(), passes

    def tick(self):
        """One plan pass (forward or backward, any kind) begins."""
        if self.used is None:
            return
        if self.ticks_left <= 0:
            self.prune()
        self.ticks_left -= 1

    def prune(self):
        if self.used is None:
            return
        used, self.used = self.used, None                    # (closed first: lanes on other host threads keep touching / adding)
        drop = [k for k in list(self.entries) if k not in used and k not in self.inactive]
        if drop:
            self.inactive.update(drop)
            self.table = None
            self.version += 1

    def activate(self, layer, key):
        k = (id(layer), key)
        if k in self.inactive:
            self.inactive.discard(k)
            self.table = None

    def add(self, layer, key, buf, desc):
        self.entries[(id(layer), key)] = (layer, key, buf, desc)
        self.table = None
        self.version += 1
        if self.state is None and self.state_fn is not None:
            self.state = self.state_fn()     # the first buffers were packed from the current weights

    def drop(self, layer):
        self.entries = {k: v for k, v in self.entries.items() if v[0] is not layer}
        self.inactive = {k for k in self.inactive if k in self.entries}
        self.table = None
        self.version += 1

    def prepare(self):
        """Builds the device descriptor table (a host-to-device copy: must not happen inside a graph capture)."""
        if self.table is None and len(self.entries) > len(self.inactive):
            ents = [v for k, v in self.entries.items() if k not in self.inactive]
            rows = [dict(d, src=layer.kernel, dst=buf) for layer, _, buf, d in ents]
            self.table = C.repack_table(rows, ents[0][2].device)

    def refresh(self):
        if len(self.entries) <= len(self.inactive):
            return
        self.prepare()
        C.repack_weights(*self.table)
        for k, (layer, key, buf, _) in self.entries.items():
            if k not in self.inactive:
                layer._packed[key] = (layer._version(), buf)
        if self.state_fn is not None:
            self.state = self.state_fn()

    def refresh_if_stale(self):
        """Called by the plan at the top of a forward pass, on the main stream, BEFORE any other stream is forked:
        every consumer of a packed buffer (the side-stream query convs, the backward pass) is ordered after the
        refresh.  (A lazy refresh at the first stale access could run on one stream while another stream's convs
        were already reading their fragments.)"""
        if self.state_fn is not None and self.entries and self.state_fn() != self.state:
            self.refresh()


class Act(Layer):
    """kind 'lrelu': LeakyReLU(alpha) / ReLU (alpha = 0), fused into the preceding conv's epilogue by the plan;
    kind 'elu': tf.keras.layers.ELU(alpha), a stand-alone launch."""

    def __init__(self, alpha, kind='lrelu'):
        self.alpha, self.kind = alpha, kind

    def __call__(self, x):
        return C.act_forward(x.contiguous(), C.ACT_ELU if self.kind == 'elu' else C.ACT_LRELU, self.alpha)

    def backward(self, g, y):
        return C.act_backward(g.contiguous(), y, C.ACT_ELU if self.kind == 'elu' else C.ACT_LRELU, self.alpha)


class PixelNorm(Layer):
    """elements.py:103-121: x * rsqrt(mean_c(x^2) + 1e-8)."""
    eps = 1.0e-8

    def __call__(self, x):
        return C.pixelnorm_forward(x.contiguous(), self.eps)

    def backward(self, g, x):
        return C.pixelnorm_backward(g.contiguous(), x, self.eps)


class ChannelNorm(Layer):
    """norm = 'layer' | 'batch' (elements.py:51-56), per texel over the channel axis: y = (x - m) r gamma + beta.

    layer: tf.keras.layers.LayerNormalization(epsilon=0.001, center=True, scale=True) -- m, r from the texel's own channels.
    batch: tf.keras.layers.BatchNormalization(momentum=0.99, epsilon=0.001) the way the reference's loop executes it.
      Nothing under nlt/ passes `training=True` (networks/seq.py:36-41, models/nlt.py:154-195 call `layer(x)`), so Keras
      runs the layer in inference mode in train, vali and test alike: m = moving_mean, r = rsqrt(moving_variance + eps),
      and the moving statistics keep their initial values (0, 1) for ever, because only training-mode calls update them.
      gamma / beta are trainable and do get gradients.  (A driver that did pass training=True would need batch statistics
      and, data-parallel, a cross-replica mean: not what the reference does.)
    Variables in Keras order: gamma (ones), beta (zeros) [, moving_mean (zeros), moving_variance (ones): not trainable].
    `kernel` / `bias` alias gamma / beta so that the flat-bucket slot code treats a norm like any other two-variable layer."""
    eps = 1.0e-3

    def __init__(self, kind):
        self.kind = C.NORM_LAYER if kind == 'layer' else C.NORM_BATCH
        self.name = kind
        self.built = False
        self.kernel = self.bias = self.dkernel = self.dbias = None
        self.moving_mean = self.moving_variance = None
        self.c = None
        self._epoch = [0]
        self._packed = {}
        self._registry = None

    gamma = property(lambda self: self.kernel)
    beta = property(lambda self: self.bias)

    def build(self, cin, device='cuda'):
        if not self.built:
            self.kernel = torch.ones(cin, device=device)
            self.bias = torch.zeros(cin, device=device)
            self.c, self.built = cin, True
        if self.kind == C.NORM_BATCH and self.moving_mean is None:
            self.moving_mean = torch.zeros(cin, device=device)
            self.moving_variance = torch.ones(cin, device=device)
        return cin

    def set_weights(self, gamma, beta):
        gamma = torch.as_tensor(gamma, dtype=torch.float32)
        beta = torch.as_tensor(beta, dtype=torch.float32)
        if self.built and tuple(gamma.shape) == tuple(self.kernel.shape):
            with torch.no_grad():
                self.kernel.copy_(gamma)
                self.bias.copy_(beta)
        else:
            dev = self.kernel.device if self.built else 'cuda'
            self.kernel, self.bias = gamma.to(dev).contiguous(), beta.to(dev).contiguous()
            self.c, self.built = gamma.numel(), True
            self.build(self.c, dev)

    def variables(self):
        return [self.kernel, self.bias]

    def __call__(self, x):
        self.build(x.shape[-1], x.device)
        return C.norm_forward(self.kind, x.contiguous(), self.kernel.detach(), self.bias.detach(), self.moving_mean,
                              self.moving_variance, self.eps)

    def backward(self, g, x):
        return C.norm_backward(self.kind, g.contiguous(), x, self.kernel.detach(), self.moving_mean, self.moving_variance, self.eps,
                               self.dkernel, self.dbias)


class Pool2D(Layer):
    """MaxPooling2D / AveragePooling2D(pool_size=2, strides=2, padding='same') (elements.py:81-94)."""

    def __init__(self, kind):
        self.kind = C.POOL_MAX if kind == 'max' else C.POOL_AVG

    def __call__(self, x):
        return C.pool2x2_forward(x.contiguous(), self.kind)

    def backward(self, g, x):
        return C.pool2x2_backward(g.contiguous(), x, self.kind)


class UpSample2D(Layer):
    """tf.keras.layers.UpSampling2D(size=2, interpolation='bilinear') = tf.image.resize (half-pixel centres)."""

    def __call__(self, x):
        return C.resize_bilinear_forward(x.contiguous(), 2 * x.shape[1], 2 * x.shape[2])

    def backward(self, g, x):
        return C.resize_bilinear_backward(g.contiguous(), x.shape[1], x.shape[2])


class Identity(Layer):
    def __call__(self, x):
        return x


class Sequential(Layer):
    """tf.keras.Sequential over the layer kinds above; conv -> (identities) -> act is executed as
    one fused kernel launch."""

    def __init__(self, layers):
        self.layers = list(layers)

    @property
    def built(self):
        return all(l.built for l in self.layers)

    def build(self, cin, device='cuda'):
        for l in self.layers:
            cin = l.build(cin, device)
        return cin

    def variables(self):
        return [v for l in self.layers for v in l.variables()]

    def all_convs(self, norms=True):
        """Every layer with variables inside -- Conv2D (kernel, bias) and, with norms=True, ChannelNorm (gamma, beta) --
        nested Sequentials (upconv) included, in execution order = the order Keras lists the block's variables in."""
        out = []
        for l in self.layers:
            if isinstance(l, Conv2D) or (norms and isinstance(l, ChannelNorm)):
                out.append(l)
            elif isinstance(l, Sequential):
                out += l.all_convs(norms)
        return out

    def is_plain(self):
        """Only convs, identities and fused-able LeakyReLU / ReLU activations (what RenderPlan executes)?"""
        return all(isinstance(l, (Conv2D, Identity)) or (isinstance(l, Act) and l.kind == 'lrelu') for l in self.layers)

    def convs(self):
        """[(Conv2D, Act or None)] in execution order."""
        out, i = [], 0
        L = self.layers
        while i < len(L):
            if isinstance(L[i], Conv2D):
                j = i + 1
                while j < len(L) and isinstance(L[j], Identity):
                    j += 1
                act = L[j] if j < len(L) and isinstance(L[j], Act) else None
                out.append((L[i], act))
                i = j + 1 if act is not None else j
            elif isinstance(L[i], Identity):
                i += 1
            else:
                raise NotImplementedError("stand-alone %s" % type(L[i]).__name__)
        return out

    def __call__(self, x):
        for conv_, act_ in self.convs():
            x = conv_(x, act=act_)
        return x


def conv(kernel_size, n_ch_out, stride=1):
    return Conv2D(n_ch_out, kernel_size, stride)


def deconv(kernel_size, n_ch_out, stride=1):
    return Conv2D(n_ch_out, kernel_size, stride, transpose=True)


def upconv(n_ch_out):
    """2x bilinear upsampling + Conv2D(n, 2, padding='same') (elements.py:42-48)."""
    return Sequential([UpSample2D(), Conv2D(n_ch_out, 2, 1)])


def norm(type_):
    if type_ is None or type_.lower() == 'none':
        return iden()
    if type_ == 'pixel':
        return PixelNorm()
    if type_ in ('batch', 'layer'):
        return ChannelNorm(type_)
    if type_ == 'instance':
        raise NotImplementedError("norm = instance is tf.contrib.layers.instance_norm (elements.py:97-100): TF 2.2, the "
                                  "reference's pinned version, has no tf.contrib -- the reference cannot run it either")
    raise NotImplementedError(type_)


def act(type_):
    if type_ == 'relu':
        return Act(0.0)                 # tf.keras.layers.ReLU(negative_slope=0)
    if type_ == 'leakyrelu':
        return Act(0.3)                 # tf.keras.layers.LeakyReLU(alpha=0.3)
    if type_ == 'elu':
        return Act(1.0, kind='elu')     # tf.keras.layers.ELU(alpha=1.0)
    raise NotImplementedError(type_)


def pool(type_):
    if type_ is None or type_.lower() == 'none':
        return iden()
    if type_ in ('max', 'avg'):
        return Pool2D(type_)
    raise NotImplementedError(type_)


def iden():
    return Identity()
