"""Fused execution plan of NLT's two-path U-Net on libnlt_hip.so.

What the reference does layer by layer with materialised tf.concat / reduce_mean tensors
(nlt/models/nlt.py:141-199) is laid out here as a fixed kernel sequence over pre-allocated
HBM buffers:

  fm[l]  [N,h_l,w_l,2*C_l]  the encoder feature map of level l, stored INTERLEAVED as
                            [query C_l | mean-of-observations C_l] -- the tf.concat of
                            nlt.py:174 never exists as a copy: the query kernel writes channel
                            slice [0,C), the observation-mean kernel slice [C,2C);
  obs[l] [N,k,h_l,w_l,C_l]  per-observation features (each observation continues on its own
                            path, nlt.py:166);
  decoder layers read their two inputs (previous decoder output, popped fm[.]) through the
  conv kernels' dual-source "virtual concat" (nlt.py:190), including the bottleneck self-concat
  quirk (first decoder layer sees concat(fm[D], fm[D])).
"""
import torch

from . import _capi as C


class OpTimer:
    """HIP-event timing of individual launches on the CURRENT torch stream (the one every
    kernel of the plan is launched on).  records: label -> [n_launches, total_ms, algorithmic_bytes]."""

    def __init__(self):
        self.pending, self.records = [], {}

    def launch(self, label, nbytes, fn, *args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(*args, **kw)
        e1.record()
        self.pending.append((label, nbytes, e0, e1))

    def collect(self):
        torch.cuda.synchronize()
        for label, nbytes, e0, e1 in self.pending:
            r = self.records.setdefault(label, [0, 0.0, nbytes])
            r[0] += 1
            r[1] += e0.elapsed_time(e1)
        self.pending = []
        return self.records


class RenderPlan:
    def __init__(self, net_query, net_obs, use_obs=True):
        self.timer = None               # set to an OpTimer (or a set of labels via timer.only) to time launches
        self.q, self.o, self.use_obs = net_query, net_obs, use_obs
        self.tile_hints = {}            # label -> 16*RT+CT override (tuning aid)
        is_c = net_query.is_contracting
        self.n_down = sum(is_c) - 1                      # contracting Sequential blocks
        self.n_up = len(is_c) - sum(is_c) - 1            # expanding Sequential blocks
        assert self.n_down == self.n_up
        self._bufs = {}

    # ------------------------------------------------------------------ buffers
    def _buffers(self, n, k, h, w, device):
        key = (n, k, h, w, str(device))
        b = self._bufs.get(key)
        if b is not None:
            return b
        E = lambda *s: torch.empty(s, device=device, dtype=torch.float32)
        q, D = self.q, self.n_down
        mult = 2 if self.use_obs else 1
        cl = [q.layers[0].n_ch_out] + [q.layers[l].convs()[0][0].n_ch_out for l in range(1, D + 1)]
        b = {'C': cl, 'fm': [], 'obs': [], 'qtmp': [None], 'otmp': [None], 'dtmp': [], 'dec': []}
        hh, ww = h, w
        for l in range(D + 1):
            if l > 0:
                if (hh | ww) & 1:
                    raise ValueError("UV size %dx%d is not divisible by 2^%d" % (h, w, D))
                hh, ww = hh // 2, ww // 2
                b['qtmp'].append(E(n, hh, ww, cl[l]))
                b['otmp'].append(E(n, k, hh, ww, cl[l]))
            b['fm'].append(E(n, hh, ww, mult * cl[l]))
            b['obs'].append(E(n, k, hh, ww, cl[l]))
        for j in range(self.n_up):
            nl = q.layers[D + 1 + j].convs()[0][0].n_ch_out
            hh, ww = hh * 2, ww * 2
            b['dtmp'].append(E(n, hh, ww, nl))
            b['dec'].append(E(n, hh, ww, nl))
        b['pred'] = E(n, h, w, 3)
        self._bufs = {key: b}           # keep one shape resident
        return b

    def _launch(self, label, nbytes, fn, *args, **kw):
        t = self.timer
        if t is not None and (getattr(t, 'only', None) is None or label in t.only):
            t.launch(label, nbytes, fn, *args, **kw)
        else:
            fn(*args, **kw)

    def _conv(self, label, layer, act, src0, c0, ld0, src1, c1, ld1, n, h, w, out, ldo, algo=C.ALGO_AUTO):
        layer.build(c0 + c1, src0.device)
        assert layer.cin == c0 + c1, (layer.cin, c0, c1)
        ok = c0 % 4 == 0 and c1 % 4 == 0 and layer.n_ch_out % 4 == 0 and algo != C.ALGO_DIRECT
        oh, ow = layer.out_hw(h, w)
        # SURVEY 8d accounting: every input element read once, every output element written once
        nbytes = 4 * (n * h * w * (c0 + c1) + n * oh * ow * layer.n_ch_out)
        tile_hint = self.tile_hints.get(label, self.tile_hints.get('*', 0))
        ncols = layer.n_ch_out * (4 if layer.mode == C.DECONV_K2S2 else 1)
        if tile_hint and ((ncols + 15) // 16) % (tile_hint & 15):
            tile_hint = 0                # CT must divide the number of 16-column tiles
        self._launch(label, nbytes, C.conv_forward, layer.mode, src0, c0, ld0, src1, c1, ld1, n, h, w, layer.kernel.detach(),
                       layer.packed(c0, c1) if ok else None, layer.bias.detach(), layer.n_ch_out, out, ldo,
                       act=act is not None, alpha=act.alpha if act is not None else 0.0,
                       algo=algo if ok else C.ALGO_DIRECT, tile_hint=tile_hint if ok else 0)

    # ------------------------------------------------------------------ forward
    def forward(self, base, cvis, lvis, nn_rgb, nn_base, obs_weights=None, obs_override=None,
                skip_connect_base=True, algo=C.ALGO_AUTO):
        """base [N,H,W,3], cvis/lvis [N,H,W,1], nn_rgb/nn_base [N,k,H,W,3] -> pred [N,H,W,3]
        (texel (0,0) zeroed, base added).  Returns (pred, buffers)."""
        n, h, w, _ = base.shape
        k = nn_rgb.shape[1]
        dev = base.device
        b = self._buffers(n, k, h, w, dev)
        q, o, D, cl = self.q, self.o, self.n_down, b['C']
        mult = 2 if self.use_obs else 1
        run_obs = self.use_obs and obs_override is None

        # L0 (both paths) + first observation mean
        q0, o0 = q.layers[0], o.layers[0]
        q0.build(5, dev); o0.build(3, dev)
        if self.use_obs:
            nbytes = 4 * n * h * w * (5 + 6 * k + 2 * cl[0] + k * cl[0])
            self._launch('L0.stem', nbytes, C.stem_forward, base, cvis, lvis, nn_rgb, nn_base, obs_weights,
                         n, k, h, w, cl[0], q0.kernel.detach(), q0.bias.detach(), o0.kernel.detach(),
                         o0.bias.detach(), b['fm'][0], b['obs'][0])
            if obs_override is not None:
                b['fm'][0][..., cl[0]:].copy_(obs_override[0].expand(n, -1, -1, -1))
        else:
            x5 = torch.cat((base, cvis, lvis), 3)
            self._conv('L0.q', q0, None, x5, 5, 5, None, 0, 0, n, h, w, b['fm'][0], cl[0], algo)

        hh, ww = h, w
        for l in range(1, D + 1):
            (qa, qact_a), (qb, qact_b) = q.layers[l].convs()
            cin = mult * cl[l - 1]
            self._conv('L%d.q.s2' % l, qa, qact_a, b['fm'][l - 1], cin, cin, None, 0, 0, n, hh, ww,
                       b['qtmp'][l], cl[l], algo)
            if run_obs:
                (oa, oact_a), (ob, oact_b) = o.layers[l].convs()
                self._conv('L%d.o.s2' % l, oa, oact_a, b['obs'][l - 1], cl[l - 1], cl[l - 1], None, 0, 0,
                           n * k, hh, ww, b['otmp'][l], cl[l], algo)
            hh, ww = hh // 2, ww // 2
            self._conv('L%d.q.s1' % l, qb, qact_b, b['qtmp'][l], cl[l], cl[l], None, 0, 0, n, hh, ww,
                       b['fm'][l], mult * cl[l], algo)
            if run_obs:
                self._conv('L%d.o.s1' % l, ob, oact_b, b['otmp'][l], cl[l], cl[l], None, 0, 0, n * k, hh, ww,
                           b['obs'][l], cl[l], algo)
                self._launch('L%d.o.mean' % l, 4 * n * hh * ww * cl[l] * (k + 1), C.obs_mean_forward,
                             b['obs'][l], obs_weights, n, k, hh * ww, cl[l], b['fm'][l].view(-1)[cl[l]:], 2 * cl[l])
            elif self.use_obs:
                b['fm'][l][..., cl[l]:].copy_(obs_override[l].expand(n, -1, -1, -1))

        # decoder: x | popped encoder map, read as a virtual concat
        x, cx = b['fm'][D], mult * cl[D]
        for j in range(self.n_up):
            (da, dact_a), (db, dact_b) = q.layers[D + 1 + j].convs()
            skip = b['fm'][D - j]
            cs = mult * cl[D - j]
            lab = 'L%d.q' % (D + 1 + j)
            self._conv(lab + '.s2', da, dact_a, x, cx, cx, skip, cs, cs, n, hh, ww, b['dtmp'][j], da.n_ch_out, algo)
            hh, ww = hh * 2, ww * 2
            self._conv(lab + '.s1', db, dact_b, b['dtmp'][j], da.n_ch_out, da.n_ch_out, None, 0, 0, n, hh, ww,
                       b['dec'][j], db.n_ch_out, algo)
            x, cx = b['dec'][j], db.n_ch_out

        head = q.layers[-1]
        cs = mult * cl[0]
        head.build(cx + cs, dev)
        if head.n_ch_out != 3:
            raise NotImplementedError("output head with %d channels" % head.n_ch_out)
        self._launch('L%d.head' % (2 * D + 1), 4 * n * h * w * (cx + cs + 3 + 3), C.head_forward,
                     x, cx, cx, b['fm'][0], cs, cs, head.kernel.detach(), head.bias.detach(),
                     base if skip_connect_base else None, n, h, w, b['pred'])
        return b['pred'], b
