"""The reference's INFERENCE mode as a fused plan: `Model.call(batch, 'test', obs_override=feat_agg)`
(nlt/nlt_test.py:78-94; nlt/models/nlt.py:154-155,172-174).

`nlt_test.infer` renders every test batch with ONE set of aggregated observation feature maps, `feat_agg[l]` = [1,h_l,w_l,C_l],
standing in for the per-frame observation features: the observation network is not needed at all (the reference still runs it on a
placeholder neighbour and throws the result away), and each map enters the query network only through a concat:

    encoder level l     conv_s2(concat(q_{l-1}, ovr_{l-1}))          = Wq * q_{l-1} + [Wo * ovr_{l-1} + b]
    expanding block j   deconv_s2(concat(x, q_{D-j}, ovr_{D-j}))     = Wx * x + Wq * q_{D-j} + [Wo * ovr_{D-j} + b]
    head                conv1x1(concat(dec, q_0, ovr_0))             = ... + [Wo * ovr_0 + b]

The bracketed terms are linear in the given map and the same for every frame and every batch.  `_prepare_override` evaluates
them ONCE per (feat_agg, weights) with the ordinary conv kernels -- "override maps", one per consuming conv, at that conv's
output resolution -- and the per-frame pass reads them where a bias would be added (csrc/front_ovr.hip for the full- / half-
resolution levels, nlt_conv_forward_map for the rest): no per-frame copy of the maps into the interleaved feature buffers, half
the K of every stride-2 encoder conv, a quarter (bottleneck) to two thirds of the K of the expanding blocks' first convs.
The two fused expanding blocks (csrc/dec_block.hip) and the back kernel (csrc/fused.hip) have map variants as well
(nlt_dec_block_forward_map, nlt_back_forward_map): the given half of no interleaved feature buffer is ever written or read.

What qualifies (`can_fuse_override`): one [1,h,w,C] map per level, an expand()ed view of one, or a materialised [N,h,w,C] tensor whose
frames are copies of one map (the reference's tf.tile(x, (bs, 1, 1, 1)), nlt/nlt_test.py:83-86; checked on the device once per tensor
set).  Store-resident batches (Dataset.load_batch(resident=True)) are read in place (`_forward_resident_ovr`, nlt_front_ovr_forward_u8).

Same re-association as the folded front kernel (sum over [q | ovr] channels split into two sums): <= 1e-6 rel-L2 from the
layer-by-layer plan, <= 1e-4 from the oracle (tests/test_gpu_infer.py).
"""
import threading

import torch

from . import _capi as C


class _Derived:
    """A conv whose Keras-layout kernel is a slice (or a sum of slices) of a network layer's kernel along the input-channel
    axis; owns its arrays and fragment caches (inference: weights are fixed while an override state lives)."""

    def __init__(self, src, groups, bias, cout_pad=None):
        from .networks.elements import Conv2D
        k = src.kernel.detach()
        cols = []
        for ranges in groups:                               # ranges of one group are SUMMED (the same tensor enters twice)
            parts = [(k[..., lo:hi] if src.transpose else k[:, :, lo:hi, :]) for lo, hi in ranges]
            s = parts[0]
            for p in parts[1:]:
                s = s + p
            cols.append(s)
        kern = torch.cat(cols, 3 if src.transpose else 2) if len(cols) > 1 else cols[0]
        cout = src.n_ch_out
        if cout_pad is not None and cout_pad > cout:        # (the 3-channel head: a 4th zero column keeps 16-byte texels)
            assert not src.transpose
            kern = torch.cat((kern, kern.new_zeros(kern.shape[:3] + (cout_pad - cout,))), 3)
            bias = torch.cat((bias, bias.new_zeros(cout_pad - cout)))
            cout = cout_pad
        c = Conv2D(cout, src.kernel_size, src.stride, transpose=src.transpose)
        c.kernel, c.bias = kern.contiguous(), bias.detach().clone().contiguous()
        c.cin = kern.shape[3] if src.transpose else kern.shape[2]
        c.built = True
        self.conv = c


class OverrideMixin:
    """RenderPlan's fused `obs_override` forward (see the module docstring)."""

    def can_fuse_override(self, cl, obs_override, h, w, arrays=()):
        q, D, U = self.q, self.n_down, self.n_up
        if not (self.fuse_ends and self.fuse_override and self.use_obs and D >= 2 and U >= 2 and self.precision != 'bf16'):
            return False
        if len(obs_override) != D + 1 or (h | w) & 3:
            return False
        hh, ww = h, w
        tiled = False
        for l, t in enumerate(obs_override):                # one map per level, shared by every frame
            if not (torch.is_tensor(t) and t.dim() == 4 and t.dtype == torch.float32 and tuple(t.shape[1:]) == (hh, ww, cl[l])
                    and t.shape[0] >= 1 and (not arrays or t.device == arrays[0].device)):
                return False
            tiled = tiled or (t.shape[0] != 1 and t.stride(0) != 0)
            hh, ww = hh // 2, ww // 2
        last = q.layers[D + U].convs()
        prev = q.layers[D + U - 1].convs()
        acts = [a for blk in (q.layers[1], q.layers[2], q.layers[D + U]) for _, a in blk.convs()]
        (qa2, _), _ = q.layers[2].convs()
        if not (cl[0] == 16 and cl[1] == 16 and cl[2] == 32 and last[0][0].n_ch_out == 4 and last[1][0].n_ch_out == 4
                and prev[1][0].n_ch_out == 8 and q.layers[-1].n_ch_out == 3
                and all(a is not None for a in acts) and len({a.alpha for a in acts}) == 1 and 0.0 <= acts[0].alpha <= 1.0
                and all(c % 4 == 0 for c in cl) and C.front4_supported(*arrays)):
            return False
        # a MATERIALISED [N,h,w,C] override -- what the reference's own call site hands over: tf.tile(x, (bs, 1, 1, 1)),
        # nlt/nlt_test.py:83-86 -- takes this plan when its frames are copies of one map (checked on the device once per tensor set)
        return not tiled or self._frames_identical(obs_override)

    def _frames_identical(self, obs_override):
        key = tuple((t.data_ptr(), t._version, tuple(t.shape), t.stride(0)) for t in obs_override)
        with OverrideMixin._lock:
            c = getattr(self.q, '_ovr_same', None)
            if c is not None and c[0] == key:
                return c[1]
        if obs_override[0].is_cuda and torch.cuda.is_current_stream_capturing():
            return False                                    # (the verdict is a device -> host read: not inside a graph capture)
        same = bool(torch.stack([(t[1:] == t[:1]).all() for t in obs_override if t.shape[0] > 1 and t.stride(0) != 0]).all())
        with OverrideMixin._lock:                           # (the tensors are kept: their addresses cannot be handed out again)
            self.q._ovr_same = (key, same, list(obs_override))
        return same

    def _ovr_stamp(self, obs_override):
        q = self.q
        convs = [q.layers[0], q.layers[-1]] + [c for l in q.layers[1:-1] for c, _ in l.convs()]
        return (tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in obs_override),
                tuple((c.kernel.data_ptr(), c.kernel._version, c.bias._version, c._epoch[0]) for c in convs))

    def _prepare_override(self, b, obs_override, dev):
        """The override maps and the derived (query-rows-only) convs for this feat_agg and these weights; cached until either
        changes.  Runs outside any launch tape: a handful of ordinary conv launches, once per video."""
        stamp = self._ovr_stamp(obs_override)
        with OverrideMixin._lock:
            # one state per query network: the lanes of a RenderPipeline (plans of their own over the same nets) share the maps
            st = getattr(self.q, '_ovr_state', None)
            if st is None or st['stamp'] != stamp:
                st = self.q._ovr_state = self._build_override(obs_override, dev, stamp)
        if self._ovr is not st:
            self._ovr = st
            self._drop_tapes()
        return st

    def _build_override(self, obs_override, dev, stamp):
        q, D, U = self.q, self.n_down, self.n_up
        ovr = [t[0:1].contiguous() for t in obs_override]
        cl = [t.shape[3] for t in ovr]
        st = {'stamp': stamp, 'serial': OverrideMixin._serial[0], 'ovr': ovr, 'enc': {}, 'dec': {}}
        OverrideMixin._serial[0] += 1
        paused = C.tape_pause()
        try:
            def run_map(d, x):
                """d.conv over the given map x [1,h,w,c] (no activation) -> [1,oh,ow,cout]."""
                c = d.conv
                _, h, w, cin = x.shape
                oh, ow = c.out_hw(h, w)
                out = torch.empty((1, oh, ow, c.n_ch_out), device=dev, dtype=torch.float32)
                C.conv_forward(c.mode, x, cin, cin, None, 0, 0, 1, h, w, c.kernel, c.packed(cin, 0), c.bias, c.n_ch_out,
                               out, c.n_ch_out, act=False, alpha=0.0)
                return out
            q0, head = q.layers[0], q.layers[-1]
            (qa, _), (qb, _) = q.layers[1].convs()
            (qa2, _), _ = q.layers[2].convs()
            q0.build(5, dev); qa.build(32, dev); qb.build(16, dev); qa2.build(32, dev); head.build(36, dev)
            bq0 = q0.bias.detach()
            # L0's bias seen through L1's stride-2 conv (all four taps) and through the head: part of the maps
            b1 = qa.bias.detach() + torch.einsum('c,tco->o', bq0, qa.kernel.detach().reshape(4, 32, 16)[:, :16, :])
            bh = head.bias.detach() + bq0 @ head.kernel.detach()[0, 0, 4:20, :]
            st['p1'] = run_map(_Derived(qa, [[(16, 32)]], b1), ovr[0])
            st['s0'] = run_map(_Derived(head, [[(20, 36)]], bh, cout_pad=4), ovr[0])
            st['p2'] = run_map(_Derived(qa2, [[(16, 32)]], qa2.bias), ovr[1])
            for l in range(3, D + 1):
                (sa, _), _ = q.layers[l].convs()
                c = cl[l - 1]
                sa.build(2 * c, dev)
                zero = torch.zeros_like(sa.bias)
                st['enc'][l] = (_Derived(sa, [[(0, c)]], zero).conv, run_map(_Derived(sa, [[(c, 2 * c)]], sa.bias), ovr[l - 1]))
            cx = 2 * cl[D]
            for j in range(U):
                (da, _), (db, _) = q.layers[D + 1 + j].convs()
                c = cl[D - j]
                da.build(cx + 2 * c, dev)
                zero = torch.zeros_like(da.bias)
                if j == 0:          # bottleneck self-concat (nlt.py:180-190): x = skip = fm[D] = [q | ovr | q | ovr]
                    dq = _Derived(da, [[(0, c), (2 * c, 3 * c)]], zero).conv
                    dm = _Derived(da, [[(c, 2 * c), (3 * c, 4 * c)]], da.bias)
                else:
                    dq = _Derived(da, [[(0, cx + c)]], zero).conv
                    dm = _Derived(da, [[(cx + c, cx + 2 * c)]], da.bias)
                st['dec'][j] = (dq, run_map(dm, ovr[D - j]))
                cx = db.n_ch_out
        finally:
            C.tape_resume(paused)
        return st

    def resident_override_ok(self, res, obs_override, alpha):
        """Can the fused override forward read this store-resident batch in place (csrc/front_ovr.hip, uint8 variant)?"""
        return (res.w % 8 == 0 and all(torch.is_tensor(t) and t.device == res.cvis.device for t in obs_override)
                and 0.0 <= alpha <= 1.0 and all(t.data_ptr() % 16 == 0 for t in (res.diffuse, res.cvis, res.lvis))
                and (res.h * res.w) % 16 == 0
                and self.can_fuse_override(self._level_channels(), obs_override, res.h, res.w))

    def _forward_resident_ovr(self, res, obs_override, skip_connect_base, algo):
        """`forward(inference=True, obs_override=...)` on a store-resident batch (nlt_test.infer over Dataset.load_batch(resident=True)):
        the same plan as `_forward_ovr`, its front launch reading frame ids of the uint8 capture store -- no float batch is assembled."""
        n, h, w = res.n, res.h, res.w
        dev = res.cvis.device
        if not self.resident_override_ok(res, obs_override, self.q.layers[1].convs()[0][1].alpha):
            raise C.NLTError("this network / plan / obs_override cannot take store-resident inputs: materialise the batch")
        b = self._buffers(n, 0, h, w, dev)
        if not self._tuning:
            self.generation += 1
        reg = getattr(self.q.layers[0], '_registry', None)
        if reg is not None:
            if not self._tuning:
                reg.tick()
            reg.refresh_if_stale()
        ovr = self._prepare_override(b, obs_override, dev)
        b['train_fused'] = False
        if self.autotune and not b.get('tuned_ovr') and dev.type == 'cuda':
            b['tuned_ovr'] = True
            self._autotune(lambda: self._forward_resident_ovr(res, obs_override, skip_connect_base, algo))
        self._front_weights(dev, l2=True)
        body = lambda: self._forward_ovr(b, None, None, None, ovr, skip_connect_base, algo, resident=res)
        if self.use_tape and dev.type == 'cuda' and self.timer is None and not self._tuning and reg is not None:
            tkey = ('ovr_u8', res.diffuse.data_ptr(), res.cvis.data_ptr(), res.lvis.data_ptr(), res.ids.data_ptr(), n,
                    bool(skip_connect_base), algo, C._stream(), self._pred_out is not None, ovr['serial'])
            return self._run_taped(b, reg, tkey, body)
        return body()

    def _forward_ovr(self, b, base, cvis, lvis, st, skip_connect_base, algo, resident=None):
        """front_ovr -> levels 2..D (query path only) -> expanding blocks -> back kernel; one stream."""
        if resident is not None:
            n, h, w, dev = resident.n, resident.h, resident.w, resident.cvis.device
        else:
            n, h, w, _ = base.shape
            dev = base.device
        q, D, U, cl = self.q, self.n_down, self.n_up, b['C']
        alpha = q.layers[1].convs()[0][1].alpha
        if b['skip3'] is None:
            b['skip3'] = torch.empty((n, h, w, 3), device=dev, dtype=torch.float32)
        blob, blob_l2 = self._front_weights(dev, l2=True)

        h2, w2, h4, w4 = h // 2, w // 2, h // 4, w // 4
        # SURVEY 8d accounting of what this launch replaces (query path of L0, L1 and L2's stride-2 conv)
        nbytes = 4 * n * h * w * (5 + 16) + 4 * n * h2 * w2 * (32 + 16 + 16 + 16) + 4 * n * (h2 * w2 * 32 + h4 * w4 * 32)
        flops = 2 * n * h2 * w2 * (20 + 64) * 16 + 2 * n * h * w * 15 + 2 * n * h4 * w4 * 64 * 32
        moved = 4 * n * h * w * (5 + 3) + 4 * n * h2 * w2 * 16 + 4 * n * h4 * w4 * 32 + 4 * (h2 * w2 * 16 + h * w * 4 + h4 * w4 * 32)
        if resident is not None:         # uint8 store in, 5 bytes a texel instead of 20
            moved -= 15 * n * h * w
            self._launch('F.front', nbytes, C.front_ovr_forward_u8, resident.diffuse, resident.cvis, resident.lvis, resident.ids, n, h, w,
                         blob, blob_l2, st['p1'], st['s0'], st['p2'], skip_connect_base, alpha, b['fm'][1], 2 * cl[1], b['skip3'],
                         b['qtmp'][2], flops=flops, moved=moved)
        else:
            self._launch('F.front', nbytes, C.front_ovr_forward, base, cvis, lvis, n, h, w, blob, blob_l2, st['p1'], st['s0'], st['p2'],
                         skip_connect_base, alpha, b['fm'][1], 2 * cl[1], b['skip3'], b['qtmp'][2], flops=flops, moved=moved)
        hh, ww = h2, w2
        for l in range(2, D + 1):
            (qa, qact_a), (qb, qact_b) = q.layers[l].convs()
            c = cl[l]
            if l > 2:
                dq, pmap = st['enc'][l]
                self._conv('V%d.q.s2' % l, dq, qact_a, b['fm'][l - 1], cl[l - 1], 2 * cl[l - 1], None, 0, 0, n, hh, ww,
                           b['qtmp'][l], c, algo, bmap=pmap)
            hh, ww = hh // 2, ww // 2
            self._conv_enc('L%d.q.s1' % l, qb, qact_b, b['qtmp'][l], c, c, n, 1, hh, ww, b['fm'][l], 2 * c, algo)
        x, cx = b['fm'][D], 2 * cl[D]
        for j in range(U - 1):
            (da, dact_a), (db, dact_b) = q.layers[D + 1 + j].convs()
            skip, cs = b['fm'][D - j], 2 * cl[D - j]
            lab = 'L%d.q' % (D + 1 + j)
            nl = da.n_ch_out
            if (self.fuse_dec and nl in (8, 16) and db.n_ch_out == nl and cx == 2 * nl and cs == 8 * nl and algo == C.ALGO_AUTO
                    and dact_a is not None and dact_b is not None and dact_a.alpha == dact_b.alpha and j > 0):
                # csrc/dec_block.hip on [x | QUERY half of fm[D - j]] + the block's override map: the given half is never read
                dq, dmap = st['dec'][j]
                db.build(nl, dev)
                nbytes = 4 * n * hh * ww * ((cx + cs // 2) + 4 * nl) + 4 * n * 4 * hh * ww * 2 * nl
                self._launch(lab, nbytes, C.dec_block_forward_map, x, skip, cs, n, hh, ww, dq.kernel.detach(), db.kernel.detach(),
                             db.bias.detach(), nl, dact_a.alpha, dmap, b['dec'][j],
                             flops=2 * n * hh * ww * (cx + cs // 2) * 4 * nl + 2 * n * 4 * hh * ww * 4 * nl * nl,
                             moved=4 * n * hh * ww * ((cx + cs // 2) + 4 * nl) + 4 * 4 * hh * ww * nl)
                hh, ww = hh * 2, ww * 2
                x, cx = b['dec'][j], nl
                continue
            dq, dmap = st['dec'][j]
            c = cl[D - j]
            if j == 0:
                self._conv('V' + lab[1:] + '.s2', dq, dact_a, skip, c, 2 * c, None, 0, 0, n, hh, ww, b['dtmp'][j], nl, algo, bmap=dmap)
            else:
                self._conv('V' + lab[1:] + '.s2', dq, dact_a, x, cx, cx, skip, c, 2 * c, n, hh, ww, b['dtmp'][j], nl, algo, bmap=dmap)
            hh, ww = hh * 2, ww * 2
            self._conv(lab + '.s1', db, dact_b, b['dtmp'][j], nl, nl, None, 0, 0, n, hh, ww, b['dec'][j], db.n_ch_out, algo)
            x, cx = b['dec'][j], db.n_ch_out
        (da, _), (db, _) = q.layers[D + U].convs()
        head = q.layers[-1]
        db.build(4, dev)
        dq, dmap = st['dec'][U - 1]
        assert (hh, ww) == (h2, w2) and cx == 8 and dq.cin == 24
        nbytes = 4 * n * h * w * ((6 + 4) + (4 + 4) + (36 + 3))
        back_args = (x, b['fm'][1], 2 * cl[1], b['skip3'], n, hh, ww, dq.kernel.detach(), db.kernel.detach(), db.bias.detach(),
                     head.kernel.detach(), alpha, dmap)
        back_kw = dict(flops=2 * n * hh * ww * 24 * 16 + 2 * n * h * w * (64 + 12), moved=4 * n * h * w * (6 + 3 + 3) + 4 * h * w * 4)

        def back(pred):
            self._launch('F.back', nbytes, C.back_forward_map, *back_args, pred, **back_kw)
        out = self._pred_out if not self._tuning else None
        if out is None:
            back(b['pred'])
            return b['pred'], b
        b['back_infer'] = back
        paused = C.tape_pause()
        try:
            back(out)
        finally:
            C.tape_resume(paused)
        return out, b


OverrideMixin._serial = [1]
OverrideMixin._lock = threading.Lock()
