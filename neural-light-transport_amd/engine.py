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
import os

import torch

from . import _capi as C
from .engine_infer import OverrideMixin


class OpTimer:
    """HIP-event timing of individual launches on the CURRENT torch stream (the one every
    kernel of the plan is launched on).  records: label -> [n_launches, total_ms, algorithmic_bytes]."""

    def __init__(self):
        self.pending, self.records = [], {}
        self.flops = {}                 # label -> fp32 FLOPs of one launch (2 x MACs), filled by the plan
        self.moved = {}                 # label -> bytes a FUSED launch itself has to move (its inputs + outputs)

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


class RenderPlan(OverrideMixin):
    def __init__(self, net_query, net_obs, use_obs=True):
        self.timer = None               # set to an OpTimer (or a set of labels via timer.only) to time launches
        self.q, self.o, self.use_obs = net_query, net_obs, use_obs
        self.tile_hints = {}            # label -> 16*RT+CT wave tile ('*' = every launch)
        self.algo_hints = {}            # label -> C.ALGO_DIRECT for the few tiny-channel layers where it wins
        self.autotune = os.environ.get('NLT_AUTOTUNE', '1') != '0'
        self.fuse_ends = os.environ.get('NLT_FUSED', '1') != '0'   # inference: csrc/fused.hip for layers 0-1 and the last block + head
        self._front_blob = None
        # inference with obs_override = one map per level shared by all frames (nlt_test.infer): the fused query-only plan of
        # engine_infer.py (0: the general layer-by-layer plan, which also serves per-frame override maps)
        self.fuse_override = os.environ.get('NLT_FUSED_OVERRIDE', '1') != '0'
        self._ovr = None                # override maps + derived convs of the feat_agg seen last (engine_infer.py)
        self.front_l2 = os.environ.get('NLT_FRONT_L2', '1') != '0'      # front kernel also runs level 2's stride-2 convs (k <= 4)
        # inference: expanding blocks with 8 / 16 output channels as ONE launch each (csrc/dec_block.hip: intermediate map in LDS)
        self.fuse_dec = os.environ.get('NLT_FUSED_DEC', '1') != '0'
        # second-generation front kernel (csrc/front4.hip: barrier-free, one wave per level-1 strip); 0 = first generation
        self.front_v4 = os.environ.get('NLT_FRONT4', '1') != '0'
        self.fuse_train = os.environ.get('NLT_FUSED_TRAIN', '1') != '0'   # fused ends in the train step too (csrc/train_fused.hip)
        # train forward on the second-generation front kernel (nlt_front4_forward_train: keeps the level-1 maps AND folds level
        # 2's stride-2 convs, so no front_kernel<false> and no L2.{q,o}.s2 launches).  r03: opt-in, SLOWER in the step (3.46 ->
        # 3.53 ms l2).  r04: the kernel is persistent (weights once per wave, next strip prefetched) and the step is faster with
        # it: 3.213 / 3.195 -> 3.173 / 3.173 ms l2 (two pairs, one box), so it is the default; 0 = front_kernel<false>
        self.front4_train = os.environ.get('NLT_FRONT4_TRAIN', '1') != '0'
        self.two_streams = os.environ.get('NLT_STREAMS', '2') != '1'   # inference: query-path encoder convs on a side stream
        self._side = None               # (side stream, [events]) created on first use
        self._bside = None              # backward: (side stream for the weight gradients, [events], cursor)
        self.wgrad_tiled = os.environ.get('NLT_WGRAD', 'tiled') != 'atomic'   # csrc/wgrad_tile.hip vs first-generation csrc/wgrad.hip
        self.wgrad_narrow = os.environ.get('NLT_WGRAD_NARROW', '1') != '0'
        # weight gradients on a side stream, off the backward-data chain (config 4: 5.17 -> 4.63 ms / step)
        self.bwd_streams = int(os.environ.get('NLT_BWD_STREAMS', '1'))   # 0: one stream; 1: weight gradients on a side stream; 2: on two, alternately
        # launch tape: replay a step's C calls with their resolved arguments instead of re-deriving them (see _capi.py)
        self.use_tape = os.environ.get('NLT_TAPE', '1') != '0'
        # backward, one observation per frame: the per-level LeakyReLU' / observation-mean adjoint pass folded into the
        # epilogue of the backward-data launch that completes dfm[l] (0: the separate nlt_level_split_backward launches)
        self.fold_split = os.environ.get('NLT_FOLD_SPLIT', '1') != '0'
        self.tile_dgrad = os.environ.get('NLT_TILE_DGRAD', '1') != '0'   # backward-data launches may go to the LDS-tiled kernel
        self._tuning = False
        self.tune_backward = os.environ.get('NLT_TUNE_BWD', '1') != '0'   # plan-time trials for the backward-data launches too
        # 'bf16' (BASELINE config 5): encoder levels >= 3 and the expanding blocks mirroring them run on csrc/conv_bf16.hip with
        # bf16-stored activations (inference, fused plan); everything at full / half / quarter resolution stays fp32.
        # 'f32x3' / 'f32x3_9': fp32 storage everywhere; the LDS-tiled encoder convs multiply fp32 operands as three bf16 terms
        # on the bf16 matrix cores (6 / all 9 term products, csrc/conv_tile3.hip) -- forward launches of inference AND training
        self.precision = os.environ.get('NLT_PRECISION', 'fp32')
        self._pred_out = None           # this forward's caller-owned output tensor (see `forward`)
        self._pred_slot = 'back_infer'
        self.prune_packs = os.environ.get('NLT_PRUNE_PACKS', '1') != '0'   # retire fragment buffers only plan-time trials read
        self.grad_hook = None           # grad_hook(i): fired by backward() once range i of the gradient bucket has its weight gradients queued
        self.grad_mid_level = 0         # encoder level that closes range 1 (0: no such range); set by Model._flatten
        self.generation = 0             # bumped by every forward: the activations in the plan's buffers belong to that pass
        self.tape_replays = 0
        self._trial_direct = False
        self._ran_direct = set()
        self._trial_lds = 0             # autotune: try the LDS-tiled kernel with this many output channels per workgroup
        self._ran_lds = set()
        self.lds_hints = {}             # label -> tn (32 / 64) [+256: observations unfolded]: launches that go to csrc/conv_tile.hip
        # Winograd F(2x2, 2x2) kernel for the stride-1 k2 convs (csrc/conv_wino.hip: 9/16 of the matrix-pipe work); 0 = never
        self.use_wino = os.environ.get('NLT_WINO', '1') != '0'
        self._trial_wino = 0            # autotune: try it with this many output channels per workgroup
        self._ran_wino = set()
        self.wino_hints = {}            # label -> tn (32 / 64) [+256: observations unfolded]: launches that go to csrc/conv_wino.hip
        # narrow stride-1 convs (cin 16 | 32 -> 32) with LDS-resident weights, a frame per stage (csrc/conv_c32.hip)
        self.use_c32 = os.environ.get('NLT_C32', '1') != '0'
        self._trial_c32 = 0             # autotune: 1 = observations folded (mean in registers), 2 = unfolded
        self._ran_c32 = set()
        self.c32_hints = {}             # label -> 1 | 2
        self.alias_obs = os.environ.get('NLT_ALIAS_OBS', '1') != '0'   # k = 1 inference: observation features live in fm[l]'s second half
        self._trial_splitk = 0          # autotune: K slices to try on the small deep launches
        self._ran_splitk = set()
        self.splitk_hints = {}          # label -> K slices (split-K, csrc/conv_mfma.hip; < 0: its two-launch form) for launches with few GEMM rows
        is_c = net_query.is_contracting
        self.n_down = sum(is_c) - 1                      # contracting Sequential blocks
        self.n_up = len(is_c) - sum(is_c) - 1            # expanding Sequential blocks
        assert self.n_down == self.n_up
        self._bufs = {}

    def __del__(self):
        try:
            C.drop_workspace_scope(id(self))    # split-K scratch cached under this plan's identity (an id may be reused)
        except Exception:
            pass

    # ------------------------------------------------------------------ buffers
    def _buffers(self, n, k, h, w, device):
        key = (n, k, h, w, str(device), self.precision)
        b = self._bufs.get(key)
        if b is not None:
            return b
        bf = self.precision == 'bf16'
        # bf16 region: levels >= 3, expanding blocks j <= D - 3 (the last of them, 16 output channels, hands fp32 on)
        E = lambda *s, lo=False: torch.empty(s, device=device, dtype=torch.bfloat16 if (bf and lo) else torch.float32)
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
                b['qtmp'].append(E(n, hh, ww, cl[l], lo=l >= 3))
                b['otmp'].append(E(n, k, hh, ww, cl[l], lo=l >= 3))
            b['fm'].append(E(n, hh, ww, mult * cl[l], lo=l >= 3))
            b['obs'].append(E(n, k, hh, ww, cl[l], lo=l >= 3))
        for j in range(self.n_up):
            nl = q.layers[D + 1 + j].convs()[0][0].n_ch_out
            hh, ww = hh * 2, ww * 2
            b['dtmp'].append(E(n, hh, ww, nl, lo=j <= D - 3))
            b['dec'].append(E(n, hh, ww, nl, lo=j < D - 3))
        b['pred'] = E(n, h, w, 3)
        b['skip3'] = None               # allocated by the fused inference path on first use
        self._bufs = {key: b}           # keep one shape resident
        return b

    def _level_channels(self):
        q = self.q
        return [q.layers[0].n_ch_out] + [q.layers[l].convs()[0][0].n_ch_out for l in range(1, self.n_down + 1)]

    def _launch(self, label, nbytes, fn, *args, flops=0, moved=None, **kw):
        t = self.timer
        if t is not None:
            t.flops[label] = flops
            if moved is not None:
                t.moved[label] = moved
        if t is not None and (getattr(t, 'only', None) is None or label in t.only):
            t.launch(label, nbytes, fn, *args, **kw)
        else:
            fn(*args, **kw)

    def _conv(self, label, layer, act, src0, c0, ld0, src1, c1, ld1, n, h, w, out, ldo, algo=C.ALGO_AUTO, bmap=None):
        """bmap: per-output-texel bias map [1 | n, oh, ow, cout] added before the activation (engine_infer.py; MFMA path only)."""
        layer.build(c0 + c1, src0.device)
        assert layer.cin == c0 + c1, (layer.cin, c0, c1)
        small = (c0 + c1) * layer.n_ch_out <= 1024
        if self.algo_hints.get(label) == C.ALGO_DIRECT or (self._trial_direct and small):
            algo = C.ALGO_DIRECT
            self._ran_direct.add(label)
        ok = c0 % 4 == 0 and c1 % 4 == 0 and layer.n_ch_out % 4 == 0 and algo != C.ALGO_DIRECT
        oh, ow = layer.out_hw(h, w)
        # SURVEY 8d accounting: every input element read once, every output element written once
        nbytes = 4 * (n * h * w * (c0 + c1) + n * oh * ow * layer.n_ch_out)
        tile_hint = self.tile_hints.get(label, self.tile_hints.get('*', 0))
        ncols = layer.n_ch_out * (4 if layer.mode == C.DECONV_K2S2 else 1)
        taps = 4 if layer.mode in (C.CONV_K2S2, C.CONV_K2S1, C.DECONV_K2S1) else 1
        rows = n * h * w // (4 if layer.mode == C.CONV_K2S2 else 1)
        flops = 2 * rows * taps * (c0 + c1) * ncols
        if tile_hint and ((ncols + 15) // 16) % (tile_hint & 15):
            tile_hint = 0                # CT must divide the number of 16-column tiles
        if c1 == 0 and ok and algo == C.ALGO_AUTO and self._wino(label, layer, act, src0, c0, ld0, n, 1, h, w, out, ldo, None, 0, flops):
            return
        ks = self.splitk_hints.get(label, 1)
        if self._trial_splitk and ok:
            rt, ct = (tile_hint >> 4, tile_hint & 15) if tile_hint else (1, 1)
            waves = -(-rows // (16 * rt)) * (-(-ncols // 16) // ct)
            npad = -(-ncols // 16) * 16
            ks = self._trial_splitk if (waves < 4096 and rows * npad * abs(self._trial_splitk) <= (1 << 24)) else 1
        if bmap is not None:
            if not ok:
                raise C.NLTError("a bias-map conv needs channel counts that are multiples of 4 (%s)" % label)
            if abs(ks) > 1:
                self._ran_splitk.add(label)
            self._launch(label, nbytes, C.conv_forward_map, layer.mode, ks or 1, src0, c0, ld0, src1, c1, ld1, n, h, w,
                         layer.packed(c0, c1), layer.bias.detach(), layer.n_ch_out, out, ldo, bmap, act=act is not None,
                         alpha=act.alpha if act is not None else 0.0, tile_hint=tile_hint, w_keras=layer.kernel.detach(), flops=flops)
            return
        if abs(ks) > 1 and ok:
            self._ran_splitk.add(label)
            self._launch(label, nbytes, C.conv_forward_splitk, layer.mode, ks, src0, c0, ld0, src1, c1, ld1, n, h, w,
                         layer.packed(c0, c1), layer.bias.detach(), layer.n_ch_out, out, ldo, act=act is not None,
                         alpha=act.alpha if act is not None else 0.0, tile_hint=tile_hint, flops=flops)
            return
        self._launch(label, nbytes, C.conv_forward, layer.mode, src0, c0, ld0, src1, c1, ld1, n, h, w, layer.kernel.detach(),
                       layer.packed(c0, c1) if ok else None, layer.bias.detach(), layer.n_ch_out, out, ldo,
                       act=act is not None, alpha=act.alpha if act is not None else 0.0,
                       algo=algo if ok else C.ALGO_DIRECT, tile_hint=tile_hint if ok else 0, flops=flops)

    def _wino(self, label, layer, act, src, cin, ld, frames, kobs, h, w, out, ldo, mean_out, ldm, flops, obs_weights=None):
        """The launch on the Winograd kernel if the plan (or the running trial) gave it to it; False otherwise.
        The observation mean stays in the kernel's registers; +256 runs the observations as frames and the mean in its own launch."""
        hint = self._trial_wino or self.wino_hints.get(label, 0)
        tn, unfold = hint & 255, bool(hint >> 8)
        if not (tn and self.use_wino and obs_weights is None and layer.mode in (C.CONV_K2S1, C.DECONV_K2S1) and layer.cin == cin
                and cin % 8 == 0 and layer.n_ch_out % tn == 0 and ld % 4 == 0 and ldo % 4 == 0):
            return False
        if layer.mode == C.DECONV_K2S1 and (kobs > 1 or mean_out is not None):
            return False
        if kobs == 1 and mean_out is None:
            if self._trial_wino >> 8:
                return False                            # nothing to unfold here: leave this launch to the other trials
            unfold = False
        nf = frames * kobs
        c = layer.n_ch_out
        fold_mean = mean_out is not None and not unfold
        nbytes = 4 * nf * h * w * (cin + c) + (4 * frames * h * w * c * (kobs + 1) if fold_mean else 0)
        self._ran_wino.add(label)
        self._launch(label, nbytes, C.conv_wino_forward, layer.mode, src, ld, cin, nf if unfold else frames, 1 if unfold else kobs,
                     h, w, layer.packed_wino(tn), layer.bias.detach(), c, tn, out, ldo, mean_out if fold_mean else None, ldm,
                     act=act is not None, alpha=act.alpha if act is not None else 0.0, flops=flops)
        if mean_out is not None and unfold:
            self._launch(label.replace('.s1', '.mean'), 4 * frames * h * w * c * (kobs + 1), C.obs_mean_forward,
                         out, None, frames, kobs, h * w, c, mean_out, ldm)
        return True

    def _conv_bf(self, label, layer, act, src0, c0, ld0, src1, c1, ld1, n, h, w, out, ldo):
        """One conv of the bf16 region (csrc/conv_bf16.hip); sources / output fp32 or bf16 as their tensors are."""
        layer.build(c0 + c1, src0.device)
        assert layer.cin == c0 + c1, (layer.cin, c0, c1)
        oh, ow = layer.out_hw(h, w)
        bpe = lambda t: 4 if t.dtype == torch.float32 else 2
        moved = n * h * w * (c0 * bpe(src0) + (c1 * bpe(src1) if c1 else 0)) + n * oh * ow * layer.n_ch_out * bpe(out)
        ncols = layer.n_ch_out * (4 if layer.mode == C.DECONV_K2S2 else 1)
        taps = 4 if layer.mode in (C.CONV_K2S2, C.CONV_K2S1, C.DECONV_K2S1) else 1
        rows = n * h * w // (4 if layer.mode == C.CONV_K2S2 else 1)
        self._launch(label, 4 * (n * h * w * (c0 + c1) + n * oh * ow * layer.n_ch_out), C.conv_bf16_forward, layer.mode,
                     src0, c0, ld0, src1, c1, ld1, n, h, w, layer.packed_bf16(c0, c1), layer.bias.detach(), layer.n_ch_out, out, ldo,
                     act=act is not None, alpha=act.alpha if act is not None else 0.0,
                     tile_hint=self.tile_hints.get('bf.' + label, 0), flops=2 * rows * taps * (c0 + c1) * ncols, moved=moved)

    def _conv_enc(self, label, layer, act, src, cin, ld, frames, kobs, h, w, out, ldo, algo, mean_out=None, ldm=0,
                  obs_weights=None):
        """One encoder conv (single source) over frames*kobs frames; with mean_out also the mean over the kobs
        observations (label + '.mean' when it needs its own launch).  Goes to the LDS-tiled kernel when the plan
        chose it for this launch, else to the register-tiled MFMA / direct kernels."""
        layer.build(cin, src.device)
        chint = self._trial_c32 or self.c32_hints.get(label, 0)
        if (chint and self.use_c32 and algo == C.ALGO_AUTO and obs_weights is None and layer.cin == cin and ld % 4 == 0 and ldo % 4 == 0
                and C.conv_c32_supported(layer.mode, cin, layer.n_ch_out) and not (chint == 2 and kobs == 1 and self._trial_c32)):
            unfold = chint == 2 and kobs > 1
            nf, c = frames * kobs, layer.n_ch_out
            fold_mean = mean_out is not None and not unfold
            nbytes = 4 * nf * h * w * (cin + c) + (4 * frames * h * w * c * (kobs + 1) if fold_mean else 0)
            self._ran_c32.add(label)
            self._launch(label, nbytes, C.conv_c32_forward, layer.mode, src, ld, cin, nf if unfold else frames, 1 if unfold else kobs, h, w,
                         layer.packed_tile(32), layer.bias.detach(), c, out, ldo, mean_out if fold_mean else None, ldm,
                         act=act is not None, alpha=act.alpha if act is not None else 0.0, flops=2 * nf * h * w * 4 * cin * c)
            if mean_out is not None and unfold:
                self._launch(label.replace('.s1', '.mean'), 4 * frames * h * w * c * (kobs + 1), C.obs_mean_forward,
                             out, None, frames, kobs, h * w, c, mean_out, ldm)
            return
        if (algo == C.ALGO_AUTO and layer.mode == C.CONV_K2S1 and (self._trial_wino or label in self.wino_hints)
                and self._wino(label, layer, act, src, cin, ld, frames, kobs, h, w, out, ldo, mean_out, ldm,
                               2 * frames * kobs * h * w * 4 * cin * layer.n_ch_out, obs_weights)):
            return
        hint = self._trial_lds or (0 if (self._trial_wino or self._trial_c32) else self.lds_hints.get(label, 0))
        tn, unfold = hint & 255, bool(hint >> 8)       # +256: observations as separate frames, mean in its own launch
        ok = (tn and obs_weights is None and algo == C.ALGO_AUTO and layer.mode in (C.CONV_K2S2, C.CONV_K2S1)
              and layer.cin == cin and cin % 16 == 0 and layer.n_ch_out % tn == 0)
        if ok and unfold and kobs == 1:
            ok = self._trial_lds == 0                   # nothing to unfold here: leave this launch to the other trials
            unfold = False
        oh, ow = layer.out_hw(h, w)
        if ok:
            nf = frames * kobs
            nbytes = 4 * (nf * h * w * cin + nf * oh * ow * layer.n_ch_out)
            fold_mean = mean_out is not None and not unfold
            if fold_mean:
                nbytes += 4 * frames * oh * ow * layer.n_ch_out * (kobs + 1)       # what the separate mean launch would move
            flops = 2 * nf * oh * ow * 4 * cin * layer.n_ch_out
            self._ran_lds.add(label)
            if self.precision in ('f32x3', 'f32x3_9'):
                # fp32 operands as three bf16 terms on the bf16 matrix cores (csrc/conv_tile3.hip): 6 or all 9 term products
                self._launch(label, nbytes, C.conv_tile3_forward, layer.mode, src, ld, cin, nf if unfold else frames,
                             1 if unfold else kobs, h, w, layer.packed_tile3(tn), layer.bias.detach(), layer.n_ch_out, tn, out, ldo,
                             mean_out if fold_mean else None, ldm, act=act is not None,
                             alpha=act.alpha if act is not None else 0.0, nprod=9 if self.precision == 'f32x3_9' else 6, flops=flops)
            else:
                self._launch(label, nbytes, C.conv_tile_forward, layer.mode, src, ld, cin, nf if unfold else frames,
                             1 if unfold else kobs, h, w, layer.packed_tile(tn), layer.bias.detach(), layer.n_ch_out, tn, out, ldo,
                             mean_out if fold_mean else None, ldm, act=act is not None,
                             alpha=act.alpha if act is not None else 0.0, flops=flops)
            if mean_out is not None and unfold:
                c = layer.n_ch_out
                self._launch(label.replace('.s1', '.mean'), 4 * frames * oh * ow * c * (kobs + 1), C.obs_mean_forward,
                             out, None, frames, kobs, oh * ow, c, mean_out, ldm)
            return
        self._conv(label, layer, act, src, cin, ld, None, 0, 0, frames * kobs, h, w, out, ldo, algo)
        if mean_out is not None:
            oh, ow = layer.out_hw(h, w)
            c = layer.n_ch_out
            self._launch(label.replace('.s1', '.mean'), 4 * frames * oh * ow * c * (kobs + 1), C.obs_mean_forward,
                         out, obs_weights, frames, kobs, oh * ow, c, mean_out, ldm)

    # ------------------------------------------------------------------ autotune
    def _autotune(self, run, backward=False):
        """Plan-creation-time choice of the MFMA wave tile (RT x CT) per launch: streaming layers
        want many small waves (memory-level parallelism), deep layers big register tiles (MFMA
        bound); a few 4/8-channel layers are faster on the direct kernel.  Times every candidate
        with HIP events on this shape and keeps the fastest."""
        saved = (self.timer, dict(self.tile_hints), dict(self.algo_hints))
        self._tuning = True                 # no launch tapes while trial plans run
        results = {}
        trials = [('tile', 16 * r + c) for r in (1, 2, 4) for c in (1, 2, 4)]
        if not self.fuse_ends and not backward:
            trials.append(('direct', 0))    # only the 4/8-channel full-resolution layers ever preferred it
        if not backward:
            trials += [('lds', 32), ('lds', 64), ('lds', 256 + 32), ('lds', 256 + 64)]
        elif self.tile_dgrad:
            trials += [('lds', 32), ('lds', 64)]                  # backward-data launches on the LDS-tiled kernel
        if self.use_c32 and not backward:
            trials += [('c32', 1), ('c32', 2)]                      # narrow stride-1 launches with LDS-resident weights
        if self.use_wino:                                           # stride-1 k2 launches on the Winograd kernel
            trials += [('wino', 32), ('wino', 64)] + ([('wino', 256 + 32), ('wino', 256 + 64)] if not backward else [])
        # split-K: launches with few GEMM rows and a long K (the deep levels; at depth 1024 a 1 x 1-texel level streams 33 MB of
        # weights through 4 rows) need thousands of waves each walking a short K slice to keep HBM busy: up to 128 slices
        mode = os.environ.get('NLT_SPLITK', 'all')                   # 'all' | 'fwd' (forward plans only) | 'off': A/B switch
        if mode == 'all' or (mode == 'fwd' and not backward):
            # (ks > 0: one launch -- slices meet in LDS, groups of slices through a ticket counter; ks < 0: every slice a wave of its
            # own and a second launch that adds them -- the faster form where 4-16 GEMM rows meet 32-128 slices, tools/bench_deep.py)
            forms = os.environ.get('NLT_SPLITK_FORMS', 'both')     # 'both' | 'one' (launch) | 'two' (launches): A/B switch
            cand = ((4, 8, 16, 32, 64, 128) if forms != 'two' else ()) + ((-16, -32, -64, -128) if forms == 'both' else ()) + \
                   ((-4, -8, -16, -32, -64, -128) if forms == 'two' else ())
            trials += [('splitk', (16 * r + c, ks)) for (r, c) in ((1, 1), (1, 2), (2, 2), (1, 4)) for ks in cand]
        saved_lds, saved_sk, saved_wino, saved_c32 = dict(self.lds_hints), dict(self.splitk_hints), dict(self.wino_hints), dict(self.c32_hints)
        for kind, hint in trials:
            self.tile_hints = {'*': hint} if kind == 'tile' else ({'*': hint[0]} if kind == 'splitk' else {})
            self.algo_hints = {}
            self.lds_hints, self.splitk_hints, self.wino_hints, self.c32_hints = {}, {}, {}, {}
            self._trial_direct = kind == 'direct'
            self._trial_lds = hint if kind == 'lds' else 0
            self._trial_wino = hint if kind == 'wino' else 0
            self._trial_c32 = hint if kind == 'c32' else 0
            self._trial_splitk = hint[1] if kind == 'splitk' else 0
            self._ran_direct, self._ran_lds, self._ran_splitk, self._ran_wino, self._ran_c32 = set(), set(), set(), set(), set()
            self.timer = None
            run()
            self.timer = OpTimer()
            run(); run()
            rec = self.timer.collect()
            for label, r in rec.items():
                t = r[1] / r[0]
                if label.endswith('.o.s1') and label.replace('.s1', '.mean') in rec:
                    m = rec[label.replace('.s1', '.mean')]
                    t += m[1] / m[0]        # the LDS kernel folds the mean in: compare like with like
                if (kind == 'tile' or label in self._ran_direct or label in self._ran_lds or label in self._ran_splitk
                        or label in self._ran_wino or label in self._ran_c32):
                    results.setdefault(label, []).append((t, kind, hint))
        self._trial_direct, self._trial_lds, self._trial_splitk, self._trial_wino, self._trial_c32 = False, 0, 0, 0, 0
        self.timer, self.tile_hints, self.algo_hints = saved
        self.lds_hints, self.splitk_hints, self.wino_hints, self.c32_hints = saved_lds, saved_sk, saved_wino, saved_c32
        for label, res in results.items():
            if '.s1' not in label and '.s2' not in label and label != 'L0.q':
                continue
            t, kind, hint = min(res)
            if backward and 'dgrad' not in label:
                continue                                            # (a backward trial pass only chooses backward-data launches)
            if kind == 'direct':
                self.algo_hints.setdefault(label, C.ALGO_DIRECT)
            elif kind == 'lds':
                if label not in self.wino_hints and label not in self.c32_hints:
                    self.lds_hints.setdefault(label, hint)
            elif kind == 'wino':
                if label not in self.lds_hints and label not in self.c32_hints:
                    self.wino_hints.setdefault(label, hint)
            elif kind == 'c32':
                if label not in self.lds_hints and label not in self.wino_hints:
                    self.c32_hints.setdefault(label, hint)
            elif kind == 'splitk':
                if label not in self.tile_hints and label not in self.splitk_hints:
                    self.tile_hints[label], self.splitk_hints[label] = hint
            else:
                self.tile_hints.setdefault(label, hint)
        self.tuned = {**getattr(self, 'tuned', {}), **results}
        self._tuning = False
        self._drop_tapes()
        reg = getattr(self.q.layers[0], '_registry', None)
        if reg is not None and self.prune_packs:
            reg.begin_census()              # which of the candidates' fragment buffers does the chosen plan read?

    def _drop_tapes(self):
        for b in self._bufs.values():
            b.pop('tapes', None)

    def export_tuning(self):
        """The plan-time choices (wave tiles, direct / LDS-tiled kernel, split-K slices per launch label) as one dict."""
        return {'tile_hints': dict(self.tile_hints), 'algo_hints': dict(self.algo_hints), 'lds_hints': dict(self.lds_hints),
                'splitk_hints': dict(self.splitk_hints), 'wino_hints': dict(self.wino_hints), 'c32_hints': dict(self.c32_hints)}

    def import_tuning(self, d):
        """Takes another plan's (or an earlier run's) choices and skips the plan-time trials: two plans with the same
        choices issue the same kernels with the same summation orders."""
        self.tile_hints.update(d['tile_hints']); self.algo_hints.update(d['algo_hints'])
        self.lds_hints.update(d.get('lds_hints', {}))
        self.splitk_hints.update(d.get('splitk_hints', {}))
        self.wino_hints.update(d.get('wino_hints', {}))
        self.c32_hints.update(d.get('c32_hints', {}))
        self.autotune = False
        self._drop_tapes()

    def save_tuning(self, path):
        import json
        with open(path, 'w') as f:
            json.dump(self.export_tuning(), f)

    def load_tuning(self, path):
        """Re-uses tile choices measured by an earlier run (skips the plan-time trials)."""
        import json
        with open(path) as f:
            self.import_tuning(json.load(f))

    # ------------------------------------------------------------------ forward
    def can_fuse(self, b, obs_weights, obs_override):
        """The fused ends cover the released first/last levels (depth0 = 16 fixes them: L0 -> 16, first
        contracting block 16, last expanding block 8 -> 4, head 36 -> 3) with plain observation means."""
        q, D, U, cl = self.q, self.n_down, self.n_up, b['C']
        if not (self.fuse_ends and self.use_obs and obs_weights is None and obs_override is None and D >= 2 and U >= 2):
            return False
        if b['obs'][0].shape[1] > 14 and not self.front_v4:     # the first-generation front kernel keeps (1 + k) haloed tiles in LDS
            return False
        last = q.layers[D + U].convs()
        prev = q.layers[D + U - 1].convs()
        acts = [a for blk in (q.layers[1], self.o.layers[1], q.layers[D + U]) for _, a in blk.convs()]
        return (cl[0] == 16 and cl[1] == 16 and last[0][0].n_ch_out == 4 and last[1][0].n_ch_out == 4
                and prev[1][0].n_ch_out == 8 and q.layers[-1].n_ch_out == 3
                and all(a is not None for a in acts) and len({a.alpha for a in acts}) == 1)

    def _front_weights(self, dev, l2=True):
        """Folded + fragment-packed weights of the front kernel, re-derived when any source kernel changed.  l2 = False
        (training: the level-2 fold is an inference-only part of the kernel) leaves the level-2 blob alone."""
        q, o, D, U = self.q, self.o, self.n_down, self.n_up
        q0, o0, head = q.layers[0], o.layers[0], q.layers[-1]
        (qa, _), (qb, _) = q.layers[1].convs()
        (oa, _), (ob, _) = o.layers[1].convs()
        q0.build(5, dev); o0.build(3, dev); qa.build(32, dev); qb.build(16, dev); oa.build(16, dev); ob.build(16, dev)
        head.build(36, dev)
        stamp = lambda cs: tuple((c.kernel.data_ptr(), c.kernel._version, c.bias._version, c._epoch[0]) for c in cs)
        w = lambda c: (c.kernel.detach(), c.bias.detach())
        if self._front_blob is None:
            self._front_blob = [None, None, None, None]              # [stamp, blob, level-2 blob, its stamp]
        fb = self._front_blob                                        # refilled in place: launch tapes / graphs keep the addresses
        ver = stamp((q0, o0, qa, qb, oa, ob, head))
        if fb[0] != ver:
            fb[1] = C.front_pack_weights(*w(q0), *w(o0), *w(qa), *w(qb), *w(oa), *w(ob), *w(head), out=fb[1])
            fb[0] = ver
        if l2:
            (qa2, _), _ = q.layers[2].convs()
            (oa2, _), _ = o.layers[2].convs()
            qa2.build(32, dev); oa2.build(16, dev)
            ver2 = stamp((qa2, oa2))
            if fb[3] != ver2:
                ok = qa2.n_ch_out == 32 and oa2.n_ch_out == 32 and qa2.cin == 32 and oa2.cin == 16
                fb[2] = C.front_pack_l2_weights(*w(qa2), *w(oa2), out=fb[2]) if ok else None
                fb[3] = ver2
        return fb[1], fb[2]

    def resident_ok(self, n, k, h, w, alpha=0.3):
        """Can `forward(resident=...)` read the uint8 capture store directly (csrc/front4.hip, uint8 variant)?"""
        return (self.fuse_ends and self.front_v4 and self.front_l2 and self.use_obs and h % 4 == 0 and w % 8 == 0
                and 0.0 <= alpha <= 1.0)

    def forward(self, base, cvis, lvis, nn_rgb, nn_base, obs_weights=None, obs_override=None,
                skip_connect_base=True, algo=C.ALGO_AUTO, inference=False, resident=None, pred_out=None):
        """base [N,H,W,3], cvis/lvis [N,H,W,1], nn_rgb/nn_base [N,k,H,W,3] -> pred [N,H,W,3]
        (texel (0,0) zeroed, base added).  Returns (pred, buffers).
        inference=True lets the plan use the fused ends (csrc/fused.hip), which do not keep the
        activations a backward pass would need (fm0, obs0, the L1 / last-block intermediates).
        resident = ResidentTexels (datasets/nlt.py): the five float buffers are None and the front kernel reads the
        uint8 capture store itself (inference only; `resident_ok` says when -- `resident_override_ok` with an obs_override).
        pred_out [N,H,W,3] (inference, fused ends): the last launch writes the rendered texels THERE instead of the plan's
        reusable buffer, so the caller can hand them out without a copy (50 MB and 19 us per step at 4 x 1024^2).  That one
        launch stays out of the launch tape -- its output address changes every step -- and is re-issued after a replay.
        Returns (pred, buffers); pred is pred_out when it was used."""
        C.set_workspace_scope(id(self))
        self._pred_out = pred_out
        self._pred_slot = 'back_infer' if inference else 'back_train'
        if resident is not None:
            if obs_override is not None:
                return self._forward_resident_ovr(resident, obs_override, skip_connect_base, algo)
            return self._forward_resident(resident, skip_connect_base, algo)
        n, h, w, _ = base.shape
        k = nn_rgb.shape[1]
        dev = base.device
        # the reference's inference mode (one given map per level for every frame): the fused query-only plan, no observation buffers
        use_ovr = (obs_override is not None and inference and obs_weights is None
                   and self.can_fuse_override(self._level_channels(), obs_override, h, w, (base, cvis, lvis)))
        b = self._buffers(n, 0 if use_ovr else k, h, w, dev)
        if not self._tuning:
            self.generation += 1
        reg = getattr(self.q.layers[0], '_registry', None)
        if reg is not None:
            if not self._tuning:
                reg.tick()
            reg.refresh_if_stale()          # all packed fragments, one launch, before any stream is forked
        ovr = self._prepare_override(b, obs_override, dev) if use_ovr else None
        fused = not use_ovr and (inference or self.fuse_train) and self.can_fuse(b, obs_weights, obs_override)
        if fused and not inference and w % 8:                           # the training ends: w/2 in groups of 4 texels
            fused = False
        b['train_fused'] = fused and not inference
        tuned_key = 'tuned_ovr' if use_ovr else (('tuned_fused' if inference else 'tuned_train') if fused else 'tuned')
        if self.autotune and not b.get(tuned_key) and base.is_cuda:
            b[tuned_key] = True
            self._autotune(lambda: self.forward(base, cvis, lvis, nn_rgb, nn_base, obs_weights, obs_override,
                                                skip_connect_base, algo, inference, pred_out=pred_out))
        # launch tape (second sight of the same inputs records, later sights replay)
        if fused or use_ovr:
            self._front_weights(dev, l2=inference or self.front4_train)   # folded front-kernel weights, refreshed in place OUTSIDE any tape
        tkey = None
        if (self.use_tape and base.is_cuda and self.timer is None and not self._tuning and reg is not None
                and obs_weights is None and (obs_override is None or use_ovr)
                and all(t.is_contiguous() for t in (base, cvis, lvis, nn_rgb, nn_base))):    # (a replay skips the adapters' layout checks)
            tkey = ('fwd', base.data_ptr(), cvis.data_ptr(), lvis.data_ptr(), nn_rgb.data_ptr(), nn_base.data_ptr(),
                    bool(skip_connect_base), algo, inference, fused, C._stream(), pred_out is not None,
                    ovr['serial'] if use_ovr else 0)
            return self._run_taped(b, reg, tkey, lambda: self._forward_body(
                b, base, cvis, lvis, nn_rgb, nn_base, obs_weights, obs_override, skip_connect_base, algo, fused, inference, ovr))
        return self._forward_body(b, base, cvis, lvis, nn_rgb, nn_base, obs_weights, obs_override, skip_connect_base, algo,
                                  fused, inference, ovr)

    def _run_taped(self, b, reg, tkey, body):
        """`body()` under the launch tape of this (inputs, plan state) key: first sight runs it, second sight records it, later
        sights replay the record (csrc/tape.hip) -- as long as the buffers and packed fragments it baked in are still the ones."""
        tapes = b.setdefault('tapes', {})
        if len(tapes) > 16:                     # ever-changing input addresses (a loader that allocates per step): forget
            tapes.clear()
        ent = tapes.get(tkey, 0)
        if isinstance(ent, tuple):
            if self._replayable(b, ent, reg):
                reg.touch_keys(ent[4])              # (a replay reads its fragment buffers without asking: tell a running census)
                C.replay(ent)
                self.tape_replays += 1
                return self._finish_pred(b), b
            ent = 1
        tapes[tkey] = 1
        if ent == 1:
            C.tape_begin()
            reg.begin_record()
            try:
                out = body()
            except BaseException:
                C.tape_abort()
                reg.end_record()
                raise
            tapes[tkey] = C.tape_end(reg.version, reg.end_record()) or 1  # (None: a workspace grew while recording -> record again)
            return out
        return body()

    def _forward_resident(self, res, skip_connect_base, algo):
        """Inference forward whose inputs are still in the resident uint8 store: same plan, the front launch is
        nlt_front4_forward_u8 (frame ids in, `_load_data`'s conversion in registers)."""
        n, k, h, w = res.n, res.k, res.h, res.w
        dev = res.cvis.device
        b = self._buffers(n, k, h, w, dev)
        if not self._tuning:
            self.generation += 1
        reg = getattr(self.q.layers[0], '_registry', None)
        if reg is not None:
            if not self._tuning:
                reg.tick()
            reg.refresh_if_stale()
        if not self.can_fuse(b, None, None):
            raise C.NLTError("this network / plan cannot take store-resident inputs: materialise the batch")
        b['train_fused'] = False
        if self.autotune and not b.get('tuned_fused'):
            b['tuned_fused'] = True
            self._autotune(lambda: self._forward_resident(res, skip_connect_base, algo))
        self._front_weights(dev)
        tkey = None
        if self.use_tape and self.timer is None and not self._tuning and reg is not None:
            tkey = ('fwd_u8',) + res.key() + (bool(skip_connect_base), algo, C._stream(), self._pred_out is not None)
            return self._run_taped(b, reg, tkey, lambda: self._forward_fused(b, None, None, None, None, None, skip_connect_base, algo,
                                                                             resident=res))
        return self._forward_fused(b, None, None, None, None, None, skip_connect_base, algo, resident=res)

    def _forward_body(self, b, base, cvis, lvis, nn_rgb, nn_base, obs_weights, obs_override, skip_connect_base, algo, fused,
                      inference, ovr=None):
        if ovr is not None:
            return self._forward_ovr(b, base, cvis, lvis, ovr, skip_connect_base, algo)
        n, h, w, _ = base.shape
        k = nn_rgb.shape[1]
        dev = base.device
        q, o, D, cl = self.q, self.o, self.n_down, b['C']
        mult = 2 if self.use_obs else 1
        run_obs = self.use_obs and obs_override is None
        if self.precision == 'bf16' and not (fused and inference):
            raise C.NLTError("precision = bf16 is an inference mode of the fused plan (training, obs_override / obs_weights and the "
                             "layer-by-layer plan run fp32)")
        if fused:
            return self._forward_fused(b, base, cvis, lvis, nn_rgb, nn_base, skip_connect_base, algo, train=not inference)

        # L0 (both paths) + first observation mean
        q0, o0 = q.layers[0], o.layers[0]
        q0.build(5, dev); o0.build(3, dev)
        if run_obs:
            nbytes = 4 * n * h * w * (5 + 6 * k + 2 * cl[0] + k * cl[0])
            self._launch('L0.stem', nbytes, C.stem_forward, base, cvis, lvis, nn_rgb, nn_base, obs_weights,
                         n, k, h, w, cl[0], q0.kernel.detach(), q0.bias.detach(), o0.kernel.detach(),
                         o0.bias.detach(), b['fm'][0], b['obs'][0])
        else:
            # no observation path to run (use_obs = False, or its features are given): query L0 alone
            x5 = b['x5'] = torch.cat((base, cvis, lvis), 3)              # (kept: the L0 weight gradient reads it)
            self._conv('L0.q', q0, None, x5, 5, 5, None, 0, 0, n, h, w, b['fm'][0], mult * cl[0], algo)
            if self.use_obs:
                b['fm'][0][..., cl[0]:].copy_(obs_override[0].expand(n, -1, -1, -1))

        hh, ww = h, w
        for l in range(1, D + 1):
            (qa, qact_a), (qb, qact_b) = q.layers[l].convs()
            cin = mult * cl[l - 1]
            self._conv_enc('L%d.q.s2' % l, qa, qact_a, b['fm'][l - 1], cin, cin, n, 1, hh, ww, b['qtmp'][l], cl[l], algo)
            if run_obs:
                (oa, oact_a), (ob, oact_b) = o.layers[l].convs()
                self._conv_enc('L%d.o.s2' % l, oa, oact_a, b['obs'][l - 1], cl[l - 1], cl[l - 1], n, k, hh, ww,
                               b['otmp'][l], cl[l], algo)
            hh, ww = hh // 2, ww // 2
            self._conv_enc('L%d.q.s1' % l, qb, qact_b, b['qtmp'][l], cl[l], cl[l], n, 1, hh, ww, b['fm'][l], mult * cl[l], algo)
            if run_obs:
                self._conv_enc('L%d.o.s1' % l, ob, oact_b, b['otmp'][l], cl[l], cl[l], n, k, hh, ww, b['obs'][l], cl[l],
                               algo, mean_out=b['fm'][l].view(-1)[cl[l]:], ldm=2 * cl[l], obs_weights=obs_weights)
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

    def _forward_fused(self, b, base, cvis, lvis, nn_rgb, nn_base, skip_connect_base, algo, train=False, resident=None):
        """front kernel (layers 0-1) -> unfused levels 2..D and decoder blocks -> back kernel.
        train=True keeps the activations the backward pass reads (qtmp[1], otmp[1], obs[1], the last block's two
        4-channel maps) and leaves level 2's stride-2 convs to their own launches (their inputs must be stored anyway)."""
        if resident is not None:
            n, k, h, w = resident.n, resident.k, resident.h, resident.w
            dev = resident.cvis.device
        else:
            n, h, w, _ = base.shape
            k = nn_rgb.shape[1]
            dev = base.device
        q, o, D, U, cl = self.q, self.o, self.n_down, self.n_up, b['C']
        alpha = q.layers[1].convs()[0][1].alpha
        if b['skip3'] is None:
            b['skip3'] = torch.empty((n, h, w, 3), device=dev, dtype=torch.float32)
        blob, blob_l2 = self._front_weights(dev, l2=(not train) or self.front4_train)
        # With k <= 4 the front kernel also runs level 2's stride-2 convs (its 8 x 16 level-1 tile is a 4 x 8 tile of
        # level 2): the per-observation level-1 maps never reach HBM and L2.{q,o}.s2 are not launched.
        v4 = self.front_v4 and 0.0 <= alpha <= 1.0 and (resident is not None or C.front4_supported(base, cvis, lvis, nn_rgb, nn_base))
        # (training: only the second-generation kernel has a form that also keeps the level-1 maps the backward reads)
        front2 = (self.front_l2 and blob_l2 is not None and (k <= 4 or v4) and h % 4 == 0 and w % 4 == 0 and not self._trial_direct
                  and (not train or (v4 and self.front4_train)))
        if resident is not None and not (front2 and v4 and w % 8 == 0):
            raise C.NLTError("store-resident inputs need the fused front kernel (front4, level-2 fold, w % 8 == 0)")
        nbytes = 4 * n * h * w * ((5 + 3 * k + 16 + 16 * k) + (36 + 20 * k + 8 + 8 * k))     # L0 + L1 (+ means)
        flops = 2 * n * (h // 2) * (w // 2) * ((32 + 64) * 16 + k * (12 + 64) * 16) + 2 * n * h * w * 24
        if front2:
            nbytes += 4 * n * (h // 2) * (w // 2) * (32 + 16 * k) + 4 * n * (h // 4) * (w // 4) * 32 * (1 + k)   # + the two L2 s2 launches
            flops += 2 * n * (h // 4) * (w // 4) * 32 * (128 + 64 * k)
            moved = 4 * n * h * w * (5 + 6 * k + 3 + 8) + 4 * n * (h // 4) * (w // 4) * 32 * (1 + k)   # in + skip3 + fm1 | qtmp2 + otmp2
            if resident is not None:
                moved = n * h * w * (5 + 6 * k) + 4 * n * h * w * (3 + 8) + 4 * n * (h // 4) * (w // 4) * 32 * (1 + k)   # uint8 in
                self._launch('F.front', nbytes, C.front4_forward_u8, resident.diffuse, resident.rgb, resident.cvis, resident.lvis,
                             resident.ids, resident.nn_ids, n, k, h, w, blob, blob_l2, skip_connect_base, alpha, b['fm'][1],
                             b['skip3'], b['qtmp'][2], b['otmp'][2], flops=flops, moved=moved)
            elif train:
                moved += 4 * n * (h // 2) * (w // 2) * 16 * (1 + 2 * k)      # + the three level-1 maps kept for the backward
                self._launch('F.front', nbytes, C.front4_forward_train, base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, blob, blob_l2,
                             skip_connect_base, alpha, b['fm'][1], b['skip3'], b['qtmp'][2], b['otmp'][2], b['obs'][1], b['qtmp'][1],
                             b['otmp'][1], flops=flops, moved=moved)
            elif v4:
                self._launch('F.front', nbytes, C.front4_forward, base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, blob, blob_l2,
                             skip_connect_base, alpha, b['fm'][1], b['skip3'], b['qtmp'][2], b['otmp'][2],
                             flops=flops, moved=moved)
            else:
                self._launch('F.front', nbytes, C.front2_forward, base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, blob, blob_l2,
                             skip_connect_base, alpha, b['fm'][1], b['skip3'], b['qtmp'][2], b['otmp'][2], flops=flops, moved=moved)
        else:
            # what the fused launch itself must move: raw inputs + skip3 out, fm1 + obs1 out (per texel: 5+6k+3 | (32+16k)/4)
            moved = 4 * n * h * w * (5 + 6 * k + 3 + 8 + 4 * k)
            if train:
                moved += 4 * n * h * w * (4 + 4 * k)
                self._launch('F.front', nbytes, C.front_forward_train, base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, blob,
                             skip_connect_base, alpha, b['fm'][1], b['obs'][1], b['skip3'], b['qtmp'][1], b['otmp'][1],
                             flops=flops, moved=moved)
            else:
                self._launch('F.front', nbytes, C.front_forward, base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, blob,
                             skip_connect_base, alpha, b['fm'][1], b['obs'][1], b['skip3'], flops=flops, moved=moved)
        # Levels 2..D.  The observation chain (k frames per frame: three quarters of the encoder's work at k = 4)
        # never waits for the query path; the query convs of a level only need the previous level's observation mean.
        # With two HIP streams the small deep-level launches of one path fill the CUs the other leaves idle.
        concurrent = (self.two_streams and dev.type == 'cuda' and (self.timer is None or getattr(self.timer, 'only', None) is not None)
                      and not self._trial_lds and not self._trial_splitk and not self._trial_direct and not self._trial_wino
                      and not self._trial_c32)
        if concurrent:
            if self._side is None:
                self._side = (torch.cuda.Stream(device=dev), [C.new_event() for _ in range(D + 3)])
            side, ev = self._side
            main = torch.cuda.current_stream()
            # (An event record is a marker packet on the recording stream: the launch behind it starts 4-12 us late -- 46 us of
            # such gaps per pass.  Forking the query stream later, behind level 2's observation conv, removed one gap and
            # delayed the query path by as much: measured r04, no gain, removed in r06; profiles/README.md.)
            C.record_event(ev[0], main)                             # front kernel done: fm[1], obs[1]
            C.wait_event(side, ev[0])
        hh, ww = h // 2, w // 2
        bf = self.precision == 'bf16' and not train
        alias_obs = k == 1 and not train and not bf and self.alias_obs
        for l in range(2, D + 1):
            (qa, qact_a), (qb, qact_b) = q.layers[l].convs()
            (oa, oact_a), (ob, oact_b) = o.layers[l].convs()
            cin = 2 * cl[l - 1]
            s2_done = front2 and l == 2                             # the front kernel already wrote qtmp[2] / otmp[2]
            c = cl[l]
            h2_, w2_ = hh // 2, ww // 2

            def obs_path():
                if bf and l >= 3:                                   # bf16 region: stored-bf16 maps, mean in its own launch
                    self._conv_bf('L%d.o.s2' % l, oa, oact_a, b['obs'][l - 1], cl[l - 1], cl[l - 1], None, 0, 0, n * k, hh, ww,
                                  b['otmp'][l], c)
                    self._conv_bf('L%d.o.s1' % l, ob, oact_b, b['otmp'][l], c, c, None, 0, 0, n * k, h2_, w2_, b['obs'][l], c)
                    self._launch('L%d.o.mean' % l, 4 * n * h2_ * w2_ * c * (k + 1), C.obs_mean_bf16, b['obs'][l], n, k, h2_ * w2_, c,
                                 b['fm'][l].view(-1)[c:], 2 * c, moved=2 * n * h2_ * w2_ * c * (k + 1))
                    return
                if alias_obs:
                    # ONE observation per frame, inference: the level's observation features ARE its "mean" -- the stride-1 conv
                    # writes them straight into the second half of the interleaved map and the next level reads them there
                    # (no obs[l] buffer traffic, no mean launch: one launch per level off the chain of the released shapes)
                    if not s2_done:
                        if l > 2:
                            self._conv_enc('L%d.o.s2' % l, oa, oact_a, b['fm'][l - 1].view(-1)[cl[l - 1]:], cl[l - 1], 2 * cl[l - 1],
                                           n, 1, hh, ww, b['otmp'][l], c, algo)
                        else:
                            self._conv_enc('L%d.o.s2' % l, oa, oact_a, b['obs'][1], cl[1], cl[1], n, 1, hh, ww, b['otmp'][l], c, algo)
                    self._conv_enc('L%d.o.s1' % l, ob, oact_b, b['otmp'][l], c, c, n, 1, h2_, w2_, b['fm'][l].view(-1)[c:], 2 * c, algo)
                    return
                if not s2_done:
                    self._conv_enc('L%d.o.s2' % l, oa, oact_a, b['obs'][l - 1], cl[l - 1], cl[l - 1], n, k, hh, ww, b['otmp'][l], c, algo)
                self._conv_enc('L%d.o.s1' % l, ob, oact_b, b['otmp'][l], c, c, n, k, h2_, w2_, b['obs'][l], c, algo,
                               mean_out=b['fm'][l].view(-1)[c:], ldm=2 * c)

            def query_path():
                if bf and l >= 3:
                    self._conv_bf('L%d.q.s2' % l, qa, qact_a, b['fm'][l - 1], cin, cin, None, 0, 0, n, hh, ww, b['qtmp'][l], c)
                    self._conv_bf('L%d.q.s1' % l, qb, qact_b, b['qtmp'][l], c, c, None, 0, 0, n, h2_, w2_, b['fm'][l], 2 * c)
                    return
                if not s2_done:
                    self._conv_enc('L%d.q.s2' % l, qa, qact_a, b['fm'][l - 1], cin, cin, n, 1, hh, ww, b['qtmp'][l], c, algo)
                self._conv_enc('L%d.q.s1' % l, qb, qact_b, b['qtmp'][l], c, c, n, 1, h2_, w2_, b['fm'][l], 2 * c, algo)

            obs_path()
            if concurrent:
                C.record_event(ev[l], main)                         # fm[l]'s observation half is complete
                with torch.cuda.stream(side):
                    if l > 2:
                        C.wait_event(side, ev[l - 1])
                    query_path()
            else:
                query_path()
            hh, ww = hh // 2, ww // 2
        if concurrent:
            C.record_event(ev[D + 1], side)
            C.wait_event(main, ev[D + 1])                              # the decoder needs both halves of every fm[l]
        x, cx = b['fm'][D], 2 * cl[D]
        for j in range(U - 1):
            (da, dact_a), (db, dact_b) = q.layers[D + 1 + j].convs()
            skip, cs = b['fm'][D - j], 2 * cl[D - j]
            lab = 'L%d.q' % (D + 1 + j)
            nl = da.n_ch_out
            if bf and j <= D - 3:                                   # bf16 region (the block with 16 outputs hands fp32 on)
                self._conv_bf(lab + '.s2', da, dact_a, x, cx, cx, skip, cs, cs, n, hh, ww, b['dtmp'][j], nl)
                hh, ww = hh * 2, ww * 2
                self._conv_bf(lab + '.s1', db, dact_b, b['dtmp'][j], nl, nl, None, 0, 0, n, hh, ww, b['dec'][j], db.n_ch_out)
                x, cx = b['dec'][j], db.n_ch_out
                continue
            if (self.fuse_dec and not train and nl in (8, 16) and db.n_ch_out == nl and cx % 4 == 0 and algo == C.ALGO_AUTO
                    and dact_a is not None and dact_b is not None and dact_a.alpha == dact_b.alpha and not self._trial_direct):
                da.build(cx + cs, dev); db.build(nl, dev)
                nbytes = 4 * n * hh * ww * ((cx + cs) + 4 * nl) + 4 * n * 4 * hh * ww * 2 * nl      # both convs
                self._launch(lab, nbytes, C.dec_block_forward, x, cx, skip, cs, n, hh, ww, da.kernel.detach(), da.bias.detach(),
                             db.kernel.detach(), db.bias.detach(), nl, dact_a.alpha, b['dec'][j],
                             flops=2 * n * hh * ww * (cx + cs) * 4 * nl + 2 * n * 4 * hh * ww * 4 * nl * nl,
                             moved=4 * n * hh * ww * ((cx + cs) + 4 * nl))
                hh, ww = hh * 2, ww * 2
                x, cx = b['dec'][j], nl
                continue
            self._conv(lab + '.s2', da, dact_a, x, cx, cx, skip, cs, cs, n, hh, ww, b['dtmp'][j], da.n_ch_out, algo)
            hh, ww = hh * 2, ww * 2
            self._conv(lab + '.s1', db, dact_b, b['dtmp'][j], da.n_ch_out, da.n_ch_out, None, 0, 0, n, hh, ww,
                       b['dec'][j], db.n_ch_out, algo)
            x, cx = b['dec'][j], db.n_ch_out
        (da, _), (db, _) = q.layers[D + U].convs()
        head = q.layers[-1]
        da.build(cx + 2 * cl[1], dev); db.build(4, dev)
        assert (hh, ww) == (h // 2, w // 2) and cx == 8 and da.cin == 40
        # last block (40 -> 4 -> 4 at full resolution) + head (36 -> 3) in SURVEY 8d accounting
        nbytes = 4 * n * h * w * ((10 + 4) + (4 + 4) + (36 + 3))
        extra = (b['dtmp'][U - 1], b['dec'][U - 1]) if train else ()
        back_args = (x, b['fm'][1], b['skip3'], n, hh, ww, da.kernel.detach(), da.bias.detach(), db.kernel.detach(), db.bias.detach(),
                     head.kernel.detach(), alpha)
        back_kw = dict(flops=2 * n * hh * ww * 40 * 16 + 2 * n * h * w * (64 + 12), moved=4 * n * h * w * (10 + 3 + 3 + (8 if train else 0)))

        def back(pred):
            self._launch('F.back', nbytes, C.back_forward_train if train else C.back_forward, *back_args, pred, *extra, **back_kw)
        out = self._pred_out if not self._tuning else None
        if out is None:
            # (leaves the closures alone: a timed survey or a copy-out pass over these buffers must not take them away from
            # the tapes recorded with pred_out -- they would hand out a stale b['pred'])
            back(b['pred'])
            return b['pred'], b
        # the caller's own output tensor: this launch is not part of the launch tape (its output address differs every step).
        # One closure per kind of pass: the train forward's last launch also keeps two maps for the backward.
        b['back_train' if train else 'back_infer'] = back
        paused = C.tape_pause()
        try:
            back(out)
        finally:
            C.tape_resume(paused)
        return out, b

    def _finish_pred(self, b):
        """After a tape replay: the launch that was kept out of the tape (see `forward`, pred_out)."""
        if self._pred_out is None:
            return b['pred']
        b[self._pred_slot](self._pred_out)      # (`_replayable` made sure it exists)
        return self._pred_out

    def _replayable(self, b, ent, reg):
        """A recorded forward tape may be replayed: still valid, and -- when this call brought its own output tensor -- the
        launch that was kept out of it is at hand (written only by a pred_out pass over these buffers)."""
        return C.tape_valid(ent, reg.version) and (self._pred_out is None or b.get(self._pred_slot) is not None)

    # ------------------------------------------------------------------ backward
    def _grad_buffers(self, b):
        g = b.get('grads')
        if g is None:
            Z = torch.empty_like
            g = {'fm': [Z(t) for t in b['fm']], 'obs': [Z(t) for t in b['obs']],
                 'qtmp': [None] + [Z(t) for t in b['qtmp'][1:]], 'otmp': [None] + [Z(t) for t in b['otmp'][1:]],
                 'dtmp': [Z(t) for t in b['dtmp']], 'dec': [Z(t) for t in b['dec']],
                 'zero_bias': torch.zeros(4096, device=b['pred'].device)}
            b['grads'] = g
        return g

    def _wgrad(self, label, layer, src0, c0, ld0, src1, c1, ld1, n, h, w, dpre, ldp):
        """Weight + bias gradient of one conv.  Nothing downstream of it in the backward pass reads the result, so with
        two streams it is queued on the side stream (after an event on the gradient it consumes) and the backward-data
        chain -- the critical path -- carries on; `backward` joins the streams at the end."""
        bs = self._bside
        if bs is not None and bs[2] is not None:
            # (An event per launch.  Handing the weight gradients over in batches of 3 / 6 behind one event was measured SLOWER in
            # r04 -- 3.22-3.24 -> 3.28 / 3.33 ms l2: their later start costs more than the ~5 us marker gaps on the chain --
            # and removed in r06.)
            side, events, cur = bs[:3]
            if cur[0] == len(events):
                events.append(C.new_event())
            ev = events[cur[0]]
            if len(bs) > 3 and bs[3] is not None and cur[0] % 2:            # two weight-gradient streams, dealt alternately
                side = bs[3]                                                # (scratch is per stream: _capi's wgrad workspaces)
            cur[0] += 1
            C.record_event(ev, torch.cuda.current_stream())
            with torch.cuda.stream(side):
                C.wait_event(side, ev)
                self._wgrad_now(label, layer, src0, c0, ld0, src1, c1, ld1, n, h, w, dpre, ldp)
            return
        self._wgrad_now(label, layer, src0, c0, ld0, src1, c1, ld1, n, h, w, dpre, ldp)

    def _grad_range_done(self, i):
        """Range i of the flat gradient bucket (models/nlt.py:_flatten: 0 = expanding blocks, 1 = encoder levels D .. grad_mid_level)
        has all its weight-gradient launches queued: fire the hook on their stream."""
        bs = self._bside
        if bs is not None and bs[2] is not None and bs[3] is not None:
            side, events, cur, side2 = bs                            # (two weight-gradient streams: the hook's stream waits for the other)
            if cur[0] == len(events):
                events.append(C.new_event())
            C.record_event(events[cur[0]], side2)
            C.wait_event(side, events[cur[0]])
            cur[0] += 1
        C.tape_call(self._fire_grad_hook, bs[0] if (bs is not None and bs[2] is not None) else None, i)

    def _fire_grad_hook(self, side, i=0):
        """Runs `grad_hook(i)` on the stream the expanding blocks' weight gradients were queued on.  `side` is an ARGUMENT
        of the recorded call (not looked up in `_bside`, whose cursor only exists while a plan is being issued): a
        launch-tape replay re-invokes this with the same stream the recorded wgrad launches keep, so the collective the
        hook starts is ordered after them -- on the main stream it would race the side stream's accumulation.
        Plan-time trial passes (`_tuning`) never fire it: their gradients are garbage and their count differs per rank."""
        hook = self.grad_hook
        if hook is None or self._tuning:
            return
        if side is not None:
            with torch.cuda.stream(side):
                hook(i)
        else:
            hook(i)

    def _wgrad_now(self, label, layer, src0, c0, ld0, src1, c1, ld1, n, h, w, dpre, ldp):
        oh, ow = layer.out_hw(h, w)
        nbytes = 4 * (n * h * w * (c0 + c1) + n * oh * ow * layer.n_ch_out)
        gw = w // 2 if layer.mode == C.CONV_K2S2 else w
        tiled = (self.wgrad_tiled and c0 % 4 == 0 and c1 % 4 == 0 and layer.n_ch_out % 4 == 0 and ld0 % 4 == 0
                 and (c1 == 0 or ld1 % 4 == 0) and ldp % 4 == 0 and gw >= 4)
        fn = C.conv_backward_weights_tiled if tiled else C.conv_backward_weights
        ncols = layer.n_ch_out * (4 if layer.mode == C.DECONV_K2S2 else 1)
        kdim = (c0 + c1) * (1 if layer.mode in (C.DECONV_K2S2, C.CONV1X1) else 4)
        if self.wgrad_narrow and ncols <= 32 and kdim <= 128 and layer.mode != C.CONV1X1 and gw >= 4:
            fn = C.conv_backward_weights_narrow     # few output columns: MFMA tile matched to the layer (csrc/wgrad_narrow.hip)
        self._launch(label, nbytes, fn, layer.mode, src0, c0, ld0, src1, c1, ld1, n, h, w,
                     dpre, ldp, layer.n_ch_out, layer.dkernel, layer.dbias)

    def _dgrad(self, label, layer, lo, hi, dpre, ldp, n, oh, ow, out, ldo, mask_src=None, ldm=0, mask_alpha=0.3,
               accumulate=False, zero_bias=None, split=None):
        """Backward-data of `layer` w.r.t. its input channels [lo,hi): the adjoint conv family on
        the gradient w.r.t. the layer's pre-activation output dpre [n,oh,ow,cout].  mask_src (the
        saved activation the result corresponds to) turns the result into the gradient w.r.t. the
        PRODUCER's pre-activation.  split = (c, obs_y, dobs, alpha_o, has_partial): the target is dfm[l] of a level with one
        observation per frame -- its observation half leaves the launch as the finished gradient of the observation path's
        pre-activation (nlt_conv_backward_data), so the level needs no `level_split` pass."""
        packed, ks = layer.packed_adjoint(lo, hi)
        adj = layer.ADJOINT[layer.mode]
        out_px = n * oh * ow * (4 if adj == C.DECONV_K2S2 else 1) // (4 if adj == C.CONV_K2S2 else 1)
        nbytes = 4 * (n * oh * ow * layer.n_ch_out + out_px * (hi - lo))
        # wave tile / split-K of this launch: chosen by timing at plan time like the forward's (`_autotune` on the backward)
        ncols = (hi - lo) * (4 if adj == C.DECONV_K2S2 else 1)
        rows = n * oh * ow // (4 if adj == C.CONV_K2S2 else 1)
        tile_hint = self.tile_hints.get(label, self.tile_hints.get('*', 0) if self._tuning else 0)
        if tile_hint and ((ncols + 15) // 16) % (tile_hint & 15):
            tile_hint = 0
        nks = self.splitk_hints.get(label, 1)
        if self._trial_splitk:
            rt, ct = (tile_hint >> 4, tile_hint & 15) if tile_hint else (1, 1)
            waves = -(-rows // (16 * rt)) * (-(-ncols // 16) // ct)
            npad = -(-ncols // 16) * 16
            nks = self._trial_splitk if (waves < 4096 and rows * npad * abs(self._trial_splitk) <= (1 << 24)) else 1
        taps = 1 if adj == C.DECONV_K2S2 else 4
        flops = 2 * rows * taps * layer.n_ch_out * ncols
        # LDS-tiled kernel (csrc/conv_tile.hip) for the launches the plan-time trials gave to it: the adjoint families it has
        # (CONV_K2S1 / CONV_K2S2 of the expanding blocks, the transposed k2s1 of the encoder's stride-1 convs), no split epilogue
        wtn = (self._trial_wino or self.wino_hints.get(label, 0)) & 255
        if (wtn and self.use_wino and split is None and adj in (C.CONV_K2S1, C.DECONV_K2S1) and layer.n_ch_out % 8 == 0
                and (hi - lo) % wtn == 0 and ldp % 4 == 0 and ldo % 4 == 0 and layer.kernel.is_contiguous()):
            self._ran_wino.add(label)
            self._launch(label, nbytes, C.conv_wino_backward_data, adj, dpre, layer.n_ch_out, ldp, n, oh, ow,
                         layer.packed_adjoint_wino(lo, hi, wtn), hi - lo, wtn, out, ldo, mask_src=mask_src, ldm=ldm,
                         mask_alpha=mask_alpha, accumulate=accumulate, flops=flops)
            return
        tn = (self._trial_lds or (0 if self._trial_wino else self.lds_hints.get(label, 0))) & 255
        tile_ok = tn and ldp % 4 == 0 and ldo % 4 == 0 and layer.kernel.is_contiguous()
        if tile_ok and adj == C.DECONV_K2S2:                        # transposed k2s2: a GEMM with 4 (hi - lo) columns; takes the split
            tile_ok = layer.n_ch_out % 32 == 0 and (hi - lo) % 16 == 0 and (4 * (hi - lo)) % tn == 0
        elif tile_ok:
            tile_ok = (split is None and adj in (C.CONV_K2S1, C.CONV_K2S2, C.DECONV_K2S1) and layer.n_ch_out % 16 == 0
                       and (hi - lo) % tn == 0)
        if tile_ok:
            self._ran_lds.add(label)
            self._launch(label, nbytes, C.conv_tile_backward_data, adj, dpre, layer.n_ch_out, ldp, n, oh, ow,
                         layer.packed_adjoint_tile(lo, hi, tn), hi - lo, tn, out, ldo, mask_src=mask_src, ldm=ldm, mask_alpha=mask_alpha,
                         accumulate=accumulate, split=split, w_keras=ks, flops=flops)
            return
        if abs(nks) > 1:
            self._ran_splitk.add(label)
        self._launch(label, nbytes, C.conv_backward_data, adj, dpre, layer.n_ch_out, ldp, n, oh, ow, packed, zero_bias, hi - lo,
                     out, ldo, mask_src=mask_src, ldm=ldm, mask_alpha=mask_alpha, accumulate=accumulate, tile_hint=tile_hint,
                     ksplit=nks, split=split, w_keras=ks, flops=flops)

    def backward(self, dpred, base, cvis, lvis, nn_rgb, nn_base, obs_weights=None, generation=None):
        """Gradient of everything `forward` computed, given dpred = dL/d(pred) [N,H,W,3]; uses the
        activations the last forward left in the plan's buffers.  Weight gradients are ACCUMULATED
        into each layer's dkernel / dbias (views of the model's flat gradient bucket: zero it first).
        generation: `plan.generation` right after the forward this backward belongs to -- the plan keeps ONE set of
        activations, so a backward after any other forward (a second micro-batch, a vali / test call) would silently
        pair this pass's inputs with that pass's activations; it raises instead."""
        C.set_workspace_scope(id(self))
        if generation is not None and generation != self.generation:
            raise RuntimeError("backward() of a forward pass whose activations have been overwritten by a later forward "
                               "(pass %d, the plan now holds pass %d): run each backward before the next forward"
                               % (generation, self.generation))
        n, h, w, _ = base.shape
        k = nn_rgb.shape[1]
        b = self._buffers(n, k, h, w, base.device)
        g = self._grad_buffers(b)
        q, o, D, U, cl = self.q, self.o, self.n_down, self.n_up, b['C']
        zb = g['zero_bias']
        reg = getattr(self.q.layers[0], '_registry', None)
        if reg is not None and not self._tuning:
            reg.tick()
        if self.autotune and self.tune_backward and dpred.is_cuda and not b.get('tuned_bwd') and not self._tuning:
            # plan-time choice of the backward-data launches' wave tiles / split-K (the same trial machinery as the forward).
            # The trial passes accumulate into the weight-gradient bucket: it is cleared again before the real pass.
            b['tuned_bwd'] = True
            self._autotune(lambda: self._backward_streams(dpred, base, cvis, lvis, nn_rgb, nn_base, obs_weights, b, g, n, h, w, k),
                           backward=True)
            bucket = getattr(q.layers[0].dkernel, '_base', None)
            if bucket is not None:
                bucket.zero_()
        tkey = None
        if (self.use_tape and dpred.is_cuda and self.timer is None and reg is not None and obs_weights is None
                and all(t.is_contiguous() for t in (dpred, base, cvis, lvis, nn_rgb, nn_base))):
            tkey = ('bwd', dpred.data_ptr(), base.data_ptr(), cvis.data_ptr(), lvis.data_ptr(), nn_rgb.data_ptr(),
                    nn_base.data_ptr(), bool(b.get('train_fused')), self.bwd_streams, C._stream())
            tapes = b.setdefault('tapes', {})
            if len(tapes) > 16:                     # ever-changing input addresses (a loader that allocates per step): forget
                tapes.clear()
            ent = tapes.get(tkey, 0)
            if isinstance(ent, tuple):
                if C.tape_valid(ent, reg.version):
                    reg.touch_keys(ent[4])
                    C.replay(ent)
                    self.tape_replays += 1
                    return
                ent = 1
            tapes[tkey] = 1
            if ent == 1:
                C.tape_begin()
                reg.begin_record()
        try:
            self._backward_streams(dpred, base, cvis, lvis, nn_rgb, nn_base, obs_weights, b, g, n, h, w, k)
        except BaseException:
            if tkey is not None and ent == 1:
                C.tape_abort()
                reg.end_record()
            raise
        if tkey is not None and ent == 1:
            b['tapes'][tkey] = C.tape_end(reg.version, reg.end_record()) or 1

    def _backward_streams(self, dpred, base, cvis, lvis, nn_rgb, nn_base, obs_weights, b, g, n, h, w, k):
        concurrent = self.bwd_streams and dpred.is_cuda and self.timer is None
        if concurrent:
            if self._bside is None:
                self._bside = [torch.cuda.Stream(device=dpred.device), [], None,
                               torch.cuda.Stream(device=dpred.device) if self.bwd_streams > 1 else None]
            self._bside[2] = [0]                                     # event cursor: weight gradients go to the side stream
        try:
            self._backward_plan(dpred, base, cvis, lvis, nn_rgb, nn_base, obs_weights, b, g, n, h, w, k)
        finally:
            if concurrent:
                side, events, cur, side2 = self._bside
                self._bside[2] = None
                for sd in (side, side2):
                    if sd is None:
                        continue
                    if cur[0] == len(events):
                        events.append(C.new_event())
                    C.record_event(events[cur[0]], sd)
                    C.wait_event(torch.cuda.current_stream(), events[cur[0]])   # the optimizer step needs every gradient
                    cur[0] += 1

    def _backward_plan(self, dpred, base, cvis, lvis, nn_rgb, nn_base, obs_weights, b, g, n, h, w, k):
        q, o, D, U, cl = self.q, self.o, self.n_down, self.n_up, b['C']
        zb = g['zero_bias']
        mult = 2 if self.use_obs else 1         # use_obs = False (nlt.py:176-177): no observation half in any fm[l]

        # ---- head
        head = q.layers[-1]
        x_last = b['dec'][U - 1]
        cx = x_last.shape[-1]
        cs = mult * cl[0]
        fused = bool(b.get('train_fused'))
        if fused:
            # last expanding block + head in one launch (csrc/train_back.hip); the head's skip rows are F.front.bwd's
            (da, act_a), (db, _) = q.layers[D + U].convs()
            bb_in = (b['dec'][U - 2], b['fm'][1], b['dtmp'][U - 1], b['dec'][U - 1], dpred, n, h // 2, w // 2, da.kernel.detach(),
                     db.kernel.detach(), head.kernel.detach(), act_a.alpha)
            bb_w = (da.dkernel, da.dbias, db.dkernel, db.dbias, head.dkernel, head.dbias)
            # (F.back.bwd as a backward-data launch on the chain + a weight-gradient launch on the side stream was built and measured
            # in r05 -- 3.16 vs 3.06 ms per l2 step -- and removed in r06.)
            self._launch('F.back.bwd', 4 * n * h * w * (4 + 4 + 3 + 20), C.back_backward, *bb_in, g['dec'][U - 2], g['fm'][1], *bb_w)
        else:
            self._launch('bwd.head', 4 * n * h * w * (2 * (cx + cs) + 3), C.head_backward, x_last, cx, cx, b['fm'][0], cs, cs,
                         head.kernel.detach(), dpred, n, h, w, g['dec'][U - 1], cx, g['fm'][0], cs, head.dkernel, head.dbias)

        # ---- decoder (expanding blocks), last to first
        hh, ww = h, w
        if fused:
            hh, ww = h // 2, w // 2
        masked = fused      # g['dec'][j] already holds the gradient w.r.t. the PRE-activation (mask fused into its producer)
        # One observation per frame (the training configs): the `level_split` pass of every level -- LeakyReLU' of the query
        # half of dfm[l], the observation mean's adjoint + the observation path's own gradient + its LeakyReLU' -- is the
        # epilogue of the LAST backward-data launch that writes dfm[l] (level D: the bottleneck's skip half; level l < D:
        # level l + 1's query stride-2 conv).  That launch has to come after the observation path's own gradient of the
        # level exists, so a level's observation convs are taken before its query convs.
        fold = self.fold_split and self.use_obs and obs_weights is None and k == 1

        def split_of(l):
            (_, _), (_, oact) = o.layers[l].convs()
            (_, _), (_, qact) = q.layers[l].convs()
            return dict(mask_src=b['fm'][l], ldm=2 * cl[l], mask_alpha=qact.alpha,
                        split=(cl[l], b['obs'][l], g['obs'][l], oact.alpha, l < D))
        for j in range(U - 2 if fused else U - 1, -1, -1):
            (da, act_a), (db, act_b) = q.layers[D + 1 + j].convs()
            nl = db.n_ch_out
            lab = 'bwd.L%d.q' % (D + 1 + j)
            # s1:  dec[j] = act(deconv_s1(dtmp[j]))
            if not masked:
                self._launch(lab + '.s1.act', 12 * n * hh * ww * nl, C.lrelu_backward, g['dec'][j], nl, b['dec'][j], nl, nl,
                             n * hh * ww, act_b.alpha, g['dec'][j], nl)
            self._wgrad(lab + '.s1.wgrad', db, b['dtmp'][j], nl, nl, None, 0, 0, n, hh, ww, g['dec'][j], nl)
            self._dgrad(lab + '.s1.dgrad', db, 0, nl, g['dec'][j], nl, n, hh, ww, g['dtmp'][j], nl,
                        mask_src=b['dtmp'][j], ldm=nl, mask_alpha=act_a.alpha, zero_bias=zb)
            # s2:  dtmp[j] = act(deconv_s2(concat(x, skip)))
            if j > 0:
                x, cxj = b['dec'][j - 1], b['dec'][j - 1].shape[-1]
                dx = g['dec'][j - 1]
            else:
                x, cxj = b['fm'][D], mult * cl[D]
                dx = g['fm'][D]
            skip, csj, dskip = b['fm'][D - j], mult * cl[D - j], g['fm'][D - j]
            self._wgrad(lab + '.s2.wgrad', da, x, cxj, cxj, skip, csj, csj, n, hh // 2, ww // 2, g['dtmp'][j], nl)
            if j > 0:
                # dx = gradient w.r.t. dec[j-1], whose own LeakyReLU derivative is applied in this launch's epilogue
                prev_act = q.layers[D + j].convs()[1][1]
                self._dgrad(lab + '.s2.dgrad.x', da, 0, cxj, g['dtmp'][j], nl, n, hh, ww, dx, cxj, mask_src=x, ldm=cxj,
                            mask_alpha=prev_act.alpha, zero_bias=zb)
                masked = True
            else:
                self._dgrad(lab + '.s2.dgrad.x', da, 0, cxj, g['dtmp'][j], nl, n, hh, ww, dx, cxj, zero_bias=zb)
            self._dgrad(lab + '.s2.dgrad.skip', da, cxj, cxj + csj, g['dtmp'][j], nl, n, hh, ww, dskip, csj,
                        accumulate=(j == 0), zero_bias=zb, **(split_of(D) if (fold and j == 0) else {}))
            hh, ww = hh // 2, ww // 2

        # every weight gradient of the expanding blocks is queued now: the leading range of the flat gradient
        # bucket (models/nlt.py:_flatten) can start its all-reduce while the encoder's backward runs
        self._grad_range_done(0)

        # ---- encoder (contracting blocks), deepest first; hh, ww = dims of level D
        for l in range(D, 0, -1):
            (qa, qact_a), (qb, qact_b) = q.layers[l].convs()
            (oa, oact_a), (ob, oact_b) = o.layers[l].convs()
            c, cp = cl[l], cl[l - 1]
            lab = 'bwd.L%d' % l
            if fold:
                pass                                                # dfm[l] / dobs[l] arrived finished (see `fold` above)
            elif self.use_obs and obs_weights is None:
                # both halves of dfm[l] in one launch: query half -> gradient w.r.t. q.s1's pre-activation (in place);
                # observation half: the mean's gradient distributed, the obs path's own added, its activation backward
                self._launch(lab + '.split', 4 * n * hh * ww * c * (3 + 1 + 3 * k), C.level_split_backward, g['fm'][l], b['fm'][l],
                             2 * c, b['obs'][l], None, g['obs'][l] if l < D else None, n, k, hh * ww, c, qact_b.alpha,
                             oact_b.alpha, g['obs'][l])
            else:
                # query half of dfm[l] -> gradient w.r.t. the pre-activation of q.s1
                self._launch(lab + '.q.s1.act', 12 * n * hh * ww * c, C.lrelu_backward, g['fm'][l], mult * c, b['fm'][l], mult * c,
                             c, n * hh * ww, qact_b.alpha, g['fm'][l], mult * c)
                if self.use_obs:
                    # observation half: distribute the mean's gradient, add the obs path's own, activation backward
                    self._launch(lab + '.o.mean', 4 * n * hh * ww * c * (1 + 3 * k), C.obs_mean_backward,
                                 g['fm'][l].view(-1)[c:], 2 * c, b['obs'][l], obs_weights, g['obs'][l] if l < D else None,
                                 n, k, hh * ww, c, oact_b.alpha, g['obs'][l])
            if self.use_obs:
                # o.s1 / o.s2 (n*k observation frames) -- before the query convs: q.s2's backward-data finishes dobs[l - 1]
                self._wgrad(lab + '.o.s1.wgrad', ob, b['otmp'][l], c, c, None, 0, 0, n * k, hh, ww, g['obs'][l], c)
                self._dgrad(lab + '.o.s1.dgrad', ob, 0, c, g['obs'][l], c, n * k, hh, ww, g['otmp'][l], c,
                            mask_src=b['otmp'][l], ldm=c, mask_alpha=oact_a.alpha, zero_bias=zb)
                if not (fused and l == 1):
                    self._wgrad(lab + '.o.s2.wgrad', oa, b['obs'][l - 1], cp, cp, None, 0, 0, n * k, 2 * hh, 2 * ww,
                                g['otmp'][l], c)
                    self._dgrad(lab + '.o.s2.dgrad', oa, 0, cp, g['otmp'][l], c, n * k, hh, ww, g['obs'][l - 1], cp, zero_bias=zb)
            # q.s1 / q.s2
            self._wgrad(lab + '.q.s1.wgrad', qb, b['qtmp'][l], c, c, None, 0, 0, n, hh, ww, g['fm'][l], mult * c)
            self._dgrad(lab + '.q.s1.dgrad', qb, 0, c, g['fm'][l], mult * c, n, hh, ww, g['qtmp'][l], c,
                        mask_src=b['qtmp'][l], ldm=c, mask_alpha=qact_a.alpha, zero_bias=zb)
            if not (fused and l == 1):
                self._wgrad(lab + '.q.s2.wgrad', qa, b['fm'][l - 1], mult * cp, mult * cp, None, 0, 0, n, 2 * hh, 2 * ww,
                            g['qtmp'][l], c)
                self._dgrad(lab + '.q.s2.dgrad', qa, 0, mult * cp, g['qtmp'][l], c, n, hh, ww, g['fm'][l - 1], mult * cp,
                            accumulate=True, zero_bias=zb, **(split_of(l - 1) if (fold and l > 1) else {}))
            hh, ww = hh * 2, ww * 2
            if l == self.grad_mid_level and l > 1:
                # levels D .. l of both paths -- the bulk of the bucket -- are queued: the second range goes out while
                # the wide, cheap-in-parameters levels below still run
                self._grad_range_done(1)

        # ---- L0 (both paths)
        q0, o0 = q.layers[0], o.layers[0]
        if fused:
            (qa, _), _ = q.layers[1].convs()
            (oa, _), _ = o.layers[1].convs()
            D_ = lambda t: t.detach()
            self._launch('F.front.bwd', 4 * n * h * w * (5 + 6 * k + 3) + 4 * n * (h // 2) * (w // 2) * 16 * (1 + k),
                         C.front_backward, base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, g['qtmp'][1], g['otmp'][1], dpred,
                         (D_(q0.kernel), D_(q0.bias), D_(o0.kernel), D_(o0.bias), D_(qa.kernel), D_(oa.kernel), D_(head.kernel)),
                         (q0.dkernel, q0.dbias, o0.dkernel, o0.dbias, qa.dkernel, qa.dbias, oa.dkernel, oa.dbias, head.dkernel))
            return
        if not self.use_obs:
            # query L0 alone (the observation net is evaluated by the reference but feeds nothing: its gradients stay zero)
            self._launch('bwd.L0.q', 4 * n * h * w * (5 + cl[0]), C.conv_backward_weights, C.CONV1X1, b['x5'], 5, 5, None, 0, 0,
                         n, h, w, g['fm'][0], cl[0], cl[0], q0.dkernel, q0.dbias)
            return
        self._launch('bwd.L0.stem', 4 * n * h * w * (5 + 6 * k + 2 * cl[0] + k * cl[0]), C.stem_backward,
                     base, cvis, lvis, nn_rgb, nn_base, obs_weights, n, k, h, w, cl[0], g['fm'][0], g['obs'][0],
                     q0.dkernel, q0.dbias, o0.dkernel, o0.dbias)
