"""GAEngine: the static launch plans (weight prep / forward / backward) of one GA_ConvNeXt for a fixed
(batch, train|eval, math mode), over persistent NHWC device buffers.

Data layout in HBM (bf16 mode; fp32 mode is identical with 4-byte elements):
  * activations: row-major [B*H*W, C] (= NHWC), bf16; per-row LayerNorm rstd fp32; BatchNorm stats fp32 [C];
  * weights: fp32 masters in ONE flat buffer ([decay | no-decay]); per step they are re-laid-out once into the
    "effective" bf16 GEMM operands (k = (ky,kx,ci); LayerNorm scale and LayerScale gamma folded in; a transposed
    copy for the data-gradient product) by ga_weight_prep -- so the hot GEMMs only ever see bias / GELU / residual
    epilogues;
  * gradients: fp32, ONE flat buffer aliased by every param.grad; the wgrad kernels atomically accumulate either
    directly into it or into a zeroed scratch arena of effective-weight gradients that ga_weight_unfold maps back.
What is saved for backward per ConvNeXt block: xhat (LN output, no affine), rstd, h (pre-GELU hidden) and the
block output of fc1 as a = gelu(h) and g = gelu'(h) (both written by the fc1 epilogue), and the block output.

Reference semantics restated here: /root/reference/GA/ga_convnext.py:98-112 (block), :139-150 (stage + taps),
:294-318 (Bottleneck), :452-467 (get_gram), :153-248 (class attention block), :469-505 (forward).
"""
import contextlib
import os

import torch

from . import ops
from .ops import (A_CONV3, A_PATCH2, A_STEM4_NCHW, ACT_GELU, ASYNC_LANE, C_UNPATCH2, GA_BF16, GA_F32, Plan)


def pad8(n):
    return (n + 7) // 8 * 8


def tap_indices(nblocks, naggre):
    """ga_convnext.py:141-147"""
    taps = []
    if nblocks > 5:
        for i in range(nblocks):
            if (i + 1) % (nblocks // (naggre + 1)) == 0 and len(taps) < naggre:
                taps.append(i)
    return taps


class GAFunction(torch.autograd.Function):
    """Autograd glue: one node for the whole network. Parameter gradients are accumulated by the HIP kernels
    straight into the flat gradient buffer behind every param.grad (so this node returns no tensor grads)."""

    @staticmethod
    def forward(ctx, eng, x, anchor):
        ctx.eng = eng
        return eng.forward(x)

    @staticmethod
    def backward(ctx, dlogits):
        ctx.eng.backward(dlogits)
        return None, None, None


class GAEngine:
    def __init__(self, model, batch, training, mode):
        self.m = model
        self.cfg = model.cfg
        self.B = batch
        self.training = training
        self.dt = GA_BF16 if mode == 'bf16' else GA_F32
        self.tdt = ops.torch_dtype(self.dt)
        flat = model.flat_state()
        self.dev = flat['params'].device
        self.P = dict(model.named_parameters())
        self.Bf = dict(model.named_buffers())
        self.bufs = {}
        self.tmps = {}
        # trunk weight-gradient launches on the backward plan's asynchronous lane (GAEXT_ASYNC_WGRAD=0: in line)
        self.async_wgrad = os.environ.get('GAEXT_ASYNC_WGRAD', '1') != '0'
        self.fwd_split = max(1, int(os.environ.get('GAEXT_FWD_SPLIT', '2')))
        self.par_branch = os.environ.get('GAEXT_PAR_BRANCH', '1') != '0'   # stage-4 shortcut branch beside the main branch (forward)
        self.fwd_skew = int(os.environ.get('GAEXT_FWD_SKEW', '-1'))   # chain k+1 starts when chain k has passed this stage
        self._chain = None
        self._cur_stage = 0
        self._bwd_seq = 0        # trunk blocks recorded on the backward plan so far
        self._pre_dyz = {}       # block prefix -> DropPath-scaled dy already written by the block before it (backward order)
        self.fuse_dp = os.environ.get('GAEXT_FUSE_DP', '1') != '0'
        self.sync_bn = getattr(model, 'sync_bn_comm', None)      # FlatModel.convert_sync_batchnorm(comm): --sync-bn (GA/train.py:449-455)
        self.W = {}
        self.weights_dirty = True
        self.anchor = torch.zeros((), device=self.dev, requires_grad=True)
        self.input_descs = []
        self.blocks = {}
        self.x_ref = None
        self.img = 224
        # DropPath schedule (ga_convnext.py:362,376,413)
        self.dp_rates = self._drop_path_rates()
        self.dp_scale = {}   # block prefix -> fp32 [B] (mask / keep): rows of ONE (n_sites, B) tensor
        if training:
            sites = [pre for pre, r in self.dp_rates.items() if r > 0]
            if sites:
                self.dp_all = torch.ones(len(sites), batch, device=self.dev)
                self.dp_keep = torch.tensor([1.0 - self.dp_rates[p] for p in sites], device=self.dev)
                self.dp_counter = torch.zeros(1, dtype=torch.int64, device=self.dev)
                self.dp_plan = Plan(name='droppath')
                self.dp_plan.drop_path_sample(self.dp_all, self.dp_keep, len(sites), batch, torch.initial_seed(), self.dp_counter)
                for i, pre in enumerate(sites):
                    self.dp_scale[pre] = self.dp_all[i]
        # scratch arena for effective-weight gradients (zeroed once per backward)
        self.arena = None
        self.arena_off = 0
        if training:
            self.arena = torch.zeros(int(flat['total'] * 1.15) + (1 << 20) + self._arena_extra(), device=self.dev)
        self._nbt = [t for n, t in self.Bf.items() if n.endswith('num_batches_tracked')]
        self.prep = Plan(name='prep', defer_small=True)
        self.fwd = Plan(name='fwd')
        self.bwd = Plan(name='bwd', defer_small=True) if training else None
        # BatchNorm column-sum accumulators live in one pool that the forward plan zeroes with a single memset
        self.bn_pool = torch.zeros(1 << 16, device=self.dev)
        self.bn_pool_off = 0
        self.bott_prefix = 'stages.4.'   # the SE-Bottleneck's parameter prefix (GA-CSWin with stage5='bottleneck': 'stage5.')
        # zero-padded copies of parameters of the odd-width variants: name -> (padded buffer, how to copy); gradients: name -> arena buffer
        self.ppad, self.pgrad = {}, {}
        self.pad_gmlp = os.environ.get('GAEXT_PAD_GMLP', '1') != '0'    # odd-width grouped one-token layers on padded MFMA layouts
        self._build()
        assert not self.pgrad or getattr(self, '_unpadded', False), 'padded parameter gradients were never copied back'

    # ------------------------------------------------------------------------------------------
    # buffers
    # ------------------------------------------------------------------------------------------
    def buf(self, name, shape, dtype=None, zero=False):
        dtype = dtype or self.tdt
        if name not in self.bufs:
            self.bufs[name] = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.dev)
        t = self.bufs[name]
        assert tuple(t.shape) == tuple(shape) and t.dtype == dtype, name
        return t

    def _chains(self):
        """(lane, first image, end image) of the forward trunk's independent batch parts: with GAEXT_FWD_SPLIT=n > 1
        stages 0-3 run as n chains on side streams (rows of different images never mix before the stage-4 BatchNorm),
        so that one chain's launches fill the tails of the other's"""
        n = self.fwd_split if self.B >= 2 * self.fwd_split else 1
        if n <= 1:
            return [(0, 0, self.B)]
        per = self.B // n
        cuts = [s_ * per for s_ in range(n)] + [self.B]
        return [(1 + s_, cuts[s_], cuts[s_ + 1]) for s_ in range(n)]

    def _fsplits(self, HW):
        """(lane, first row, end row, first image, end image) the current pass of the trunk records"""
        ch = getattr(self, '_chain', None)
        if ch is None:
            return [(0, 0, self.B * HW, 0, self.B)]
        lane, b0, b1 = ch
        return [(lane, b0 * HW, b1 * HW, b0, b1)]

    @contextlib.contextmanager
    def _wlane(self):
        """weight-gradient launches recorded inside go to the backward plan's asynchronous lane (trunk / shared parts
        only: inside a head's lane they stay in that lane)"""
        Bk = self.bwd
        prev = Bk.lane
        if self.async_wgrad and prev == 0:
            Bk.lane = ASYNC_LANE
        try:
            yield
        finally:
            Bk.lane = prev

    def tmp(self, tag, shape, dtype=None):
        """transient buffer shared by every call site with the same (tag, shape, dtype) -- stream order makes it safe"""
        dtype = dtype or self.tdt
        key = (getattr(self, 'tmp_prefix', '') + tag, tuple(shape), dtype)   # per-head copies when the heads run concurrently
        if key not in self.tmps:
            self.tmps[key] = torch.empty(shape, dtype=dtype, device=self.dev)
        return self.tmps[key]

    def act(self, name, shape, dtype=None):
        """activation saved for backward (uniquely named, persistent)"""
        return self.buf(name, shape, dtype)

    def blk_act(self, name, shape, dtype=None):
        """per-block saved activation: persistent when training, one shared transient per shape in eval"""
        return self.buf(name, shape, dtype) if self.training else self.tmp(name.rsplit('.', 1)[-1], shape, dtype)

    def gbuf(self, shape):
        n = 1
        for s in shape:
            n *= s
        off = (self.arena_off + 63) // 64 * 64
        assert off + n <= self.arena.numel(), 'gradient scratch arena too small'
        self.arena_off = off + n
        return self.arena[off:off + n].view(shape)

    def grad(self, name):
        return self.P[name].grad

    def _drop_path_rates(self):
        dep = self.cfg['depths']
        rate = self.cfg['drop_path_rate']
        pts = torch.linspace(0, rate, sum(dep)).split(list(dep))
        out = {}
        for i in range(4):
            for j in range(dep[i]):
                out[f'stages.{i}.blocks.{j}.'] = float(pts[i][j])
        out['stages.4.'] = float(rate)
        for k in range(self.cfg['branches']):
            out[f'gram_layer.{k}.blocks.0.'] = float(pts[-1][0])   # dp_rates[-1] (ga_convnext.py:413)
            out[f'ga.{k}.'] = 0.0                                    # LayerScaleBlockClassAttn default drop_path=0
        return out

    def sample_drop_path(self):
        """fresh per-sample Bernoulli(keep)/keep factors for every stochastic-depth site (timm DropPath): one launch of
        ga_drop_path_sample, keyed by torch.initial_seed() at build time and a device-side call counter"""
        if self.dp_scale:
            self.dp_plan.run()

    def set_drop_path_masks(self, masks):
        for pre, t in self.dp_scale.items():
            t.copy_(masks[pre].to(self.dev).float())

    # ------------------------------------------------------------------------------------------
    # build
    # ------------------------------------------------------------------------------------------
    # parameter names of the ConvNeXt trunk: GA-ConvNeXt follows timm's ConvNeXt (ga_convnext.py:86-137,356-359), MAP-ConvNeXt
    # the FB one (map_convnext.py:16-83) -- same arithmetic, different module names
    NAMES = dict(stem_conv='stem.0.', stem_ln='stem.1.', ds_ln='stages.{i}.downsample.0.', ds_conv='stages.{i}.downsample.1.',
                 block='stages.{i}.blocks.{j}.', dw='conv_dw.', fc1='mlp.fc1.', fc2='mlp.fc2.')

    def _build(self):
        cfg = self.cfg
        d = cfg['dims']
        B, T, F = self.B, self.training, self.fwd
        dt = self.dt
        feats, taps, stage_in, _ = self._build_trunk()
        # ---------------- aggregate (ga_convnext.py:479-483) ----------------
        Hc = 14
        M4 = B * Hc * Hc
        ctot = sum(d[:-1]) + d[2] * cfg['naggre']
        cat = self.act('agg.cat', (M4, ctot))
        segs = [(feats[0][0], feats[0][1], d[0], 0), (feats[1][0], feats[1][1], d[1], 0)]
        segs += [(t, feats[2][1], d[2], 0) for t in taps]
        segs += [(feats[2][0], feats[2][1], d[2], 0), (feats[3][0], feats[3][1], d[3], 1)]
        off = 0
        self.agg_segs = []
        for src, hw, c, mode in segs:
            F.pool_concat_fwd(src, cat, B, hw, hw, c, Hc, Hc, ctot, off, mode, dt, label=f'agg.{off}')
            self.agg_segs.append((src, hw, c, mode, off))
            off += c
        assert off == ctot
        # ---------------- Bottleneck "stage 4" ----------------
        x4 = self._bottleneck_fwd(cat, M4, ctot, d[4])
        self.cout = d[4]
        self._build_heads(x4, M4, Hc)
        # ---------------- backward ----------------
        if T:
            self._build_backward(feats, taps, stage_in, x4, M4, ctot)
            if self.async_wgrad:
                self.bwd.join_async()
            self.bwd.flush('end.')
        self.prep.flush('prep.')

    def _build_trunk(self):
        """stem + the four ConvNeXt stages; returns (feats [(x, res)], taps, stage_in [(x, H)], stem output)"""
        cfg, nm = self.cfg, self.NAMES
        d, dep = cfg['dims'], cfg['depths']
        B, T, F = self.B, self.training, self.fwd
        dt = self.dt
        S0 = self.img // 4
        self.x_in = None
        # ---------------- stem ----------------
        M0 = B * S0 * S0
        if T:
            F.zero(self.bn_pool, label='zero.bn_sums')
        sc, sl = nm['stem_conv'], nm['stem_ln']
        Wst = self._w_plain(sc + 'weight', d[0], 3, 4, 4, stem=True, need_T=False)
        stem_pre = self.act('stem.pre', (M0, d[0]))
        self.x_placeholder = torch.zeros(8, device=self.dev)  # patched by set_input
        mean = self.act('stem.mean', (M0,), torch.float32)
        rstd = self.act('stem.rstd', (M0,), torch.float32)
        x = self.buf('stem.out', (M0, d[0]))
        self.stem_call = None
        if dt == GA_BF16 and d[0] in (96, 128) and self.img % 4 == 0 and os.environ.get('GAEXT_STEM_FUSED', '1') != '0':
            # conv + bias + LayerNorm in one pass over the image (ga_stem4_ln_fwd): the input pointer is argument 0 of this call
            self.stem_call = len(F.calls)
            F.stem4_ln_fwd(self.x_placeholder, Wst, Wst.shape[1], self.P[sc + 'bias'], self.P[sl + 'weight'], self.P[sl + 'bias'],
                           stem_pre, x, mean, rstd, B, self.img, self.img, d[0], 1e-6, label='stem.conv+ln')
        else:
            F.gemm(self.x_placeholder, Wst, stem_pre, M0, d[0], 48, dt, a_kind=A_STEM4_NCHW, a_dims=(self.img, self.img, 3),
                   bias=self.P[sc + 'bias'], label='stem.conv')
            self.input_descs.append(self._last_desc(F))
            F.layernorm_fwd(stem_pre, self.P[sl + 'weight'], self.P[sl + 'bias'], x, mean, rstd, M0, d[0], 1e-6, dt,
                            label='stem.ln')
        # ---------------- stages 0..3 ----------------
        # one pass per chain (batch part): with GAEXT_FWD_SPLIT > 1 the chains run on side streams; every pass names
        # the same full-batch buffers and records its own rows only
        x_stem = x
        chains = self._chains()
        skew_ev = None
        for ci, chain in enumerate(chains):
            self._chain = chain if len(chains) > 1 else None
            if self._chain is not None and skew_ev is not None:
                F.lane_wait(chain[0], skew_ev)       # start this chain when the previous one has passed stage fwd_skew
                skew_ev = None
            feats, taps = [], []
            x = x_stem
            res = S0
            stage_in = []
            for i in range(4):
                if i > 0:
                    Hp = res
                    res //= 2
                    Mi = B * res * res
                    Mp = B * Hp * Hp
                    pre = f'stages.{i}.downsample.'          # buffer names only
                    pln, pcv = nm['ds_ln'].format(i=i), nm['ds_conv'].format(i=i)
                    ln = self.act(pre + 'ln', (Mp, d[i - 1]))
                    mean = self.act(pre + 'mean', (Mp,), torch.float32)
                    rstd = self.act(pre + 'rstd', (Mp,), torch.float32)
                    Wd = self._w_plain(pcv + 'weight', d[i], d[i - 1], 2, 2)
                    xo = self.buf(pre + 'out', (Mi, d[i]))
                    for lane, r0, r1, b0, b1 in self._fsplits(Hp * Hp):
                        F.lane = lane
                        F.layernorm_fwd(x[r0:r1], self.P[pln + 'weight'], self.P[pln + 'bias'], ln[r0:r1], mean[r0:r1],
                                        rstd[r0:r1], r1 - r0, d[i - 1], 1e-6, dt, label=pre + 'ln')
                        F.gemm(ln[r0:r1], Wd, xo[r0 // 4:r1 // 4], (r1 - r0) // 4, d[i], 4 * d[i - 1], dt, a_kind=A_PATCH2,
                               a_dims=(Hp, Hp, d[i - 1]), bias=self.P[pcv + 'bias'], label=pre + 'conv')
                    F.lane = 0
                    stage_in.append((x, Hp))
                    x = xo
                tap_at = tap_indices(dep[i], cfg['naggre']) if i == 2 else []
                for j in range(dep[i]):
                    x = self._block_fwd(nm['block'].format(i=i, j=j), x, res, d[i])
                    if j in tap_at:
                        taps.append(x)
                feats.append((x, res))
                if self._chain is not None and i == self.fwd_skew and ci + 1 < len(chains):
                    skew_ev = F.lane_signal(chain[0])
        self._chain = None
        return feats, taps, stage_in, x_stem

    def _build_heads(self, x4, M4, Hc):
        """the five GA heads on the stage-4 / stage-5 map x4 [M4, cout] (ga_convnext.py:491-504, ga_cswin.py:677-692)"""
        cfg, B, T, F, dt = self.cfg, self.B, self.training, self.fwd, self.dt
        cout = self.cout
        NC, K = cfg['num_classes'], cfg['branches']
        assert NC % 8 == 0, 'num_classes must be a multiple of 8 (pad the classifier)'
        self.logits = self.buf('logits', (K, B, NC), torch.float32)
        self.heads = []
        # LayerScaleBlockClassAttn.norm1 acts row-wise on cat(x_cls, tokens) (ga_convnext.py:244-246): the normalised
        # image tokens are the SAME for all heads (only the affine part differs, and that is folded into each head's
        # k | v | q weights like the ConvNeXt block's LayerNorm into fc1).  One LayerNorm over the tokens instead of five
        # over the concatenation, no concatenated copy, one backward LayerNorm over the summed gradient.
        E_, nh_ = cfg['dim_embed'], cfg['num_heads']
        # ga_class_attn_*2 wants head widths that are multiples of 8 and E <= 512.  Odd head widths (688 / 976 variants:
        # dim_embed 168 / 240 = 8 heads of 21 / 30) run hd_p = pad8(hd) wide on zero-padded COPIES of the q / k / v rows and the
        # proj columns (the Bottleneck's scheme, _bott_pad): the extra channels of q, k, v and of the attention output are exact
        # zeros, so are their gradients; the real part of each padded weight gradient is copied back after the backward pass.
        hd_ = E_ // max(nh_, 1)
        self.hd_p = pad8(hd_)
        self.Ea = nh_ * self.hd_p
        self.shared_tok = E_ % nh_ == 0 and self.Ea <= 512
        if self.shared_tok and self.Ea != E_:
            for k in range(K):
                pre = f'ga.{k}.attn.'
                pk, pv = self.P[pre + 'k.weight'], self.P[pre + 'v.weight']
                assert pv.data_ptr() == pk.data_ptr() + pk.numel() * 4, 'k/v weights must be adjacent in the flat buffer'
                # (name, padded shape, rows, cols, source row pitch, padded row pitch)
                spec = (('k.weight', (2 * self.Ea, cout), 2 * nh_, hd_ * cout, hd_ * cout, self.hd_p * cout),
                        ('q.weight', (self.Ea, cout), nh_, hd_ * cout, hd_ * cout, self.hd_p * cout),
                        ('proj.weight', (cout, self.Ea), cout * nh_, hd_, hd_, self.hd_p))
                for name, shape, rows, cols, lds, ldd in spec:
                    buf = self.buf('pad.' + pre + name, shape, torch.float32, zero=True)
                    self.prep.pad_copy_f32(self.P[pre + name], buf, rows, cols, lds, ldd, label='prep.pad.' + pre + name)
                    self.ppad[pre + name] = (buf, 'copy', (rows, cols, lds, ldd))
        if self.shared_tok:
            self.tok = dict(xn=self.act('ga.tok.xn', (M4, cout)), rstd=self.act('ga.tok.rstd', (M4,), torch.float32))
            F.layernorm_fwd(x4, None, None, self.tok['xn'], None, self.tok['rstd'], M4, cout, 1e-5, dt, label='ga.tok.ln')
            # the k | v rows of the image tokens of ALL heads from one GEMM over the shared tokens: the heads' effective
            # (norm1-folded) k|v weights are stacked along N; head k reads / writes the column slice [k*2E, (k+1)*2E)
            E2 = 2 * self.Ea
            tk = self.tok
            tk['E2'], tk['ld'] = E2, K * E2
            tk['W'] = self.buf('w.ga.kv_all', (K * E2, cout))
            tk['WT'] = self.buf('wT.ga.kv_all', (cout, K * E2)) if T else None
            tk['b'] = self.buf('w.ga.bkv_all', (K * E2,), torch.float32)
            P = self.P
            for k in range(K):
                pre = f'ga.{k}.'
                pk, pv = P[pre + 'attn.k.weight'], P[pre + 'attn.v.weight']
                assert pv.data_ptr() == pk.data_ptr() + pk.numel() * 4, 'k/v weights must be adjacent in the flat buffer'
                pk = self._pw(pre + 'attn.k.weight')
                self.prep.weight_prep(pk, 1, E2, cout, 1, 1, dt, out=tk['W'][k * E2:], ldo=cout,
                                      outT=tk['WT'][:, k * E2:] if T else None, ldt=K * E2 if T else 0,
                                      cs=P[pre + 'norm1.weight'], t_cols=E2, label='prep.' + pre + 'kv')
                self.prep.bias_fold(pk, None, None, P[pre + 'norm1.bias'], tk['b'][k * E2:], E2, cout)
            tk['kv'] = self.act('ga.kv_all', (M4, K * E2))
            F.gemm(tk['xn'], tk['W'], tk['kv'], M4, K * E2, cout, dt, bias=tk['b'], label='ga.kv_all')
        self._contract_all_fwd(x4, M4)
        # classifiers: inputs of the five heads in one [K][B][cout] buffer, weights stacked -> one batched GEMM
        fa = self.fc_all = dict(x=self.act('fc.all.x', (K, B, cout)))
        ldn = pad8(NC)
        fa['W'] = self.buf('w.fc.all', (K, NC, cout))
        fa['WT'] = self.buf('wT.fc.all', (K, cout, ldn)) if T else None
        fa['b'] = self.buf('w.fc.ball', (K, NC), torch.float32)
        for k in range(K):
            self.prep.weight_prep(self.P[f'fc.{k}.weight'], 1, NC, cout, 1, 1, dt, out=fa['W'][k], ldo=cout,
                                  outT=fa['WT'][k] if T else None, ldt=ldn if T else 0, label=f'prep.fc.{k}')
            self.prep.bias_fold(None, self.P[f'fc.{k}.bias'], None, None, fa['b'][k], NC, cout)
        # the heads are independent chains of mostly small launches: with GAEXT_HEAD_STREAMS=n > 1 head k runs on side
        # stream k % n (between a fork / join of the plan) with its own transient buffers
        self.head_lanes = int(os.environ.get('GAEXT_HEAD_STREAMS', '5')) if self.shared_tok else 1
        for k in range(K):
            if self.head_lanes > 1:
                F.lane, self.tmp_prefix = 1 + k % self.head_lanes, f'h{k}.'
            self.heads.append(self._head_fwd(k, x4, M4, cout, Hc))
        F.lane, self.tmp_prefix = 0, ''
        F.gemm(fa['x'], fa['W'], self.logits, B, NC, cout, dt, batch=K, strideA=B * cout, strideB=NC * cout, strideC=B * NC,
               bias=fa['b'], strideBias=NC, c_f32=True, label='fc.all')

    def _arena_extra(self):
        """floats of gradient scratch beyond 1.15 x the parameters: the padded gradient copies of the odd-width variants"""
        cfg = self.cfg
        d = cfg.get('dims')
        if not d or len(d) < 5 or any(k not in cfg for k in ('gram_dim', 'gram_groups', 'mlp_groups', 'dim_embed', 'num_heads', 'branches')):
            return 0
        cout, g, groups = d[4], cfg['gram_dim'], cfg['gram_groups']
        if cout % (8 * groups) == 0 and cout % (8 * cfg['mlp_groups']) == 0 and (cfg['dim_embed'] // max(cfg['num_heads'], 1)) % 8 == 0:
            return 0
        cp = cout + 8 * max(groups, cfg['mlp_groups'])
        return cfg['branches'] * (cp * (g * (g + 1) // 2 // groups + 8) + 8 * cp * cp // cfg['mlp_groups'] + 4 * cp * 512) + (1 << 20)

    def _pw(self, name):
        """master copy of a parameter: the zero-padded one where the layer runs on a padded layout"""
        return self.ppad[name][0] if name in self.ppad else self.P[name]

    def _pg(self, name):
        """gradient buffer of a parameter (padded: a zeroed arena buffer, copied back by _unpad_all)"""
        if name not in self.ppad:
            return self.grad(name)
        if name not in self.pgrad:
            self.pgrad[name] = self.gbuf(tuple(self.ppad[name][0].shape))
        return self.pgrad[name]

    def _pad_groups(self, name, R, Cdim, RG, RGp, CG, CGp):
        """register (once) the two-level group-padded copy of parameter `name` ([R][C] -> [R/RG*RGp][C/CG*CGp], ga_pad_groups_f32)"""
        if name not in self.ppad:
            buf = self.buf('pad.' + name, (R // RG * RGp, Cdim // CG * CGp), torch.float32, zero=True)
            self.prep.pad_groups_f32(self.P[name], buf, R, Cdim, RG, RGp, CG, CGp, label='prep.pad.' + name)
            self.ppad[name] = (buf, 'groups', (R, Cdim, RG, RGp, CG, CGp))
        return self.ppad[name][0]

    def _unpad_all(self):
        """real part of every padded parameter gradient back into the parameter's gradient; call AFTER the flush of the heads'
        deferred weight-unfold jobs (they write the padded gradients) and before the 'heads' mark"""
        self._unpadded = True
        for name, g in self.pgrad.items():
            _, kind, a = self.ppad[name]
            if kind == 'copy':
                rows, cols, lds, ldd = a
                self.bwd.pad_copy_f32(g, self.grad(name), rows, cols, ldd, lds, accumulate=True, label=name + '.unpad')
            else:
                self.bwd.pad_groups_f32(g, self.grad(name), *a, unpad=True, accumulate=True, label=name + '.unpad')

    def _contract_all_fwd(self, x4, M4):
        cfg, T, F, dt, cout = self.cfg, self.training, self.fwd, self.dt, self.cout
        K = cfg['branches']
        # gram_contraction (conv1x1 768 -> 192 + BN) of the five heads reads the same x4: ONE GEMM with the five weight
        # matrices stacked along N; head k owns the column slice [k*g, (k+1)*g) of its output / statistics
        g_ = cfg['gram_dim']
        gc = self.gcon = dict(ld=K * g_)
        gc['W'] = self.buf('w.gram_contraction.all', (K * g_, cout))
        gc['WT'] = self.buf('wT.gram_contraction.all', (cout, K * g_)) if T else None
        gc['b'] = self.buf('w.gram_contraction.ball', (K * g_,), torch.float32)
        gc['s'], gc['q'] = self._bn_pool(K * g_), self._bn_pool(K * g_)
        for k in range(K):
            pre = f'gram_contraction.{k}.'
            self.prep.weight_prep(self.P[pre + '0.weight'], 1, g_, cout, 1, 1, dt, out=gc['W'][k * g_:], ldo=cout,
                                  outT=gc['WT'][:, k * g_:] if T else None, ldt=K * g_ if T else 0, t_cols=g_,
                                  label='prep.' + pre + 'w')
            self.prep.bias_fold(None, self.P[pre + '0.bias'], None, None, gc['b'][k * g_:], g_, cout)
        gc['out'] = self.act('gram_contraction.all.out', (M4, K * g_))
        F.gemm(x4, gc['W'], gc['out'], M4, K * g_, cout, dt, bias=gc['b'], colsum=gc['s'] if T else None,
               colsumsq=gc['q'] if T else None, label='gram_contraction.all')

    @staticmethod
    def _last_desc(plan):
        # the ctypes descriptor of the most recently recorded gemm/wgrad call
        for obj in reversed(plan.keep):
            if hasattr(obj, '_fields_'):
                return obj
        raise RuntimeError('no descriptor')

    # ------------------------------------------------------------------------------------------
    # effective weights (recorded into self.prep)
    # ------------------------------------------------------------------------------------------
    def _w_plain(self, name, Co, Ci, KH, KW, stem=False, need_T=True, flip=False, groups=1, rs=None, cs=None,
                 row_perm=None, ldo=None, key=None, src=None):
        """effective copy (and transposed copy when training) of a conv/linear weight; returns the forward copy"""
        key = key or name
        if key in self.W:
            return self.W[key]
        KK = Ci * KH * KW
        ldo = ldo or pad8(KK)
        out = self.buf('w.' + key, (groups * Co, ldo))
        outT = None
        ldt = 0
        if need_T and self.training:
            if flip:
                ldt = pad8(KH * KW * Co)
                outT = self.buf('wT.' + key, (groups * Ci, ldt))
            else:
                ldt = pad8(Co)
                outT = self.buf('wT.' + key, (groups * KK, ldt))
            self.W[key + '.T'] = outT
        self.prep.weight_prep(self.P[name] if src is None else src, groups, Co, Ci, KH, KW, self.dt, out=out, ldo=ldo, outT=outT, ldt=ldt, rs=rs,
                              cs=cs, flip=flip, stem=stem, row_perm=row_perm, label='prep.' + key)
        self.W[key] = out
        return out

    def _block_weights(self, pre, C):
        if pre + 'w49' in self.W:
            return
        P = self.P
        w49 = self.buf('w.' + pre + 'w49', (49, C), torch.float32)
        self.prep.transpose_f32(P[pre + self.NAMES['dw'] + 'weight'], w49, C, 49, label='prep.' + pre + 'w49')
        self.W[pre + 'w49'] = w49
        self._w_plain(pre + self.NAMES['fc1'] + 'weight', 4 * C, C, 1, 1, cs=P[pre + 'norm.weight'])
        b1e = self.buf('w.' + pre + 'b1e', (4 * C,), torch.float32)
        self.prep.bias_fold(P[pre + self.NAMES['fc1'] + 'weight'], P[pre + self.NAMES['fc1'] + 'bias'], None, P[pre + 'norm.bias'], b1e, 4 * C, C)
        self.W[pre + 'b1e'] = b1e
        self._w_plain(pre + self.NAMES['fc2'] + 'weight', C, 4 * C, 1, 1, rs=P[pre + 'gamma'])
        b2e = self.buf('w.' + pre + 'b2e', (C,), torch.float32)
        self.prep.bias_fold(None, P[pre + self.NAMES['fc2'] + 'bias'], P[pre + 'gamma'], None, b2e, C, 4 * C)
        self.W[pre + 'b2e'] = b2e

    # ------------------------------------------------------------------------------------------
    # ConvNeXt block
    # ------------------------------------------------------------------------------------------
    def _block_fwd(self, pre, x, res, C):
        assert C % 8 == 0
        F, dt, B = self.fwd, self.dt, self.B
        M = B * res * res
        self._block_weights(pre, C)
        W = self.W
        u = self.tmp('u', (M, C))
        xn = self.blk_act(pre + 'xn', (M, C))
        rstd = self.blk_act(pre + 'rstd', (M,), torch.float32)
        # fc1 stores a = gelu(h) and, when training, g = gelu'(h): backward never re-evaluates erf, and neither the
        # fc2 operand loader nor the wgrad loader has to (they used to, once per N tile)
        fused = self._mlp_fused(C)
        a = None if fused else self.blk_act(pre + 'a', (M, 4 * C))
        g = self.buf(pre + 'g', (M, 4 * C)) if (self.training and not fused) else None
        y = self.buf(pre + 'y', (M, C))
        dp = self.dp_scale.get(pre)
        cur = F.lane      # inside a head's lane: one chain in that lane
        for lane, r0, r1, b0, b1 in (self._fsplits(res * res) if cur == 0 else [(cur, 0, M, 0, B)]):
            F.lane = lane
            F.dwconv7_fwd(x[r0:r1], W[pre + 'w49'], self.P[pre + self.NAMES['dw'] + 'bias'], u[r0:r1], b1 - b0, res, res, C, dt,
                          label=pre + 'dw')
            F.layernorm_fwd(u[r0:r1], None, None, xn[r0:r1], None, rstd[r0:r1], r1 - r0, C, 1e-6, dt, label=pre + 'ln')
            if fused:     # fc1 -> GELU -> fc2 in one kernel: the [M, 4C] hidden activation never reaches HBM (csrc/mlp.hip)
                F.mlp_fwd(xn[r0:r1], W[pre + self.NAMES['fc1'] + 'weight'], W[pre + 'b1e'], W[pre + self.NAMES['fc2'] + 'weight'],
                          W[pre + 'b2e'], y[r0:r1], r1 - r0, C, dt, R=x[r0:r1], rowscale=dp[b0:b1] if dp is not None else None,
                          rows_per_scale=res * res, label=pre + 'mlp')
                continue
            F.gemm(xn[r0:r1], W[pre + self.NAMES['fc1'] + 'weight'], a[r0:r1], r1 - r0, 4 * C, C, dt, bias=W[pre + 'b1e'], act=ACT_GELU,
                   C2=g[r0:r1] if g is not None else None, c2_mode=2 if g is not None else 0, label=pre + 'fc1')
            F.gemm(a[r0:r1], W[pre + self.NAMES['fc2'] + 'weight'], y[r0:r1], r1 - r0, C, 4 * C, dt, bias=W[pre + 'b2e'],
                   rowscale=dp[b0:b1] if dp is not None else None, rows_per_scale=res * res, R=x[r0:r1], ldr=C,
                   label=pre + 'fc2')
        F.lane = cur
        self.blocks[pre] = dict(x=x, xn=xn, rstd=rstd, a=a, g=g, y=y, res=res, C=C, fused=fused)
        return y

    def _mlp_fused(self, C):
        """the fused MLP bodies (ga_mlp_fwd / ga_mlp_bwd) replace the fc1 / fc2 / dgrad2 / dgrad1 launches where they win: bf16 and
        the channel counts in GA_FUSED_MLP (default 96: stage 0, where the unfused GEMMs run at the HBM rate; measured at
        B = 256: forward 0.66 -> 0.27 ms, backward pair 0.45 -> 0.54 ms per block; at C = 192 the backward loses more than the
        forward gains)"""
        allowed = [int(v) for v in os.environ.get('GA_FUSED_MLP', '96').split(',') if v.strip()]
        if not self.training and 'GA_FUSED_MLP' not in os.environ:
            allowed.append(192)          # forward only: the fused body wins at C = 192 too (0.36 -> 0.29 ms per block)
        return C in allowed and ops.mlp_supported(C, 4 * C, self.dt)

    def _block_bwd(self, pre, dy, dx, next_pre=None):
        """dy: grad wrt the block output; writes dx (a different buffer) = grad wrt the block input.  next_pre: the block
        whose backward follows and takes dx as its dy unchanged -- its DropPath-scaled copy is written here, by the
        depthwise backward-data kernel that produces dx, instead of by a separate pass"""
        Bk, dt, B, P, W = self.bwd, self.dt, self.B, self.P, self.W
        b = self.blocks[pre]
        res, C = b['res'], b['C']
        M = B * res * res
        dp = self.dp_scale.get(pre)
        dyz = dy
        par, dz_tag = '', 'dyz'
        if self.async_wgrad and Bk.lane == 0:
            # two sets of the transients the weight-gradient launches read (dh, du; three of dyz; the caller rotates
            # three dx buffers): this block only has to wait for the asynchronous launches of the block before the
            # previous one
            self._bwd_seq += 1
            par = str(self._bwd_seq & 1)
            dz_tag = f'dyz{self._bwd_seq % 3}'
            Bk.join_async(f'blk{self._bwd_seq - 2}')
        if dp is not None:
            if pre in self._pre_dyz:
                dyz = self._pre_dyz.pop(pre)          # written by the previous block's depthwise backward
            else:
                dyz = self.tmp(dz_tag, (M, C))
                Bk.rowscale(dy, dp, dyz, M * C, res * res * C, dt, label=pre + 'dp')
        # the weight-gradient launches read only what the dgrad chain has already produced and nothing on the chain
        # reads their results: on the trunk they go to the plan's asynchronous lane and fill the tails of the chain's
        # launches (the next block joins before it overwrites dyz / dh / du)
        side = self.async_wgrad and Bk.lane == 0
        wl = ASYNC_LANE if side else Bk.lane
        ml = Bk.lane
        G2, gb2 = self.gbuf((C, 4 * C)), self.gbuf((C,))
        dh = self.tmp('dh' + par, (M, 4 * C))
        gb1 = self.gbuf((4 * C,))
        G1 = self.gbuf((4 * C, C))
        g = self.tmp('g', (M, C))
        if b['fused']:
            # hidden pre-activation re-computed; a / dh written once for the two weight-gradient GEMMs, dh stays on chip for dgrad1
            a = self.tmp('a' + par, (M, 4 * C))
            Bk.mlp_bwd(b['xn'], dyz, W[pre + self.NAMES['fc1'] + 'weight'], W[pre + 'b1e'], W[pre + self.NAMES['fc2'] + 'weight.T'],
                       W[pre + self.NAMES['fc1'] + 'weight.T'], a, dh, g, M, C, dt, label=pre + 'mlpb')
            Bk.lane = wl
            Bk.wgrad(dyz, a, G2, M, C, 4 * C, dt, dbias=gb2, label=pre + 'wg2')
            Bk.wgrad(dh, b['xn'], G1, M, 4 * C, C, dt, dbias=gb1, label=pre + 'wg1')
            Bk.lane = ml
        else:
            Bk.lane = wl
            Bk.wgrad(dyz, b['a'], G2, M, C, 4 * C, dt, dbias=gb2, label=pre + 'wg2')
            Bk.lane = ml
            Bk.gemm(dyz, W[pre + self.NAMES['fc2'] + 'weight.T'], dh, M, 4 * C, C, dt, H=b['g'], ldh=4 * C, h_is_deriv=True, colsum=gb1,
                    label=pre + 'dg2')
            Bk.lane = wl
            Bk.wgrad(dh, b['xn'], G1, M, 4 * C, C, dt, label=pre + 'wg1')
            Bk.lane = ml
            Bk.gemm(dh, W[pre + self.NAMES['fc1'] + 'weight.T'], g, M, C, 4 * C, dt, label=pre + 'dg1')
        du = self.tmp('du' + par, (M, C))
        Bk.layernorm_bwd(g, b['xn'], None, b['rstd'], None, None, du, None, None, M, C, True, dt, label=pre + 'lnb')
        dw49 = self.gbuf((49, C))
        Bk.lane = wl
        Bk.dwconv7_bwd_weight(du, b['x'], dw49, self.grad(pre + self.NAMES['dw'] + 'bias'), B, res, res, C, dt, label=pre + 'dww')
        Bk.lane = ml
        if side:
            Bk.async_mark(f'blk{self._bwd_seq}')
        dx2 = dp_next = None
        if next_pre is not None and self.fuse_dp:
            dp_next = self.dp_scale.get(next_pre)
            if dp_next is not None:
                nxt = f'dyz{(self._bwd_seq + 1) % 3}' if side else 'dyz'
                dx2 = self._pre_dyz[next_pre] = self.tmp(nxt, (M, C))
        Bk.dwconv7_bwd_data(du, W[pre + 'w49'], dy, dx, B, res, res, C, dt, dx2=dx2, scale2=dp_next, label=pre + 'dwd')
        Bk.weight_unfold(G2, 4 * C, C, 4 * C, gb=gb2, W=P[pre + self.NAMES['fc2'] + 'weight'], b=P[pre + self.NAMES['fc2'] + 'bias'],
                         rs=P[pre + 'gamma'], dW=self.grad(pre + self.NAMES['fc2'] + 'weight'), db=self.grad(pre + self.NAMES['fc2'] + 'bias'),
                         d_rs=self.grad(pre + 'gamma'), label=pre + 'unf2')
        Bk.weight_unfold(G1, C, 4 * C, C, gb=gb1, W=P[pre + self.NAMES['fc1'] + 'weight'], b=P[pre + self.NAMES['fc1'] + 'bias'],
                         cs=P[pre + 'norm.weight'], v=P[pre + 'norm.bias'], dW=self.grad(pre + self.NAMES['fc1'] + 'weight'),
                         db=self.grad(pre + self.NAMES['fc1'] + 'bias'), d_cs=self.grad(pre + 'norm.weight'),
                         d_v=self.grad(pre + 'norm.bias'), label=pre + 'unf1')
        Bk.transpose_f32(dw49, self.grad(pre + self.NAMES['dw'] + 'weight'), 49, C, accumulate=True, label=pre + 'unfdw')

    # ------------------------------------------------------------------------------------------
    # BatchNorm helper (stats come from the producing GEMM's colsum epilogue)
    # ------------------------------------------------------------------------------------------
    def _bn_pool(self, C):
        off = (self.bn_pool_off + 63) // 64 * 64
        assert off + C <= self.bn_pool.numel()
        self.bn_pool_off = off + C
        return self.bn_pool[off:off + C]

    def _bn_bufs(self, pre, C, zero=False):
        return dict(s=self._bn_pool(C), q=self._bn_pool(C),
                    mean=self.buf(pre + 'bmean', (C,), torch.float32, zero=zero), rstd=self.buf(pre + 'brstd', (C,), torch.float32, zero=zero),
                    scale=self.buf(pre + 'scale', (C,), torch.float32, zero=zero), shift=self.buf(pre + 'shift', (C,), torch.float32, zero=zero))

    def _sync_allreduce(self, plan, t, label):
        """SyncBatchNorm (GA/train.py:449-455, --sync-bn): sum a small fp32 statistics vector over the ranks, enqueued on the lane the
        plan call runs on (ga_allreduce_bucket through the communicator convert_sync_batchnorm() attached to the model)"""
        c = self.sync_bn
        plan._add('ga_allreduce_bucket', (c.handle, ops._ptr(t), t.numel(), GA_F32, 1.0, None, 0), label, keep=(t, c))

    def _bn_finalize(self, pre, bn, n, C):
        if self.sync_bn is not None and self.training:       # batch statistics over the GLOBAL batch: sums of all ranks, n x world
            self._sync_allreduce(self.fwd, bn['s'], pre + 'sync.s')
            self._sync_allreduce(self.fwd, bn['q'], pre + 'sync.q')
            n = n * self.sync_bn.world
        self.fwd.bn_finalize(bn['s'], bn['q'], n, self.P[pre + 'weight'], self.P[pre + 'bias'], 1e-5, 0.1,
                             self.Bf[pre + 'running_mean'], self.Bf[pre + 'running_var'], bn['mean'], bn['rstd'],
                             bn['scale'], bn['shift'], C, self.training, label=pre + 'fin')

    def _bn_bwd(self, pre, bn, dy, y_relu, x, dx, rows, C, rowscale=None, rps=1, ldx=0, lddx=0, weight=None, c_real=None):
        """weight / c_real: the zero-padded copy of the BatchNorm weight and the real channel count of a padded branch"""
        Bk = self.bwd
        s1, s2 = self.gbuf((C,)), self.gbuf((C,))
        Bk.bn_bwd_reduce(dy, y_relu, x, bn['mean'], bn['rstd'], s1, s2, rows, C, self.dt, rowscale=rowscale,
                         rows_per_scale=rps, ldx=ldx, label=pre + 'bnr')
        n = rows
        if self.sync_bn is not None:
            # torch.nn.SyncBatchNorm's backward: the parameter gradients take the LOCAL column sums (the gradient all-reduce averages
            # them later), so they are accumulated right here -- not deferred to the stage flush --; then the two sums that enter dx
            # are summed over the ranks in place
            Bk._add('ga_axpy_f32', (ops._ptr(self.grad(pre + 'weight')), ops._ptr(s2), 1.0, c_real or C), pre + 'dgamma', keep=(s2,))
            Bk._add('ga_axpy_f32', (ops._ptr(self.grad(pre + 'bias')), ops._ptr(s1), 1.0, c_real or C), pre + 'dbeta', keep=(s1,))
            self._sync_allreduce(Bk, s1, pre + 'sync.s1')
            self._sync_allreduce(Bk, s2, pre + 'sync.s2')
            n = rows * self.sync_bn.world
        Bk.bn_bwd_apply(dy, y_relu, x, bn['mean'], bn['rstd'], self.P[pre + 'weight'] if weight is None else weight, s1, s2, n, dx,
                        rows, C, self.dt, rowscale=rowscale, rows_per_scale=rps, ldx=ldx, lddx=lddx, label=pre + 'bna')
        if self.sync_bn is None:
            Bk.axpy_f32(self.grad(pre + 'weight'), s2, 1.0, c_real or C)
            Bk.axpy_f32(self.grad(pre + 'bias'), s1, 1.0, c_real or C)

    # ------------------------------------------------------------------------------------------
    # Bottleneck (ga_convnext.py:294-318)
    # ------------------------------------------------------------------------------------------
    def _bott_pad(self, pre, w, wp, ctot, cout):
        """odd-width variants (688 / 976: the Bottleneck is 172 / 244 channels wide, rows of 344 / 488 bytes): the branch runs
        wp = pad8(w) channels wide on zero-padded COPIES of its parameters (ga_pad_copy_f32 in the weight-prep plan); the extra
        channels carry exact zeros through conv / BN / ReLU / SE, their gradients are zero, and the real part of every
        gradient is copied back after the backward pass.  Returns (params, pending gradient copies)."""
        P = self.P
        R = P[pre + 'se.fc1.weight'].shape[0]
        spec = {'conv1.weight': ((wp, ctot), w, ctot, ctot, ctot), 'conv2.weight': ((wp, wp, 3, 3), w, 9 * w, 9 * w, 9 * wp),
                'conv3.weight': ((cout, wp), cout, w, w, wp), 'se.fc1.weight': ((R, wp), R, w, w, wp),
                'se.fc2.weight': ((wp, R), w, R, R, R), 'se.fc2.bias': ((wp,), 1, w, w, wp)}
        for bn in ('bn1', 'bn2'):
            spec[bn + '.weight'] = ((wp,), 1, w, w, wp)
            spec[bn + '.bias'] = ((wp,), 1, w, w, wp)
        PB, back = {}, []
        for name, (shape, rows, cols, lds, ldd) in spec.items():
            # (conv2: a source row is [ci][9], contiguous; the padded row [wp][9] holds it at its head)
            buf = self.buf('pad.' + pre + name, shape, torch.float32, zero=True)
            self.prep.pad_copy_f32(P[pre + name], buf, rows, cols, lds, ldd, label='prep.pad.' + pre + name)
            PB[pre + name] = buf
            back.append((name, rows, cols, lds, ldd))
        return PB, back

    def _bottleneck_fwd(self, cat, M4, ctot, cout):
        F, dt, B, P, T = self.fwd, self.dt, self.B, self.P, self.training
        pre = self.bott_prefix
        w_real = cout // 4
        w = pad8(w_real)
        HW = M4 // B
        st = self.bott = dict(cat=cat, w=w, w_real=w_real, cout=cout, ctot=ctot, PB=None)
        if w != w_real:
            st['PB'], st['back'] = self._bott_pad(pre, w_real, w, ctot, cout)
            P = dict(P)                      # the padded copies shadow the Bottleneck's odd-width parameters
            P.update(st['PB'])
        st['P'] = P
        stats = T  # batch statistics only in train mode

        def conv_bn(name, bnname, A, Wt, N, Kdim, bias=None, n_real=None, **kw):
            c = self.act(pre + name + '.out', (M4, N))
            bn = self._bn_bufs(pre + bnname + '.', N, zero=bool(n_real) and n_real != N)
            F.gemm(A, Wt, c, M4, N, Kdim, dt, bias=bias, colsum=bn['s'] if stats else None,
                   colsumsq=bn['q'] if stats else None, label=pre + name, **kw)
            # (padded width: the statistics buffers are N wide, zero beyond the real channels -> scale = shift = 0 there)
            self._bn_finalize(pre + bnname + '.', bn, M4, n_real or N)
            return c, bn

        # the shortcut branch (conv 1x1 of the 2208-channel concat + BN) only needs `cat`: it runs on the plan's
        # asynchronous lane beside the main branch and is joined before the sum
        F.lane = ASYNC_LANE if self.par_branch else 0
        Wds = self._w_plain(pre + 'downsample.0.weight', cout, ctot, 1, 1)
        st['sc'], st['bnd'] = conv_bn('downsample.0', 'downsample.1', cat, Wds, cout, ctot, bias=P[pre + 'downsample.0.bias'])
        t = self.tmp('bott.t', (M4, cout))
        F.affine_act(st['sc'], st['bnd']['scale'], st['bnd']['shift'], None, t, M4, cout, False, dt, label=pre + 'bnd')
        F.lane = 0
        Wc1 = self._w_plain(pre + 'conv1.weight', w, ctot, 1, 1, src=P[pre + 'conv1.weight'])
        st['c1'], st['bn1'] = conv_bn('conv1', 'bn1', cat, Wc1, w, ctot, n_real=w_real)
        st['y1'] = self.act(pre + 'y1', (M4, w))
        F.affine_act(st['c1'], st['bn1']['scale'], st['bn1']['shift'], None, st['y1'], M4, w, True, dt, label=pre + 'bn1')
        Wc2 = self._w_plain(pre + 'conv2.weight', w, w, 3, 3, flip=True, src=P[pre + 'conv2.weight'])
        st['c2'], st['bn2'] = conv_bn('conv2', 'bn2', st['y1'], Wc2, w, 9 * w, n_real=w_real, a_kind=A_CONV3, a_dims=(14, 14, w))
        st['y2'] = self.act(pre + 'y2', (M4, w))
        F.affine_act(st['c2'], st['bn2']['scale'], st['bn2']['shift'], None, st['y2'], M4, w, True, dt, label=pre + 'bn2')
        # squeeze-excite
        R = P[pre + 'se.fc1.weight'].shape[0]
        st['R'] = R
        st['sp'] = self.act(pre + 'se.sp', (B, w), torch.float32)
        st['hid'] = self.act(pre + 'se.hid', (B, R), torch.float32)
        st['gate'] = self.act(pre + 'se.gate', (B, w), torch.float32)
        F.spatial_sum(st['y2'], None, st['sp'], B, HW, w, 1.0 / HW, dt, label=pre + 'se.pool')
        F.se_mlp_fwd(st['sp'], P[pre + 'se.fc1.weight'], P[pre + 'se.fc1.bias'], P[pre + 'se.fc2.weight'],
                     P[pre + 'se.fc2.bias'], st['hid'], st['gate'], B, w, R, label=pre + 'se.mlp')
        st['z'] = self.act(pre + 'se.z', (M4, w))
        F.chan_scale(st['y2'], st['gate'], None, st['z'], B, HW, w, dt, label=pre + 'se.scale')
        Wc3 = self._w_plain(pre + 'conv3.weight', cout, w, 1, 1, src=P[pre + 'conv3.weight'])
        st['c3'], st['bn3'] = conv_bn('conv3', 'bn3', st['z'], Wc3, cout, w)
        x4 = self.buf(pre + 'out', (M4, cout))
        if self.par_branch:
            F.join_async()
        F.affine_act(st['c3'], st['bn3']['scale'], st['bn3']['shift'], t, x4, M4, cout, True, dt,
                     rowscale=self.dp_scale.get(pre), rows_per_scale=HW, label=pre + 'bn3+add')
        st['x4'] = x4
        return x4

    def _bottleneck_bwd(self, dx4, dcat):
        Bk, dt, B, P, W = self.bwd, self.dt, self.B, self.bott['P'], self.W
        pre = self.bott_prefix
        st = self.bott
        w, cout, ctot = st['w'], st['cout'], st['ctot']
        padded = st['PB'] is not None
        GB = {pre + name: self.gbuf(tuple(st['PB'][pre + name].shape)) for name, *_ in st['back']} if padded else {}

        def grad(name):          # gradient buffer of a Bottleneck parameter: the zeroed padded one where the branch is padded
            return GB[name] if name in GB else self.grad(name)

        M4 = dx4.shape[0]
        HW = M4 // B
        dp = self.dp_scale.get(pre)
        dc3 = self.tmp('bott.dc3', (M4, cout))
        self._bn_bwd(pre + 'bn3.', st['bn3'], dx4, st['x4'], st['c3'], dc3, M4, cout, rowscale=dp, rps=HW)
        # shortcut branch on the asynchronous lane: BN backward, weight gradient and the FIRST write of dcat
        dsc = self.tmp('bott.dsc', (M4, cout))
        with self._wlane():
            self._bn_bwd(pre + 'downsample.1.', st['bnd'], dx4, st['x4'], st['sc'], dsc, M4, cout)
            Bk.wgrad(dsc, st['cat'], self.grad(pre + 'downsample.0.weight'), M4, cout, ctot, dt,
                     dbias=self.grad(pre + 'downsample.0.bias'), label=pre + 'ds.wg')
            Bk.gemm(dsc, W[pre + 'downsample.0.weight.T'], dcat, M4, ctot, cout, dt, ldb=pad8(cout), label=pre + 'ds.dg')
        # conv3
        with self._wlane():
            Bk.wgrad(dc3, st['z'], grad(pre + 'conv3.weight'), M4, cout, w, dt, label=pre + 'conv3.wg')
        dz = self.tmp('bott.dz', (M4, w))
        Bk.gemm(dc3, W[pre + 'conv3.weight.T'], dz, M4, w, cout, dt, label=pre + 'conv3.dg')
        # squeeze-excite
        dgate = self.tmp('bott.dgate', (B, w), torch.float32)
        dsp = self.tmp('bott.dsp', (B, w), torch.float32)
        Bk.spatial_sum(dz, st['y2'], dgate, B, HW, w, 1.0, dt, label=pre + 'se.dgate')
        Bk.se_mlp_bwd(dgate, st['gate'], st['hid'], st['sp'], P[pre + 'se.fc1.weight'], P[pre + 'se.fc2.weight'], dsp,
                      grad(pre + 'se.fc1.weight'), grad(pre + 'se.fc1.bias'), grad(pre + 'se.fc2.weight'),
                      grad(pre + 'se.fc2.bias'), B, w, st['R'], ds_scale=1.0 / HW, label=pre + 'se.mlpb')
        dy2 = self.tmp('bott.dy2', (M4, w))
        Bk.chan_scale(dz, st['gate'], dsp, dy2, B, HW, w, dt, label=pre + 'se.back')
        dc2 = self.tmp('bott.dc2', (M4, w))
        self._bn_bwd(pre + 'bn2.', st['bn2'], dy2, st['y2'], st['c2'], dc2, M4, w, weight=P[pre + 'bn2.weight'] if padded else None,
                     c_real=st['w_real'])
        # conv2 3x3
        G = self.gbuf((w, 9 * w))
        with self._wlane():
            Bk.wgrad(dc2, st['y1'], G, M4, w, 9 * w, dt, x_kind=A_CONV3, x_dims=(14, 14, w), label=pre + 'conv2.wg')
        Bk.weight_unfold(G, 9 * w, w, w, 3, 3, dW=grad(pre + 'conv2.weight'), label=pre + 'conv2.unf')
        dy1 = self.tmp('bott.dy1', (M4, w))
        Bk.gemm(dc2, W[pre + 'conv2.weight.T'], dy1, M4, w, 9 * w, dt, a_kind=A_CONV3, a_dims=(14, 14, w),
                ldb=pad8(9 * w), label=pre + 'conv2.dg')
        dc1 = self.tmp('bott.dc1', (M4, w))
        self._bn_bwd(pre + 'bn1.', st['bn1'], dy1, st['y1'], st['c1'], dc1, M4, w, weight=P[pre + 'bn1.weight'] if padded else None,
                     c_real=st['w_real'])
        # conv1 and the shortcut conv both read `cat`; conv1's dgrad adds onto the shortcut's (already written) dcat
        with self._wlane():
            Bk.wgrad(dc1, st['cat'], grad(pre + 'conv1.weight'), M4, w, ctot, dt, label=pre + 'conv1.wg')
        if self.async_wgrad:
            Bk.join_async()
        Bk.gemm(dc1, W[pre + 'conv1.weight.T'], dcat, M4, ctot, w, dt, ldb=pad8(w), R=dcat, ldr=ctot, label=pre + 'conv1.dg')
        if padded:     # the real part of every padded gradient back into the parameter's gradient (the weight-gradient lane has joined)
            skip = ('bn1.weight', 'bn1.bias', 'bn2.weight', 'bn2.bias')       # accumulated by _bn_bwd at the real width
            Bk.flush(pre + 'pad.')      # the deferred weight-unfold job of conv2 writes the padded gradient: run it first
            for name, rows, cols, lds, ldd in st['back']:
                if name not in skip:
                    Bk.pad_copy_f32(GB[pre + name], self.grad(pre + name), rows, cols, ldd, lds, accumulate=True, label=pre + name + '.unpad')

    # ------------------------------------------------------------------------------------------
    # one GA head (ga_convnext.py:491-504)
    # ------------------------------------------------------------------------------------------
    # ------------------------------------------------------------------------------------------
    # GroupConvMlp (ga_convnext.py:190-222, ga_cswin.py:321-349): grouped fc1 -> GELU -> channel_shuffle -> grouped fc2
    # on `rows` tokens.  The shuffle is folded into the ROW ORDER of fc1's effective weights: hidden index
    # n = gi*gc + ci  <-  fc1 output channel ci*mg + gi; with Nv = gc / mg rows per "virtual group" the fc1 input group is
    # constant inside one (= v % mg), so fc1 is a batched GEMM over mg*mg virtual groups and fc2 one over mg groups.
    # ------------------------------------------------------------------------------------------
    def _gmlp_fwd(self, pre, t, rows, C, mg, out, R, rowscale, rps, gamma_name=None, act='gelu', drop_mask=None):
        """out = R + rowscale * gamma * fc2(shuffle(drop(act(fc1(t)))));  pre = '<block>.mlp.';  act 'gelu' (GA) or 'relu'
        (MAP, map.py:467) with an optional dropout mask [rows, 4C] on the hidden layer (in the SHUFFLED channel order)"""
        F, dt, P, T = self.fwd, self.dt, self.P, self.training
        gamma = self._pw(gamma_name) if gamma_name else None
        Hd = 4 * C
        gc_ = Hd // mg
        Nv = gc_ // mg          # rows per virtual group (fc1 input group is constant inside one)
        cin = C // mg
        n_idx = torch.arange(Hd)
        perm = ((n_idx % gc_) * mg + n_idx // gc_).to(torch.int32).to(self.dev)
        if (Nv % 8 or cin % 8) and self.pad_gmlp and C % mg == 0:
            # odd widths (688 / 4 = 172 channels per group) on the MFMA kernels: the layer runs as a GroupConvMlp of mg groups of
            # cin_p = pad8(cin) channels on zero-padded copies of its five parameters (fc1 rows in groups of 4 cin -> 4 cin_p and
            # columns cin -> cin_p; fc2 rows and columns in groups of cin -> cin_p; ga_pad_groups_f32) and group-padded copies of
            # its input / residual; padded hidden and output channels are exact zeros (zero weights, zero biases), the real part
            # of the output is compacted back and the real part of every padded gradient copied back after the backward pass
            cin_p = pad8(cin)
            Cp = mg * cin_p
            self._pad_groups(pre + 'fc1.weight', Hd, cin, 4 * cin, 4 * cin_p, cin, cin_p)
            self._pad_groups(pre + 'fc1.bias', 1, Hd, 1, 1, 4 * cin, 4 * cin_p)
            self._pad_groups(pre + 'fc2.weight', C, gc_, cin, cin_p, cin, cin_p)
            self._pad_groups(pre + 'fc2.bias', 1, C, 1, 1, cin, cin_p)
            if gamma_name:
                self._pad_groups(gamma_name, 1, C, 1, 1, cin, cin_p)
            # (persistent zero-initialised buffers: the pad columns are never written)
            tp = self.buf(pre + 't.pad', (rows, Cp), zero=True)
            F.pad_copy(t, tp, rows * mg, cin, cin, cin_p, dt, label=pre + 't.pad')
            Rp = None
            if R is not None:
                Rp = self.buf(pre + 'R.pad', (rows, Cp), zero=True)
                F.pad_copy(R, Rp, rows * mg, cin, cin, cin_p, dt, label=pre + 'R.pad')
            outp = self.tmp('gmlp.out.pad', (rows, Cp))
            st = self._gmlp_fwd(pre, tp, rows, Cp, mg, outp, Rp, rowscale, rps, gamma_name=gamma_name, act=act, drop_mask=drop_mask)
            F.pad_copy(outp, out, rows * mg, cin, cin_p, cin, dt, label=pre + 'out.unpad')
            st['pad'] = dict(tp=tp, Cp=Cp, cin=cin, cin_p=cin_p)
            return st
        if Nv % 8 or cin % 8:
            # the same layers on the alignment-free kernels (GAEXT_PAD_GMLP=0), fp32 master weights,
            # the channel_shuffle as a column map of fc2's input (hidden kept in the ORIGINAL channel order, pre-activation saved)
            assert act == 'gelu' and drop_mask is None and rows <= 4096, (pre, C, mg, rows)
            st = dict(naive=True, Hd=Hd, hpre=self.act(pre + 'hpre', (rows, Hd)), am=self.act(pre + 'am', (rows, Hd)),
                      yraw=self.act(pre + 'yraw', (rows, C)) if T else None)
            st['d1'] = F.small_linear_desc(t, P[pre + 'fc1.weight'], st['hpre'], rows, mg, gc_, cin, dt, lda=C, a_gstride=cin, ldy=Hd,
                                           bias=P[pre + 'fc1.bias'])
            F.small_linear_fwd(st['d1'], label=pre + 'fc1')
            F.gelu_fwd(st['hpre'], st['am'], rows * Hd, dt, label=pre + 'gelu')
            kw = dict(bias=P[pre + 'fc2.bias'], a_perm=perm, col_scale=gamma, Yraw=st['yraw'])
            d2 = F.small_linear_desc(st['am'], P[pre + 'fc2.weight'], out, rows, mg, cin, gc_, dt, lda=Hd, a_gstride=gc_, ldy=C,
                                     rowscale=rowscale, rows_per_scale=rps, R=R, ldr=C, **kw)
            F.small_linear_fwd(d2, label=pre + 'fc2')
            # the backward sees a gradient that already carries the DropPath scale: same layer without the row scale
            st['d2b'] = F.small_linear_desc(st['am'], P[pre + 'fc2.weight'], out, rows, mg, cin, gc_, dt, lda=Hd, a_gstride=gc_, ldy=C, **kw)
            return st
        assert gc_ % mg == 0, (pre, C, mg)
        Wm1 = self._w_plain(pre + 'fc1.weight', Nv, cin, 1, 1, groups=mg * mg, row_perm=perm, src=self._pw(pre + 'fc1.weight'))
        bm1 = self.buf('w.' + pre + 'bm1', (Hd,), torch.float32)
        self.prep.bias_fold(None, self._pw(pre + 'fc1.bias'), None, None, bm1, Hd, cin, row_perm=perm)
        st = dict(perm=perm, Hd=Hd, gc=gc_, Nv=Nv, cin=cin)
        st['am'] = self.act(pre + 'am', (rows, Hd))                      # gelu(hidden), shuffled order
        st['gm'] = self.act(pre + 'gm', (rows, Hd)) if T else None       # gelu'(hidden)
        if act == 'gelu':
            assert drop_mask is None
            F.gemm(t, Wm1, st['am'], rows, Nv, cin, dt, lda=C, batch=mg * mg, strideA=cin, a_batch_mod=mg,
                   strideB=Nv * pad8(cin), ldb=pad8(cin), ldc=Hd, strideC=Nv, bias=bm1, strideBias=Nv, act=ACT_GELU,
                   C2=st['gm'], c2_mode=2 if T else 0, label=pre + 'fc1')
        else:   # ReLU: its derivative (times the dropout mask) is a small pass of its own; the GEMM epilogue stays generic
            two = T or drop_mask is not None
            raw = self.tmp('am_raw', (rows, Hd)) if two else st['am']
            F.gemm(t, Wm1, raw, rows, Nv, cin, dt, lda=C, batch=mg * mg, strideA=cin, a_batch_mod=mg,
                   strideB=Nv * pad8(cin), ldb=pad8(cin), ldc=Hd, strideC=Nv, bias=bm1, strideBias=Nv, act=ops.ACT_RELU,
                   label=pre + 'fc1')
            if two:
                F.relu_drop(raw, drop_mask, st['am'], st['gm'], rows * Hd, dt, label=pre + 'relu')
        Wm2 = self._w_plain(pre + 'fc2.weight', cin, gc_, 1, 1, groups=mg, rs=gamma, src=self._pw(pre + 'fc2.weight'))
        bm2 = self.buf('w.' + pre + 'bm2', (C,), torch.float32)
        self.prep.bias_fold(None, self._pw(pre + 'fc2.bias'), gamma, None, bm2, C, gc_)
        F.gemm(st['am'], Wm2, out, rows, cin, gc_, dt, lda=Hd, batch=mg, strideA=gc_, strideB=cin * pad8(gc_),
               ldb=pad8(gc_), ldc=C, strideC=cin, bias=bm2, strideBias=cin, rowscale=rowscale,
               rows_per_scale=rps, R=R, ldr=C, strideR=cin, label=pre + 'fc2')
        return st

    def _gmlp_bwd(self, pre, st, dmz, t, rows, C, mg, dtk, gamma_name=None):
        """dmz: gradient wrt the MLP branch output (DropPath scale already applied) -> dtk = gradient wrt the input t"""
        Bk, dt, P, W = self.bwd, self.dt, self.P, self.W
        gamma = self._pw(gamma_name) if gamma_name else None
        if 'pad' in st:       # the padded layer (see _gmlp_fwd): group-padded gradient in, real part of the input gradient out
            pd = st.pop('pad')
            cin, cin_p, Cp = pd['cin'], pd['cin_p'], pd['Cp']
            dmzp = self.buf(pre + 'dmz.pad', (rows, Cp), zero=True)
            Bk.pad_copy(dmz, dmzp, rows * mg, cin, cin, cin_p, dt, label=pre + 'dmz.pad')
            dtkp = self.tmp('gmlp.dt.pad', (rows, Cp))
            self._gmlp_bwd(pre, st, dmzp, pd['tp'], rows, Cp, mg, dtkp, gamma_name=gamma_name)
            Bk.pad_copy(dtkp, dtk, rows * mg, cin, cin_p, cin, dt, label=pre + 'dt.unpad')
            st['pad'] = pd
            return
        if st.get('naive'):
            Hd = st['Hd']
            dam, dh = self.tmp('dam', (rows, Hd)), self.tmp('dhm', (rows, Hd))
            Bk.small_linear_bwd(st['d2b'], dmz, dA=dam, dW=self.grad(pre + 'fc2.weight'), dbias=self.grad(pre + 'fc2.bias'),
                                dcol_scale=self.grad(gamma_name) if gamma_name else None, label=pre + 'fc2.bwd')
            Bk.gelu_bwd(dam, st['hpre'], dh, rows * Hd, dt, label=pre + 'geluB')
            Bk.small_linear_bwd(st['d1'], dh, dA=dtk, dW=self.grad(pre + 'fc1.weight'), dbias=self.grad(pre + 'fc1.bias'),
                                label=pre + 'fc1.bwd')
            return
        Hd, gc_, Nv, cin = st['Hd'], st['gc'], st['Nv'], st['cin']
        # fc2 (mg groups)
        Gm2, gbm2 = self.gbuf((C, gc_)), self.gbuf((C,))
        Bk.wgrad(dmz, st['am'], Gm2, rows, cin, gc_, dt, ldy=C, ldx=Hd, ldw=gc_, batch=mg, strideY=cin, strideX=gc_,
                 strideW=cin * gc_, dbias=gbm2, strideDbias=cin, label=pre + 'fc2.wg')
        Bk.weight_unfold(Gm2, gc_, C, gc_, gb=gbm2, W=self._pw(pre + 'fc2.weight'), b=self._pw(pre + 'fc2.bias'),
                         rs=gamma, dW=self._pg(pre + 'fc2.weight'), db=self._pg(pre + 'fc2.bias'),
                         d_rs=self._pg(gamma_name) if gamma_name else None, label=pre + 'fc2.unf')
        dhm = self.tmp('dhm', (rows, Hd))
        gbm1 = self.gbuf((Hd,))
        Bk.gemm(dmz, W[pre + 'fc2.weight.T'], dhm, rows, gc_, cin, dt, lda=C, batch=mg, strideA=cin,
                strideB=gc_ * pad8(cin), ldb=pad8(cin), ldc=Hd, strideC=gc_, H=st['gm'], ldh=Hd, strideH=gc_, h_is_deriv=True,
                colsum=gbm1, strideCol=gc_, label=pre + 'fc2.dg')
        # fc1 (mg*mg virtual groups)
        Gm1 = self.gbuf((Hd, cin))
        Bk.wgrad(dhm, t, Gm1, rows, Nv, cin, dt, ldy=Hd, ldx=C, ldw=cin, batch=mg * mg, strideY=Nv, strideX=cin,
                 x_batch_mod=mg, strideW=Nv * cin, label=pre + 'fc1.wg')
        Bk.weight_unfold(Gm1, cin, Hd, cin, gb=gbm1, W=self._pw(pre + 'fc1.weight'), b=self._pw(pre + 'fc1.bias'),
                         row_perm=st['perm'], dW=self._pg(pre + 'fc1.weight'), db=self._pg(pre + 'fc1.bias'),
                         label=pre + 'fc1.unf')
        Wm1T = W[pre + 'fc1.weight.T']            # [mg*mg][cin][pad8(Nv)]
        for gi in range(mg):
            Bk.gemm(dhm[:, gi * gc_:], Wm1T[gi * mg * cin:], dtk, rows, cin, Nv, dt, lda=Hd, batch=mg, strideA=Nv,
                    strideB=cin * pad8(Nv), ldb=pad8(Nv), ldc=C, strideC=cin, R=dtk if gi > 0 else None, ldr=C,
                    strideR=cin, label=pre + f'fc1.dg{gi}')

    def _head_supported(self):
        cfg, cout = self.cfg, self.cout
        g, E, nh, mg, groups = cfg['gram_dim'], cfg['dim_embed'], cfg['num_heads'], cfg['mlp_groups'], cfg['gram_groups']
        # (cout / groups and cout / mg need NOT be multiples of 8: the 688 / 976 variants run those grouped one-token layers
        #  on the alignment-free ga_small_linear kernels)
        return E % nh == 0 and E % 8 == 0 and g % 8 == 0 and cout % 8 == 0 and cout % groups == 0 and cout % mg == 0

    def _head_fwd(self, k, x4, M4, cout, Hc):
        cfg = self.cfg
        if not self._head_supported():
            # ga_convnext_{tiny,small}_688 / base_976: 688 / 8 = 86 and 688 / 4 = 172 channels per group are not multiples
            # of 8, so the grouped operands of gram_embedding / GroupConvMlp are not 16-byte aligned per group -- and the
            # stage-4 Bottleneck is 688 / 4 = 172 channels wide (rows of 344 bytes).  Tried in round 1: padded group
            # layouts around the head GEMMs work, the 172-wide Bottleneck activations need padded leading dimensions in
            # the BatchNorm / 3x3-gather / squeeze-excite kernels as well (not built)
            raise NotImplementedError(
                f"GA head with dims[4]={cout}, gram groups={cfg['gram_groups']}, mlp groups={cfg['mlp_groups']}, "
                f"dim_embed={cfg['dim_embed']}: the HIP engine needs dims[4] divisible by {8 * cfg['gram_groups']} and "
                f"{8 * cfg['mlp_groups']} (per-group channel counts that are multiples of 8); the *_768 / *_1024 variants "
                f"qualify, *_688 / *_976 need padded group layouts (not built)")
        h = dict(k=k)
        self._head_contract_fwd(h, k, M4)
        h['g1'] = self._gram_layer_fwd(h, k, Hc)
        self._head_tail_fwd(h, k, x4, M4, cout, Hc)
        return h

    def _head_contract_fwd(self, h, k, M4):
        """gram_contraction[k]'s BatchNorm on this head's column slice of the stacked conv output -> h['g0'] [M4, gram_dim]"""
        F, dt, g = self.fwd, self.dt, self.cfg['gram_dim']
        # --- gram_contraction: the conv output / batch sums are column slices of the stacked GEMM; BN per head
        pre = f'gram_contraction.{k}.'
        gcn = self.gcon
        h['gc'] = gcn['out'][:, k * g:]                      # [M4, g] view, row stride gcn['ld']
        bn = dict(s=gcn['s'][k * g:(k + 1) * g], q=gcn['q'][k * g:(k + 1) * g],
                  mean=self.buf(pre + '1.bmean', (g,), torch.float32), rstd=self.buf(pre + '1.brstd', (g,), torch.float32),
                  scale=self.buf(pre + '1.scale', (g,), torch.float32), shift=self.buf(pre + '1.shift', (g,), torch.float32))
        h['bn_gc'] = bn
        self._bn_finalize(pre + '1.', bn, M4, g)
        h['g0'] = self.buf(pre + 'g0', (M4, g))
        F.affine_act(h['gc'], bn['scale'], bn['shift'], None, h['g0'], M4, g, False, dt, ldx=gcn['ld'], label=pre + 'bn')

    def _gram_layer_fwd(self, h, k, Hc):
        """gram_layer[k]: one ConvNeXt block at 14x14 (ga_convnext.py:411-415)"""
        h['blk'] = f'gram_layer.{k}.blocks.0.'
        return self._block_fwd(h['blk'], h['g0'], Hc, self.cfg['gram_dim'])

    def _gram_layer_bwd(self, h, dg1, dg0):
        self._block_bwd(h['blk'], dg1, dg0)

    def _head_tail_fwd(self, h, k, x4, M4, cout, Hc):
        """get_gram -> gram_embedding (+BN) -> LayerScaleBlockClassAttn -> this head's classifier input"""
        F, dt, B, P, T, cfg = self.fwd, self.dt, self.B, self.P, self.training, self.cfg
        g, E, nh, mg, NC = cfg['gram_dim'], cfg['dim_embed'], cfg['num_heads'], cfg['mlp_groups'], cfg['num_classes']
        groups = cfg['gram_groups']
        HW = Hc * Hc
        hd = E // nh
        h['scale'] = hd ** -0.5
        if self.shared_tok:       # padded head width (== the real one for the *_768 / *_1024 variants)
            E, hd = self.Ea, self.hd_p
        g1 = h['g1']
        # --- Gram vector (fp32 accumulate; the reference's fp64 branch for train & B<128 is covered by the 1e-3 gate)
        alpha = 1.0 / (Hc * Hc * HW)
        h['alpha'] = alpha
        G = self.tmp('gramG', (B, g, g), torch.float32)
        F.wgrad(g1, g1, G, HW, g, g, dt, batch=B, strideY=HW * g, strideX=HW * g, strideW=g * g, split_m=1,
                accumulate=False, alpha=alpha, label=f'gram.{k}')
        ntri = g * (g + 1) // 2
        Kg = ntri // groups
        Kp = pad8(Kg)
        h['Kg'], h['Kp'] = Kg, Kp
        h['vec'] = self.act(f'gram.{k}.vec', (B, groups * Kp))
        h['inv'] = self.act(f'gram.{k}.inv', (B,), torch.float32)
        F.gram_pack_fwd(G, h['vec'], h['inv'], B, g, groups, Kp, dt, label=f'gram.{k}.pack')
        # --- gram_embedding: grouped 1x1 + BN on (B, cout)
        pre = f'gram_embedding.{k}.'
        cg = cout // groups
        h['e'] = self.act(pre + 'out', (B, cout))
        h['bn_e'] = self._bn_bufs(pre + '1.', cout)
        h['emb_small'] = None
        if cg % 8:      # 688 / 8 = 86 (976 / 8 = 122) channels per group: off the 16-byte grid
            # backward: alignment-free form on the master weights.  Forward (the expensive product: K = 2316 per group): the
            # batched MFMA GEMM into a group-padded buffer [B][groups][pad8(cg)] -- aligned group bases, ragged N -- then compacted
            h['emb_small'] = F.small_linear_desc(h['vec'], P[pre + '0.weight'], h['e'], B, groups, cg, Kg, dt, lda=groups * Kp,
                                                 a_gstride=Kp, ldy=cout, bias=P[pre + '0.bias'])
            cgp = pad8(cg)
            ep = self.tmp('emb_pad', (B, groups * cgp))
            Wemb = self._w_plain(pre + '0.weight', cg, Kg, 1, 1, groups=groups, ldo=Kp, need_T=self.pad_gmlp)
            F.gemm(h['vec'], Wemb, ep, B, cg, Kp, dt, lda=groups * Kp, batch=groups, strideA=Kp, strideB=cg * Kp,
                   ldc=groups * cgp, strideC=cgp, bias=P[pre + '0.bias'], strideBias=cg, label=pre + 'conv')
            F.pad_copy(ep, h['e'], B * groups, cg, cgp, cg, dt, label=pre + 'compact')
            if T:
                F.colstats(h['e'], cout, B, cout, h['bn_e']['s'], h['bn_e']['q'], dt, label=pre + 'stats')
        else:
            Wemb = self._w_plain(pre + '0.weight', cg, Kg, 1, 1, groups=groups, ldo=Kp)
            F.gemm(h['vec'], Wemb, h['e'], B, cg, Kp, dt, lda=groups * Kp, batch=groups, strideA=Kp, strideB=cg * Kp,
                   ldc=cout, strideC=cg, bias=P[pre + '0.bias'], strideBias=cg, colsum=h['bn_e']['s'] if T else None,
                   colsumsq=h['bn_e']['q'] if T else None, strideCol=cg, label=pre + 'conv')
        self._bn_finalize(pre + '1.', h['bn_e'], B, cout)
        h['cls0'] = self.buf(pre + 'cls0', (B, cout))
        F.affine_act(h['e'], h['bn_e']['scale'], h['bn_e']['shift'], None, h['cls0'], B, cout, False, dt, label=pre + 'bn')
        # --- class-attention block
        pre = f'ga.{k}.'
        N = HW
        pk, pv = P[pre + 'attn.k.weight'], P[pre + 'attn.v.weight']
        assert pv.data_ptr() == pk.data_ptr() + pk.numel() * 4, 'k/v weights must be adjacent in the flat buffer'
        h['ao'] = self.act(pre + 'ao', (B, E))
        h['P'] = self.act(pre + 'P', (B, nh, N + 1), torch.float32)
        if self.shared_tok:
            # class-token row normalised on its own; norm1's affine part folded into the k | v and q operands
            g1, b1 = P[pre + 'norm1.weight'], P[pre + 'norm1.bias']
            h['cn'] = self.act(pre + 'cn', (B, cout))
            h['rc'] = self.act(pre + 'rc', (B,), torch.float32)
            F.layernorm_fwd(h['cls0'], None, None, h['cn'], None, h['rc'], B, cout, 1e-5, dt, label=pre + 'ln1c')
            tk = self.tok
            E2 = tk['E2']
            h['kvt'] = tk['kv'][:, k * E2:]                       # column slice, row stride tk['ld']
            h['kvc'] = self.act(pre + 'kvc', (B, E2))
            F.gemm(h['cn'], tk['W'][k * E2:], h['kvc'], B, E2, cout, dt, bias=tk['b'][k * E2:], label=pre + 'kvc')
            Wq = self._w_plain(pre + 'attn.q.weight', E, cout, 1, 1, cs=g1, src=self._pw(pre + 'attn.q.weight'))
            bq = self.buf('w.' + pre + 'bq', (E,), torch.float32)
            self.prep.bias_fold(self._pw(pre + 'attn.q.weight'), None, None, b1, bq, E, cout)
            h['q'] = self.act(pre + 'q', (B, E))
            F.gemm(h['cn'], Wq, h['q'], B, E, cout, dt, bias=bq, label=pre + 'q')
            F.class_attn_fwd2(h['q'], h['kvc'], h['kvt'], h['ao'], h['P'], B, N + 1, nh, hd, h['scale'], dt, tok_ld=tk['ld'],
                              label=pre + 'attn')
        else:
            h['u'] = self.act(pre + 'u', (B * (N + 1), cout))
            F.token_cat(h['cls0'], x4, h['u'], B, N, cout, dt, label=pre + 'cat')
            h['un'] = self.act(pre + 'un', (B * (N + 1), cout))
            h['m1'] = self.act(pre + 'm1', (B * (N + 1),), torch.float32)
            h['r1'] = self.act(pre + 'r1', (B * (N + 1),), torch.float32)
            F.layernorm_fwd(h['u'], P[pre + 'norm1.weight'], P[pre + 'norm1.bias'], h['un'], h['m1'], h['r1'], B * (N + 1),
                            cout, 1e-5, dt, label=pre + 'ln1')
            # k and v projections as one GEMM: attn.k.weight and attn.v.weight are adjacent in the flat parameter
            # buffer, so together they already are the stacked [2E, cout] matrix
            Wkv = self.buf('w.' + pre + 'kv', (2 * E, cout))
            WkvT = self.buf('wT.' + pre + 'kv', (cout, 2 * E)) if T else None
            self.prep.weight_prep(pk, 1, 2 * E, cout, 1, 1, dt, out=Wkv, ldo=cout, outT=WkvT, ldt=2 * E if T else 0,
                                  label='prep.' + pre + 'kv')
            h['Wkv'], h['WkvT'] = Wkv, WkvT
            h['kv'] = self.act(pre + 'kv', (B * (N + 1), 2 * E))
            F.gemm(h['un'], Wkv, h['kv'], B * (N + 1), 2 * E, cout, dt, label=pre + 'kv')
            Wq = self._w_plain(pre + 'attn.q.weight', E, cout, 1, 1)
            h['q'] = self.act(pre + 'q', (B, E))
            F.gemm(h['un'], Wq, h['q'], B, E, cout, dt, lda=(N + 1) * cout, label=pre + 'q')
            F.class_attn_fwd(h['q'], h['kv'], h['ao'], h['P'], B, N + 1, nh, hd, h['scale'], dt, label=pre + 'attn')
        Wpr = self._w_plain(pre + 'attn.proj.weight', cout, E, 1, 1, rs=P[pre + 'gamma_1'], src=self._pw(pre + 'attn.proj.weight'))
        bpr = self.buf('w.' + pre + 'bproj', (cout,), torch.float32)
        self.prep.bias_fold(None, P[pre + 'attn.proj.bias'], P[pre + 'gamma_1'], None, bpr, cout, E)
        dp = self.dp_scale.get(pre)
        h['cls1'] = self.buf(pre + 'cls1', (B, cout))
        F.gemm(h['ao'], Wpr, h['cls1'], B, cout, E, dt, bias=bpr, rowscale=dp, rows_per_scale=1, R=h['cls0'], ldr=cout,
               label=pre + 'proj')
        h['t'] = self.act(pre + 't', (B, cout))
        h['m2'] = self.act(pre + 'm2', (B,), torch.float32)
        h['r2'] = self.act(pre + 'r2', (B,), torch.float32)
        F.layernorm_fwd(h['cls1'], P[pre + 'norm2.weight'], P[pre + 'norm2.bias'], h['t'], h['m2'], h['r2'], B, cout, 1e-5,
                        dt, label=pre + 'ln2')
        h['cls2'] = self.fc_all['x'][k]                          # slice of [K][B][cout]: the five classifiers run as one GEMM
        h['mlp'] = self._gmlp_fwd(pre + 'mlp.', h['t'], B, cout, mg, h['cls2'], h['cls1'], dp, 1, gamma_name=pre + 'gamma_2')

    def _head_bwd(self, h, dlog, dx4, first):
        Bk, dt, B, P, W, cfg = self.bwd, self.dt, self.B, self.P, self.W, self.cfg
        k = h['k']
        g, E, nh, mg, NC = cfg['gram_dim'], cfg['dim_embed'], cfg['num_heads'], cfg['mlp_groups'], cfg['num_classes']
        groups = cfg['gram_groups']
        cout = self.cout
        M4 = dx4.shape[0]
        HW = M4 // B
        N = HW
        hd = E // nh
        if self.shared_tok:
            E, hd = self.Ea, self.hd_p
        pre = f'ga.{k}.'
        dp = self.dp_scale.get(pre)
        # classifier: done for all heads at once in _build_backward
        dcls2 = self.fc_all['dx'][k]
        dmz = dcls2
        if dp is not None:
            dmz = self.tmp('dmz', (B, cout))
            Bk.rowscale(dcls2, dp, dmz, B * cout, cout, dt)
        dtk = self.tmp('dt', (B, cout))
        self._gmlp_bwd(pre + 'mlp.', h['mlp'], dmz, h['t'], B, cout, mg, dtk, gamma_name=pre + 'gamma_2')
        dcls1 = self.tmp('dcls1', (B, cout))
        Bk.layernorm_bwd(dtk, h['cls1'], h['m2'], h['r2'], P[pre + 'norm2.weight'], dcls2, dcls1,
                         self.grad(pre + 'norm2.weight'), self.grad(pre + 'norm2.bias'), B, cout, False, dt, label=pre + 'ln2b')
        dpz = dcls1
        if dp is not None:
            dpz = self.tmp('dpz', (B, cout))
            Bk.rowscale(dcls1, dp, dpz, B * cout, cout, dt)
        # attention projection
        Gp, gbp = self.gbuf((cout, E)), self.gbuf((cout,))
        Bk.wgrad(dpz, h['ao'], Gp, B, cout, E, dt, dbias=gbp, label=pre + 'proj.wg')
        Bk.weight_unfold(Gp, E, cout, E, gb=gbp, W=self._pw(pre + 'attn.proj.weight'), b=P[pre + 'attn.proj.bias'],
                         rs=P[pre + 'gamma_1'], dW=self._pg(pre + 'attn.proj.weight'), db=self.grad(pre + 'attn.proj.bias'),
                         d_rs=self.grad(pre + 'gamma_1'), label=pre + 'proj.unf')
        dao = self.tmp('dao', (B, E))
        Bk.gemm(dpz, W[pre + 'attn.proj.weight.T'], dao, B, E, cout, dt, ldb=pad8(cout), label=pre + 'proj.dg')
        dq = self.tmp('dq', (B, E))
        # k, v are adjacent parameters -> their gradients form one [2E, cout] matrix in the flat gradient buffer
        gk, gv = self.grad(pre + 'attn.k.weight'), self.grad(pre + 'attn.v.weight')
        assert gv.data_ptr() == gk.data_ptr() + gk.numel() * 4, 'k/v gradients must be adjacent in the flat buffer'
        if self.shared_tok:
            g1, b1 = P[pre + 'norm1.weight'], P[pre + 'norm1.bias']
            dg1, db1 = self.grad(pre + 'norm1.weight'), self.grad(pre + 'norm1.bias')
            tk = self.tok
            E2 = tk['E2']
            dkvc = self.tmp('dkvc', (B, E2))
            dkvt = tk['dkv'][:, k * E2:]
            Bk.class_attn_bwd2(dao, h['q'], h['kvc'], h['kvt'], h['P'], dq, dkvc, dkvt, B, N + 1, nh, hd, h['scale'], dt,
                               tok_ld=tk['ld'], label=pre + 'attnb')
            # class-token row's share of the effective k|v weight gradient (the token rows' share: one wgrad after the loop)
            Bk.wgrad(dkvc, h['cn'], tk['G'][k * E2:], B, E2, cout, dt, dbias=tk['gb'][k * E2:], label=pre + 'kvc.wg')
            Gq, gbq = self.gbuf((E, cout)), self.gbuf((E,))
            Bk.wgrad(dq, h['cn'], Gq, B, E, cout, dt, dbias=gbq, label=pre + 'q.wg')
            Bk.weight_unfold(Gq, cout, E, cout, gb=gbq, W=self._pw(pre + 'attn.q.weight'), cs=g1, v=b1,
                             dW=self._pg(pre + 'attn.q.weight'), d_cs=dg1, d_v=db1, label=pre + 'q.unf')
            # gradient wrt the normalised class-token row (the image-token rows: one GEMM over all heads after the loop)
            dcn = self.tmp('dcn', (B, cout))
            Bk.gemm(dkvc, tk['WT'][:, k * E2:], dcn, B, cout, E2, dt, ldb=tk['ld'], label=pre + 'kvc.dg')
            Bk.gemm(dq, W[pre + 'attn.q.weight.T'], dcn, B, cout, E, dt, ldb=pad8(E), R=dcn, ldr=cout, label=pre + 'q.dg')
            # dcls0 = dcls1 + LN'(dcn)
            Bk.layernorm_bwd(dcn, h['cn'], None, h['rc'], None, dcls1, dcls1, None, None, B, cout, True, dt, label=pre + 'ln1cb')
        else:
            dkv = self.tmp('dkv', (B * (N + 1), 2 * E))
            Bk.class_attn_bwd(dao, h['q'], h['kv'], h['P'], dq, dkv, B, N + 1, nh, hd, h['scale'], dt, label=pre + 'attnb')
            Bk.wgrad(dkv, h['un'], gk, B * (N + 1), 2 * E, cout, dt, label=pre + 'kv.wg')
            dun = self.tmp('dun', (B * (N + 1), cout))
            Bk.gemm(dkv, h['WkvT'], dun, B * (N + 1), cout, 2 * E, dt, label=pre + 'kv.dg')
            Bk.wgrad(dq, h['un'], self.grad(pre + 'attn.q.weight'), B, E, cout, dt, ldx=(N + 1) * cout, label=pre + 'q.wg')
            Bk.gemm(dq, W[pre + 'attn.q.weight.T'], dun, B, cout, E, dt, ldb=pad8(E), ldc=(N + 1) * cout, R=dun,
                    ldr=(N + 1) * cout, label=pre + 'q.dg')
            du = self.tmp('du_tok', (B * (N + 1), cout))
            Bk.layernorm_bwd(dun, h['u'], h['m1'], h['r1'], P[pre + 'norm1.weight'], None, du, self.grad(pre + 'norm1.weight'),
                             self.grad(pre + 'norm1.bias'), B * (N + 1), cout, False, dt, label=pre + 'ln1b')
            # dcls0 = dcls1 + du[:,0];  dx4 (+)= du[:,1:]
            Bk.token_split(du, dcls1, dx4, B, N, cout, True, not first, dt, label=pre + 'split')
        # gram_embedding BN + grouped conv
        pre = f'gram_embedding.{k}.'
        cg = cout // groups
        Kg, Kp = h['Kg'], h['Kp']
        de = self.tmp('de', (B, cout))
        self._bn_bwd(pre + '1.', h['bn_e'], dcls1, None, h['e'], de, B, cout)
        gW = self.grad(pre + '0.weight')
        dvec = self.buf(f'gram.{k}.dvec', (B, groups * Kp), zero=True)   # pad columns stay zero
        if h['emb_small'] is not None and self.pad_gmlp:
            # odd group width (86 / 122 output channels per group): the gradient in a group-padded copy [B][groups][pad8(cg)] (pad
            # columns zero) -> both products on the MFMA kernels; the weight gradient lands in a padded scratch [groups][pad8(cg)][Kg]
            # whose real rows are added to the parameter's gradient
            cgp = pad8(cg)
            dep = self.buf(pre + 'de.pad', (B, groups * cgp), zero=True)
            Bk.pad_copy(de, dep, B * groups, cg, cg, cgp, dt, label=pre + 'de.pad')
            Gp, gbp = self.gbuf((groups * cgp, Kg)), self.gbuf((groups * cgp,))
            Bk.wgrad(dep, h['vec'], Gp, B, cgp, Kg, dt, ldy=groups * cgp, ldx=groups * Kp, ldw=Kg, batch=groups, strideY=cgp,
                     strideX=Kp, strideW=cgp * Kg, dbias=gbp, strideDbias=cgp, label=pre + 'wg')
            Bk.pad_copy_f32(Gp, gW, groups, cg * Kg, cgp * Kg, cg * Kg, accumulate=True, label=pre + 'wg.unpad')
            Bk.pad_copy_f32(gbp, self.grad(pre + '0.bias'), groups, cg, cgp, cg, accumulate=True, label=pre + 'db.unpad')
            Bk.gemm(dep, W[pre + '0.weight.T'], dvec, B, Kg, cgp, dt, lda=groups * cgp, batch=groups, strideA=cgp,
                    strideB=Kg * cgp, ldb=cgp, ldc=groups * Kp, strideC=Kp, label=pre + 'dg')
        elif h['emb_small'] is not None:
            Bk.small_linear_bwd(h['emb_small'], de, dA=dvec, dW=gW, dbias=self.grad(pre + '0.bias'), label=pre + 'bwd')
        else:
            Bk.wgrad(de, h['vec'], gW, B, cg, Kg, dt, ldy=cout, ldx=groups * Kp, ldw=Kg, batch=groups, strideY=cg, strideX=Kp,
                     strideW=cg * Kg, dbias=self.grad(pre + '0.bias'), strideDbias=cg, label=pre + 'wg')
            Bk.gemm(de, W[pre + '0.weight.T'], dvec, B, Kg, cg, dt, lda=cout, batch=groups, strideA=cg, strideB=Kg * pad8(cg),
                    ldb=pad8(cg), ldc=groups * Kp, strideC=Kp, label=pre + 'dg')
        S = self.tmp('gramS', (B, g, g))
        Bk.gram_pack_bwd(dvec, h['vec'], h['inv'], S, B, g, groups, Kp, dt, label=f'gram.{k}.packb')
        dg1 = self.tmp('dg1', (M4, g))
        Bk.gemm(h['g1'], S, dg1, HW, g, g, dt, batch=B, strideA=HW * g, strideB=g * g, strideC=HW * g, alpha=h['alpha'],
                label=f'gram.{k}.dx')
        dg0 = self.tmp('dg0', (M4, g))
        self._gram_layer_bwd(h, dg1, dg0)
        self._head_contract_bwd(h, k, dg0, M4)

    def _head_contract_bwd(self, h, k, dg0, M4):
        # gram_contraction BN backward into this head's column slice; the conv's wgrad / dgrad run once for all heads
        g = self.cfg['gram_dim']
        pre = f'gram_contraction.{k}.'
        gcn = self.gcon
        self._bn_bwd(pre + '1.', h['bn_gc'], dg0, None, h['gc'], gcn['dout'][:, k * g:], M4, g, ldx=gcn['ld'], lddx=gcn['ld'])

    # ------------------------------------------------------------------------------------------
    # whole-network backward plan
    # ------------------------------------------------------------------------------------------
    def _build_heads_backward(self, x4, M4):
        """backward of _build_heads: classifiers, the heads, the shared parts; returns dx4 [M4, cout]"""
        Bk, dt, B, P, W, cfg = self.bwd, self.dt, self.B, self.P, self.W, self.cfg
        cout = self.cout
        K, NC = cfg['branches'], cfg['num_classes']
        self.dlogits = self.buf('dlogits', (K, B, NC))
        Bk.zero(self.arena, label='zero.arena')
        dx4 = self.tmp('dx4', (M4, cout))
        # classifiers of the five heads: one batched wgrad and one batched dgrad
        fa = self.fc_all
        Gfc, gbfc = self.gbuf((K, NC, cout)), self.gbuf((K, NC))
        Bk.wgrad(self.dlogits, fa['x'], Gfc, B, NC, cout, dt, batch=K, strideY=B * NC, strideX=B * cout, strideW=NC * cout,
                 dbias=gbfc, strideDbias=NC, label='fc.all.wg')
        for k in range(K):
            Bk.axpy_f32(self.grad(f'fc.{k}.weight'), Gfc[k], 1.0, NC * cout)
            Bk.axpy_f32(self.grad(f'fc.{k}.bias'), gbfc[k], 1.0, NC)
        fa['dx'] = self.tmp('dcls2_all', (K, B, cout))
        Bk.gemm(self.dlogits, fa['WT'], fa['dx'], B, cout, NC, dt, batch=K, strideA=B * NC, strideB=cout * pad8(NC),
                ldb=pad8(NC), strideC=B * cout, label='fc.all.dg')
        if self.shared_tok:     # written slice-wise by the heads, consumed after the loop
            tk = self.tok
            tk['dkv'] = self.tmp('dkv_all', (M4, tk['ld']))
            tk['G'], tk['gb'] = self.gbuf((tk['ld'], cout)), self.gbuf((tk['ld'],))
        self.gcon['dout'] = self.tmp('dgc_all', (M4, self.gcon['ld']))
        for k in range(K):
            if self.head_lanes > 1:
                Bk.lane, self.tmp_prefix = 1 + k % self.head_lanes, f'h{k}.'
            self._head_bwd(self.heads[k], self.dlogits[k], dx4, first=(k == 0))
        Bk.lane, self.tmp_prefix = 0, ''
        self._contract_all_bwd(x4, dx4, M4)
        if self.shared_tok:
            tk = self.tok
            E2 = tk['E2']
            # token rows of all heads at once: effective k|v weight gradients, then each head's norm1 fold undone ...
            with self._wlane():
                Bk.wgrad(tk['dkv'], tk['xn'], tk['G'], M4, tk['ld'], cout, dt, dbias=tk['gb'], label='ga.kv_all.wg')
            for k in range(K):
                pre = f'ga.{k}.'
                Bk.weight_unfold(tk['G'][k * E2:], cout, E2, cout, gb=tk['gb'][k * E2:], W=self._pw(pre + 'attn.k.weight'),
                                 cs=P[pre + 'norm1.weight'], v=P[pre + 'norm1.bias'], dW=self._pg(pre + 'attn.k.weight'),
                                 d_cs=self.grad(pre + 'norm1.weight'), d_v=self.grad(pre + 'norm1.bias'), label=pre + 'kv.unf')

            # ... and dx4 += LayerNorm'(gradient wrt the shared normalised tokens, summed over the heads by the K = 5*2E GEMM)
            dxt = self.tmp('dxn_tok', (M4, cout))
            Bk.gemm(tk['dkv'], tk['WT'], dxt, M4, cout, tk['ld'], dt, label='ga.kv_all.dg')
            Bk.layernorm_bwd(dxt, tk['xn'], None, tk['rstd'], None, dx4, dx4, None, None, M4, cout, True, dt, label='ga.tok.lnb')
        return dx4

    def _contract_all_bwd(self, x4, dx4, M4):
        """weight / data gradients of the five gram_contraction convs; FIRST writer of dx4 on the shared-token path"""
        Bk, dt, cfg, cout = self.bwd, self.dt, self.cfg, self.cout
        K = cfg['branches']
        # gram_contraction convs of all heads: one wgrad (rows k*g.. -> head k's weight / bias gradient), one dgrad
        gcn, g_ = self.gcon, cfg['gram_dim']
        Gc, gbc = self.gbuf((gcn['ld'], cout)), self.gbuf((gcn['ld'],))
        with self._wlane():
            Bk.wgrad(gcn['dout'], x4, Gc, M4, gcn['ld'], cout, dt, dbias=gbc, label='gram_contraction.all.wg')
        for k in range(K):
            pre = f'gram_contraction.{k}.'
            Bk.axpy_f32(self.grad(pre + '0.weight'), Gc[k * g_:], 1.0, g_ * cout)
            Bk.axpy_f32(self.grad(pre + '0.bias'), gbc[k * g_:], 1.0, g_)
        # the first writer of dx4 on the shared-token path (the per-head token_split of the other path has written it)
        Bk.gemm(gcn['dout'], gcn['WT'], dx4, M4, cout, gcn['ld'], dt, R=None if self.shared_tok else dx4, ldr=cout,
                label='gram_contraction.all.dg')

    def _build_backward(self, feats, taps, stage_in, x4, M4, ctot):
        Bk, dt, B, P, W, cfg = self.bwd, self.dt, self.B, self.P, self.W, self.cfg
        d, dep = cfg['dims'], cfg['depths']
        dx4 = self._build_heads_backward(x4, M4)
        dcat = self.tmp('dcat', (M4, ctot))
        self._bottleneck_bwd(dx4, dcat)
        if self.async_wgrad:
            Bk.join_async()
        Bk.flush('heads.')
        self._unpad_all()
        Bk.mark('heads')      # every gradient of stages.4 / gram_* / ga / fc is final here
        # aggregate backward -> gradient seeds of the stage outputs / taps
        seeds = []
        for src, hw, c, mode, off in self.agg_segs:
            ds = self.buf(f'agg.d{off}', (B * hw * hw, c))
            Bk.pool_concat_bwd(dcat, None, ds, B, hw, hw, c, 14, 14, ctot, off, mode, dt, label=f'agg.b{off}')
            seeds.append(ds)
        ntap = len(taps)
        d_s0, d_s1 = seeds[0], seeds[1]
        d_taps = seeds[2:2 + ntap]
        d_s2, d_s3 = seeds[2 + ntap], seeds[3 + ntap]
        seed = {0: d_s0, 1: d_s1, 2: d_s2, 3: d_s3}
        tap_at = tap_indices(dep[2], cfg['naggre'])
        self._build_trunk_backward(seed, d_taps, tap_at, feats, stage_in)

    def _build_trunk_backward(self, seed, d_taps, tap_at, feats, stage_in, stem_seed=None):
        """backward of _build_trunk: seed[i] = gradient of stage i's output from the aggregation, d_taps those of the stage-2
        taps (after blocks tap_at), stem_seed that of the stem output (MAP: the stem output is a feature map itself)"""
        Bk, dt, B, P, W, cfg, nm = self.bwd, self.dt, self.B, self.P, self.W, self.cfg, self.NAMES
        d, dep = cfg['dims'], cfg['depths']
        dy = seed[3]
        for i in (3, 2, 1, 0):
            res = feats[i][1]
            Mi = B * res * res
            pp = [self.tmp(f'dxA{i}', (Mi, d[i])), self.tmp(f'dxB{i}', (Mi, d[i])), self.tmp(f'dxC{i}', (Mi, d[i]))]
            turn = 0
            for j in reversed(range(dep[i])):
                if i == 2 and j in tap_at:
                    dtap = d_taps[tap_at.index(j)]
                    Bk.affine_act(dy, None, None, dtap, dy, Mi, d[i], False, dt, label=f'tap.add.{j}')
                dx = pp[turn % 3]          # not this block's dy nor the previous block's (still read by its wgrad)
                turn += 1
                nxt = nm['block'].format(i=i, j=j - 1) if j > 0 and not (i == 2 and (j - 1) in tap_at) else None
                self._block_bwd(nm['block'].format(i=i, j=j), dy, dx, next_pre=nxt)
                dy = dx
            if i > 0:
                pre = f'stages.{i}.downsample.'          # buffer names only
                pln, pcv = nm['ds_ln'].format(i=i), nm['ds_conv'].format(i=i)
                x_prev, Hp = stage_in[i - 1]
                Mp = B * Hp * Hp
                G = self.gbuf((d[i], 4 * d[i - 1]))
                with self._wlane():
                    Bk.wgrad(dy, self.bufs[pre + 'ln'], G, Mi, d[i], 4 * d[i - 1], dt, x_kind=A_PATCH2,
                             x_dims=(Hp, Hp, d[i - 1]), dbias=self.grad(pcv + 'bias'), label=pre + 'wg')
                Bk.weight_unfold(G, 4 * d[i - 1], d[i], d[i - 1], 2, 2, dW=self.grad(pcv + 'weight'), label=pre + 'unf')
                dln = self.tmp('dln', (Mp, d[i - 1]))
                Bk.gemm(dy, W[pcv + 'weight.T'], dln, Mi, 4 * d[i - 1], d[i], dt, ldb=pad8(d[i]), c_kind=C_UNPATCH2,
                        c_dims=(Hp, Hp, d[i - 1]), label=pre + 'dg')
                dprev = self.tmp(f'dprev{i}', (Mp, d[i - 1]))
                Bk.layernorm_bwd(dln, x_prev, self.bufs[pre + 'mean'], self.bufs[pre + 'rstd'], P[pln + 'weight'],
                                 seed[i - 1], dprev, self.grad(pln + 'weight'), self.grad(pln + 'bias'), Mp, d[i - 1],
                                 False, dt, label=pre + 'lnb')
                dy = dprev
            if self.async_wgrad:
                Bk.join_async()
            Bk.flush(f'stage{i}.')
            Bk.mark(f'stage{i}')  # gradients of stages.i (incl. its downsample) are final
        # stem
        M0 = dy.shape[0]
        sc, sl = nm['stem_conv'], nm['stem_ln']
        if stem_seed is not None:
            Bk.affine_act(dy, None, None, stem_seed, dy, M0, d[0], False, dt, label='stem.seed')
        dpre = self.tmp('dstem', (M0, d[0]))
        Bk.layernorm_bwd(dy, self.bufs['stem.pre'], self.bufs['stem.mean'], self.bufs['stem.rstd'], P[sl + 'weight'], None,
                         dpre, self.grad(sl + 'weight'), self.grad(sl + 'bias'), M0, d[0], False, dt, label='stem.lnb')
        with self._wlane():
            Bk.wgrad(dpre, self.x_placeholder, self.grad(sc + 'weight'), M0, d[0], 48, dt, x_kind=A_STEM4_NCHW,
                     x_dims=(self.img, self.img, 3), dbias=self.grad(sc + 'bias'), label='stem.wg')
        self.input_bwd_desc = self._last_desc(Bk)

    # ------------------------------------------------------------------------------------------
    # run
    # ------------------------------------------------------------------------------------------
    # ImageNet statistics x 255, as timm's PrefetchLoader holds them for the uint8 batches of fast_collate (GA/train.py:567-595)
    U8_MEAN = (0.485 * 255, 0.456 * 255, 0.406 * 255)
    U8_STD = (0.229 * 255, 0.224 * 255, 0.225 * 255)

    def _normalize_u8(self, x):
        """a uint8 (B, 3, H, W) batch is normalised on the device into an engine-owned fp32 buffer (no host round trip)"""
        if x.dtype != torch.uint8:
            return x
        assert x.is_cuda and tuple(x.shape) == (self.B, 3, self.img, self.img), f'uint8 input of shape {tuple(x.shape)}'
        out = self.buf('x.u8norm', (self.B, 3, self.img, self.img), torch.float32)
        mean = getattr(self.m, 'input_mean', None) or self.U8_MEAN
        std = getattr(self.m, 'input_std', None) or self.U8_STD
        Plan(eager=True).u8_normalize(x.contiguous(), out, mean, std)
        return out

    def set_input(self, x):
        x = self._normalize_u8(x)
        assert x.is_cuda and x.dtype == torch.float32 and tuple(x.shape) == (self.B, 3, self.img, self.img), \
            f'input must be a float32 CUDA tensor of shape {(self.B, 3, self.img, self.img)}, got {tuple(x.shape)} {x.dtype}'
        if not x.is_contiguous():
            x = x.contiguous()   # channels_last callers (GA/train.py:729-730): the stem gather reads NCHW
        self.x_ref = x
        ptr = x.data_ptr()
        if getattr(self, 'stem_call', None) is not None:
            fn, args, label = self.fwd.calls[self.stem_call]
            self.fwd.calls[self.stem_call] = (fn, (ptr,) + tuple(args[1:]), label)
        else:
            self.input_descs[0].A = ptr
        if getattr(self, 'input_bwd_desc', None) is not None:
            self.input_bwd_desc.X = ptr

    def forward(self, x):
        self.set_input(x)
        if self.training or self.weights_dirty:
            self.prep.run()
            self.weights_dirty = False
            if self.training:
                for e in self.m._engines.values():
                    if e is not self:
                        e.weights_dirty = True
        if self.training and self.dp_scale and not getattr(self, 'fixed_masks', False):
            self.sample_drop_path()
        self.fwd.run()
        if self.training:
            self.m.count_training_forward()     # num_batches_tracked: host-side count, written on state_dict()
        return self.logits.view_as(self.logits)

    def _loss_operands(self):
        """(per-head logits, extra logits, d logits, d extra, heads) of the fused loss: the K GA heads here"""
        return self.logits, None, self.dlogits, None, self.logits.shape[0]

    def build_loss(self, lam, kind=0, smoothing=0.0, grad_scale=1.0, dense=False, bce_threshold=-1.0):
        """fused loss writing d(loss)/d(logits) * grad_scale straight into the backward plan's input buffer; dense: the target
        is a (B, NC) fp32 tensor (mixup / cutmix) instead of class indices"""
        org, avg, dorg, davg, K = self._loss_operands()
        _, B, NC = self.logits.shape
        self.loss_buf = self.buf('loss', (1,), torch.float32)
        self.target_buf = self.buf('target.dense', (B, NC), torch.float32) if dense else self.buf('target', (B,), torch.int64)
        lp = Plan(name='loss')
        lp.zero(self.loss_buf)
        lp.loss_dense_fwd_bwd(org, avg, None if dense else self.target_buf, self.target_buf if dense else None, self.loss_buf, dorg, davg,
                              K, B, NC, float(lam), int(kind), float(smoothing), float(bce_threshold), float(grad_scale), self.dt)
        self.loss_plan = lp
        self.loss_cfg = (lam, kind, smoothing, grad_scale, dense, bce_threshold)

    def forward_loss(self, x, target, lam, kind=0, smoothing=0.0, grad_scale=1.0, bce_threshold=-1.0):
        """forward + loss (+ dlogits) without autograd; follow with backward_range()/bwd.run(). Returns the loss buffer.
        target: class indices (B,) or a dense (B, NC) floating-point target"""
        dense = target.dim() == 2
        if getattr(self, 'loss_cfg', None) != (lam, kind, smoothing, grad_scale, dense, bce_threshold):
            self.build_loss(lam, kind, smoothing, grad_scale, dense, bce_threshold)
        self.forward(x)
        self.target_buf.copy_(target, non_blocking=True)
        self.loss_plan.run()
        return self.loss_buf

    def backward(self, dlogits):
        if dlogits.data_ptr() != self.dlogits.data_ptr():
            if dlogits.dtype == self.tdt:
                self.dlogits.copy_(dlogits)
            else:
                p = Plan(eager=True)
                p.cast_from_f32(dlogits.contiguous().float(), self.dlogits, self.dlogits.numel(), self.dt)
        self.bwd.run()
