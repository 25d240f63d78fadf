"""MAPViTEngine: launch plans of a ViT trunk (timm VisionTransformer: PatchEmbed, cls_token + pos_embed, `Block`s -- the block the
reference uses through /root/reference/MAP/models/map_pit.py:14,35-44) feeding the MAP head (engine_map.MAPEngine, map.py).

The composition (BASELINE configs[4] "MAP-ViT-B/16 @ 384") is builder-defined, modelled on how map_pit.py attaches the head
(PoolingTransformer.forward_features :185-201: the position-embedded patch tokens and the output of every stage are the feature
maps handed to MAPHead): here the 12 blocks are cut into three equal "stages"; features = [tokens after pos_embed, after block
d/3, after block 2d/3, final norm of the last block], class token dropped, each a (B, C, H/16, W/16) map.  MultiScale resizes
them to half that resolution (bilinear reduction, as map.py:322-333 does for maps above its level's size).

Trunk kernels: ga_patchify + ga_gemm (patch embedding), ga_vit_embed, per block LayerNorm(1e-6, affine folded into the next
linear) -> qkv ga_gemm -> ga_attn (global attention, flash-style MFMA in bf16) -> proj ga_gemm (+ DropPath + residual) ->
LayerNorm -> fc1 (+GELU) -> fc2 (+ DropPath + residual); weight gradients on the asynchronous lane as in the other engines.
"""
import torch

from . import ops  # noqa: F401
from .engine import pad8
from .engine_map import MAPEngine
from .ops import ACT_GELU


class MAPViTEngine(MAPEngine):
    def __init__(self, model, batch, training, mode):
        self._img = model.cfg['img_size']
        super().__init__(model, batch, training, mode)

    def _drop_path_rates(self):
        """timm VisionTransformer: linspace(0, drop_path_rate, depth); a Block applies it to both branches"""
        cfg = self.cfg
        dpr = torch.linspace(0, cfg['drop_path_rate'], cfg['depth']).tolist()
        out = {}
        for i in range(cfg['depth']):
            out[f'blocks.{i}.#1'] = out[f'blocks.{i}.#2'] = dpr[i]
        return out

    # ------------------------------------------------------------------------------------------
    def _vit_block_fwd(self, pre, x, M, C, heads, Ntok):
        """inside a forward chain (GAEngine._chains: the trunk recorded once per batch part, each part on its own lane) the launches
        cover the chain's rows of the same full-batch buffers; weight preparation and the backward's attention descriptor are recorded
        by the first pass only"""
        F, dt, B, P, T = self.fwd, self.dt, self.B, self.P, self.training
        (lane, r0, r1, b0, b1), = self._fsplits(Ntok)
        first = pre not in self.blocks
        if getattr(self, '_chain', None) is not None:
            F.lane = lane
        n = r1 - r0
        dp1, dp2 = self.dp_scale.get(pre + '#1'), self.dp_scale.get(pre + '#2')
        dp1c = dp1[b0:b1] if dp1 is not None else None
        dp2c = dp2[b0:b1] if dp2 is not None else None
        st = self.blocks.setdefault(pre, dict(x=x))
        st['xn1'] = self.blk_act(pre + 'xn1', (M, C))
        st['r1'] = self.blk_act(pre + 'r1', (M,), torch.float32)
        F.layernorm_fwd(x[r0:r1], None, None, st['xn1'][r0:r1], None, st['r1'][r0:r1], n, C, 1e-6, dt, label=pre + 'ln1')
        Wqkv = self._w_plain(pre + 'attn.qkv.weight', 3 * C, C, 1, 1, cs=P[pre + 'norm1.weight'])
        bq = self.buf('w.' + pre + 'bqkv', (3 * C,), torch.float32)
        if first:
            self.prep.bias_fold(P[pre + 'attn.qkv.weight'], P[pre + 'attn.qkv.bias'], None, P[pre + 'norm1.bias'], bq, 3 * C, C)
        st['qkv'] = self.blk_act(pre + 'qkv', (M, 3 * C))
        F.gemm(st['xn1'][r0:r1], Wqkv, st['qkv'][r0:r1], n, 3 * C, C, dt, bias=bq, label=pre + 'qkv')
        st['att'] = self.blk_act(pre + 'att', (M, C))
        st['lse'] = self.blk_act(pre + 'lse', (B, heads, Ntok), torch.float32)
        if first:
            st['desc'] = F.attn_desc(st['qkv'], st['att'], st['lse'], B, Ntok, heads, C // heads, (C // heads) ** -0.5, dt)
        dfw = st['desc'] if n == M else F.attn_desc(st['qkv'][r0:r1], st['att'][r0:r1], st['lse'][b0:b1], b1 - b0, Ntok, heads, C // heads,
                                                      (C // heads) ** -0.5, dt)
        F.attn_fwd(dfw, label=pre + 'attn')
        Wp = self._w_plain(pre + 'attn.proj.weight', C, C, 1, 1)
        x1 = self.tmp('x1', (M, C))
        F.gemm(st['att'][r0:r1], Wp, x1[r0:r1], n, C, C, dt, bias=P[pre + 'attn.proj.bias'], rowscale=dp1c, rows_per_scale=Ntok,
               R=x[r0:r1], ldr=C, label=pre + 'proj')
        st['xn2'] = self.blk_act(pre + 'xn2', (M, C))
        st['r2'] = self.blk_act(pre + 'r2', (M,), torch.float32)
        F.layernorm_fwd(x1[r0:r1], None, None, st['xn2'][r0:r1], None, st['r2'][r0:r1], n, C, 1e-6, dt, label=pre + 'ln2')
        W1 = self._w_plain(pre + 'mlp.fc1.weight', 4 * C, C, 1, 1, cs=P[pre + 'norm2.weight'])
        b1e = self.buf('w.' + pre + 'b1e', (4 * C,), torch.float32)
        if first:
            self.prep.bias_fold(P[pre + 'mlp.fc1.weight'], P[pre + 'mlp.fc1.bias'], None, P[pre + 'norm2.bias'], b1e, 4 * C, C)
        st['a'] = self.blk_act(pre + 'a', (M, 4 * C))
        st['g'] = self.buf(pre + 'g', (M, 4 * C)) if T else None
        F.gemm(st['xn2'][r0:r1], W1, st['a'][r0:r1], n, 4 * C, C, dt, bias=b1e, act=ACT_GELU, C2=st['g'][r0:r1] if T else None,
               c2_mode=2 if T else 0, label=pre + 'fc1')
        W2 = self._w_plain(pre + 'mlp.fc2.weight', C, 4 * C, 1, 1)
        y = self.buf(pre + 'y', (M, C))
        F.gemm(st['a'][r0:r1], W2, y[r0:r1], n, C, 4 * C, dt, bias=P[pre + 'mlp.fc2.bias'], rowscale=dp2c, rows_per_scale=Ntok,
               R=x1[r0:r1], ldr=C, label=pre + 'fc2')
        return y

    def _vit_block_bwd(self, pre, dy, dx, M, C, Ntok, next_pre=None):
        """dy: gradient wrt the block output; writes dx (a different buffer) = gradient wrt the block input.  next_pre: the block
        that consumes dx unchanged as ITS dy (the next one of the backward chain, when no feature seed is added in between): this
        block's last LayerNorm backward then also writes that block's DropPath-scaled copy (self._pre_dyz), saving it a pass"""
        Bk, dt, P, W = self.bwd, self.dt, self.P, self.W
        st = self.blocks[pre]
        dp1, dp2 = self.dp_scale.get(pre + '#1'), self.dp_scale.get(pre + '#2')
        side = self.async_wgrad and Bk.lane == 0
        par = ''
        if side:       # two sets of the transients the asynchronous weight-gradient launches read; this block waits for block t-2's
            self._bwd_seq += 1
            par = str(self._bwd_seq & 1)
            Bk.join_async(f'blk{self._bwd_seq - 2}')
        dyz = dy
        if dp2 is not None:
            dyz = self._pre_dyz.pop(pre, None)
            if dyz is None:
                dyz = self.tmp('dyz' + par, (M, C))
                Bk.rowscale(dy, dp2, dyz, M * C, Ntok * C, dt, label=pre + 'dp2')
        with self._wlane():
            Bk.wgrad(dyz, st['a'], self.grad(pre + 'mlp.fc2.weight'), M, C, 4 * C, dt, dbias=self.grad(pre + 'mlp.fc2.bias'),
                     label=pre + 'wg2')
        dh = self.tmp('dh' + par, (M, 4 * C))
        gb1 = self.gbuf((4 * C,))
        Bk.gemm(dyz, W[pre + 'mlp.fc2.weight.T'], dh, M, 4 * C, C, dt, ldb=pad8(C), H=st['g'], ldh=4 * C, h_is_deriv=True, colsum=gb1,
                label=pre + 'dg2')
        G1 = self.gbuf((4 * C, C))
        with self._wlane():
            Bk.wgrad(dh, st['xn2'], G1, M, 4 * C, C, dt, label=pre + 'wg1')
        gx = self.tmp('g', (M, C))
        Bk.gemm(dh, W[pre + 'mlp.fc1.weight.T'], gx, M, C, 4 * C, dt, ldb=pad8(4 * C), label=pre + 'dg1')
        dx1 = self.tmp('dx1' + par, (M, C))
        dx1z = dx1
        if dp1 is not None and self.fuse_dp:       # the DropPath-scaled copy of dx1 rides on the LayerNorm backward that writes dx1
            dx1z = self.tmp('dx1z' + par, (M, C))
            Bk.layernorm_bwd(gx, st['xn2'], None, st['r2'], None, dy, dx1, None, None, M, C, True, dt, label=pre + 'ln2b', dx2=dx1z,
                             scale2=dp1, rows_per_scale=Ntok)
        else:
            Bk.layernorm_bwd(gx, st['xn2'], None, st['r2'], None, dy, dx1, None, None, M, C, True, dt, label=pre + 'ln2b')
        Bk.weight_unfold(G1, C, 4 * C, C, gb=gb1, W=P[pre + 'mlp.fc1.weight'], b=P[pre + 'mlp.fc1.bias'], cs=P[pre + 'norm2.weight'],
                         v=P[pre + 'norm2.bias'], dW=self.grad(pre + 'mlp.fc1.weight'), db=self.grad(pre + 'mlp.fc1.bias'),
                         d_cs=self.grad(pre + 'norm2.weight'), d_v=self.grad(pre + 'norm2.bias'), label=pre + 'unf1')
        if dp1 is not None and not self.fuse_dp:
            dx1z = self.tmp('dx1z' + par, (M, C))
            Bk.rowscale(dx1, dp1, dx1z, M * C, Ntok * C, dt, label=pre + 'dp1')
        with self._wlane():
            Bk.wgrad(dx1z, st['att'], self.grad(pre + 'attn.proj.weight'), M, C, C, dt, dbias=self.grad(pre + 'attn.proj.bias'),
                     label=pre + 'proj.wg')
        datt = self.tmp('datt' + par, (M, C))
        Bk.gemm(dx1z, W[pre + 'attn.proj.weight.T'], datt, M, C, C, dt, ldb=pad8(C), label=pre + 'proj.dg')
        dqkv = self.tmp('dqkv' + par, (M, 3 * C))
        ws = self.tmp('attn_delta', (st['lse'].numel(),), torch.float32)
        Bk.attn_bwd(st['desc'], datt, dqkv, ws, label=pre + 'attnb')
        Gq, gbq = self.gbuf((3 * C, C)), self.gbuf((3 * C,))
        with self._wlane():
            Bk.wgrad(dqkv, st['xn1'], Gq, M, 3 * C, C, dt, dbias=gbq, label=pre + 'qkv.wg')
        Bk.weight_unfold(Gq, C, 3 * C, C, gb=gbq, W=P[pre + 'attn.qkv.weight'], b=P[pre + 'attn.qkv.bias'], cs=P[pre + 'norm1.weight'],
                         v=P[pre + 'norm1.bias'], dW=self.grad(pre + 'attn.qkv.weight'), db=self.grad(pre + 'attn.qkv.bias'),
                         d_cs=self.grad(pre + 'norm1.weight'), d_v=self.grad(pre + 'norm1.bias'), label=pre + 'qkv.unf')
        gq = self.tmp('g', (M, C))
        Bk.gemm(dqkv, W[pre + 'attn.qkv.weight.T'], gq, M, C, 3 * C, dt, ldb=pad8(3 * C), label=pre + 'qkv.dg')
        ndp = self.dp_scale.get(next_pre + '#2') if next_pre is not None else None
        if ndp is not None and self.fuse_dp:
            # the consumer's transients alternate with the block parity exactly like this block's (par of the next block = 1 - par)
            npar = str((self._bwd_seq + 1) & 1) if side else ''
            ndyz = self.tmp('dyz' + npar, (M, C))
            if side:           # that buffer was block t-1's: its asynchronous weight gradients must be done before it is rewritten
                Bk.join_async(f'blk{self._bwd_seq - 1}')
            Bk.layernorm_bwd(gq, st['xn1'], None, st['r1'], None, dx1, dx, None, None, M, C, True, dt, label=pre + 'ln1b', dx2=ndyz,
                             scale2=ndp, rows_per_scale=Ntok)
            self._pre_dyz[next_pre] = ndyz
        else:
            Bk.layernorm_bwd(gq, st['xn1'], None, st['r1'], None, dx1, dx, None, None, M, C, True, dt, label=pre + 'ln1b')
        if side:
            Bk.async_mark(f'blk{self._bwd_seq}')

    # ------------------------------------------------------------------------------------------
    def _build(self):
        cfg = self.cfg
        B, T, F, dt, P = self.B, self.training, self.fwd, self.dt, self.P
        self.img = img = self._img
        C, depth, heads, ps = cfg['embed_dim'], cfg['depth'], cfg['vit_heads'], cfg['patch_size']
        gw = img // ps
        Np, Ntok = gw * gw, gw * gw + 1
        M, Mp = B * Ntok, B * Np
        K0 = 3 * ps * ps
        if T:
            F.zero(self.bn_pool, label='zero.bn_sums')
        # ---------------- patch embedding + class token + position embedding ----------------
        self.x_placeholder = torch.zeros(B, 3, img, img, device=self.dev)
        patches = self.patches = self.act('patch.cols', (Mp, K0))
        F.patchify(self.x_placeholder, patches, ps, dt, label='patch.pack')
        self.pack_call = len(F.calls) - 1
        Wpe = self._w_plain('patch_embed.proj.weight', C, K0, 1, 1, need_T=False)
        tok = self.tmp('patch.tok', (Mp, C))
        x0 = self.buf('embed.x0', (M, C))
        taps = cfg['taps']                                   # block counts after which a feature map is taken (the last = depth)
        # ---------------- embedding, blocks, feature taps: one pass per forward chain (batch part on its own lane) ----------------
        chains = self._chains()
        for chain in chains:
            self._chain = chain if len(chains) > 1 else None
            (lane, r0, r1, b0, b1), = self._fsplits(Ntok)
            F.lane = lane
            p0, p1, nb = b0 * Np, b1 * Np, b1 - b0
            F.gemm(patches[p0:p1], Wpe, tok[p0:p1], p1 - p0, C, K0, dt, bias=P['patch_embed.proj.bias'], label='patch.proj')
            F.vit_embed_fwd(tok[p0:p1], P['cls_token'], P['pos_embed'], x0[r0:r1], nb, Np, C, dt, label='embed')
            x = x0
            feats = [self._tokens_to_map(x, 'f0', B, Np, Ntok, C)]
            self.tap_at = {}
            for i in range(depth):
                x = self._vit_block_fwd(f'blocks.{i}.', x, M, C, heads, Ntok)
                if i + 1 in taps:
                    self.tap_at[i + 1] = len(feats)
                    fm = self._tokens_to_map(x, f'f{len(feats)}', B, Np, Ntok, C)
                    if i + 1 == depth:       # final norm (timm forward_features) on the map rows: LayerNorm is per token
                        self.fn = dict(x=fm, y=self.act('norm.out', (Mp, C)), m=self.act('norm.m', (Mp,), torch.float32),
                                       r=self.act('norm.r', (Mp,), torch.float32))
                        F.layernorm_fwd(fm[p0:p1], P['norm.weight'], P['norm.bias'], self.fn['y'][p0:p1], self.fn['m'][p0:p1],
                                        self.fn['r'][p0:p1], p1 - p0, C, 1e-6, dt, label='norm')
                        fm = self.fn['y']
                    feats.append(fm)
        self._chain = None
        F.lane = 0
        self.x_last = x
        # ---------------- MultiScale: every map (gw x gw) reduced to gw/2 x gw/2, concat, conv1x1 + BN + GELU ----------------
        Hc = self.Hc = gw // 2
        M4 = B * Hc * Hc
        ctot = C * len(feats)
        cat = self.act('ms.cat', (M4, ctot))
        self.agg_segs = []
        for j, fm in enumerate(feats):
            F.pool_concat_fwd(fm, cat, B, gw, gw, C, Hc, Hc, ctot, j * C, 2, dt, label=f'agg.{j}')
            self.agg_segs.append((fm, gw, C, 2, j * C))
        xh = self._multi_scale_conv_fwd(cat, M4, ctot)
        self._build_map_head(xh, M4, Hc)
        if T:
            self._build_vit_backward(xh, M4, feats, B, Np, Ntok, C, M, Mp, K0)
            if self.async_wgrad:
                self.bwd.join_async()
            self.bwd.flush('end.')
        self.prep.flush('prep.')

    def _tokens_to_map(self, x, name, B, Np, Ntok, C):
        """drop the class token: rows 1.. of every image -> [B*Np, C] (the images of the current forward chain)"""
        fm = self.act('feat.' + name, (B * Np, C))
        (lane, r0, r1, b0, b1), = self._fsplits(Ntok)
        self.fwd.copy2d(x[r0 + 1:], Ntok * C, fm[b0 * Np:], Np * C, b1 - b0, Np * C, self.dt, label='feat.' + name)
        return fm

    def _build_vit_backward(self, xh, M4, feats, B, Np, Ntok, C, M, Mp, K0):
        Bk, dt, P, cfg = self.bwd, self.dt, self.P, self.cfg
        depth, gw = cfg['depth'], self.img // cfg['patch_size']
        dcat = self._build_head_backward(xh, M4)
        ctot = self.ms['ctot']
        seeds = []
        for fm, hw, c, mode, off in self.agg_segs:
            ds = self.buf(f'agg.d{off}', (B * hw * hw, c))
            Bk.pool_concat_bwd(dcat, None, ds, B, hw, hw, c, self.Hc, self.Hc, ctot, off, mode, dt, label=f'agg.b{off}')
            seeds.append(ds)
        # gradient of the token sequence, three rotating buffers (the asynchronous weight gradients read dx of the block before)
        dxs = [self.buf(f'vit.dx{j}', (M, C)) for j in range(3)]
        cur = 0
        # last feature: through the final norm, into rows 1.. of a zeroed sequence gradient
        dfm = self.tmp('dnorm', (Mp, C))
        Bk.layernorm_bwd(seeds[-1], self.fn['x'], self.fn['m'], self.fn['r'], P['norm.weight'], None, dfm, self.grad('norm.weight'),
                         self.grad('norm.bias'), Mp, C, False, dt, label='normb')
        Bk.zero(dxs[cur], label='zero.dx')
        Bk.copy2d(dfm, Np * C, dxs[cur][1:], Ntok * C, B, Np * C, dt, label='feat.last.b')
        for i in range(depth - 1, -1, -1):
            nxt = (cur + 1) % 3
            j = self.tap_at.get(i) if i > 0 else 0           # the feature taken at this block's INPUT (after block i; 0 = the embedding)
            self._vit_block_bwd(f'blocks.{i}.', dxs[cur], dxs[nxt], M, C, Ntok, next_pre=f'blocks.{i - 1}.' if i > 0 and j is None else None)
            cur = nxt
            if j is not None:
                Bk.copy2d(seeds[j], Np * C, dxs[cur][1:], Ntok * C, B, Np * C, dt, accumulate=True, label=f'feat.{j}.b')
            if i == depth // 2:
                # every gradient of blocks[depth // 2 :] and of the final norm must be FINAL at the mark (TrainStep all-reduces
                # the group's slice from here on): wait for the weight-gradient lane, emit the deferred unfold jobs
                if self.async_wgrad:
                    Bk.join_async()
                Bk.flush('stage2.')
                Bk.mark('stage2')
        # embedding: dtok, d(cls_token), d(pos_embed); patch projection weight gradient
        dtok = self.tmp('dtok', (Mp, C))
        Bk.vit_embed_bwd(dxs[cur], dtok, self.grad('cls_token'), self.grad('pos_embed'), B, Np, C, dt, label='embedb')
        with self._wlane():
            Bk.wgrad(dtok, self.patches, self.grad('patch_embed.proj.weight'), Mp, C, K0, dt, dbias=self.grad('patch_embed.proj.bias'),
                     label='patch.wg')

    def set_input(self, x):
        x = self._normalize_u8(x)
        assert x.is_cuda and x.dtype == torch.float32 and tuple(x.shape) == (self.B, 3, self.img, self.img), \
            f'input must be a float32 CUDA tensor of shape {(self.B, 3, self.img, self.img)}, got {tuple(x.shape)} {x.dtype}'
        if not x.is_contiguous():
            x = x.contiguous()
        self.x_ref = x
        fn, args, label = self.fwd.calls[self.pack_call]
        self.fwd.calls[self.pack_call] = (fn, (x.data_ptr(),) + tuple(args[1:]), label)
