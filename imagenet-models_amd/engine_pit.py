"""MAPPiTEngine: launch plans of the pooling transformer of /root/reference/MAP/models/map_pit.py (PoolingTransformer with
pool_type='map', :84-201) feeding the MAP head (engine_map.MAPEngine).

  conv_embedding (:71-81, 16 x 16 / stride 8, overlapping) ... ga_patchify_strided + ga_gemm (bias in the epilogue)
  + pos_embed (:190-191, an NCHW parameter) ..................... transposed once per step to [HW][C]; ga_pos_add_fwd / _bwd
  Transformer stages (:23-55; rearrange to tokens, timm Blocks) .. token rows ARE the NHWC map: no rearrange; the ViT block of
                                                                    engine_vit (LayerNorm folded into the next linear, ga_attn_*
                                                                    with head_dim 48 on the zero-padded 64-wide MFMA tiles)
  conv_head_pooling (:58-68) ...................................... ga_dwpool_fwd / _bwd_data / _bwd_weight
  forward_features' list (:185-201) -> MAPHead (:133-144) ......... MultiScale at the level-2 map's size: the two 27 x 27 maps
                                                                    through ga_resize_concat_* (general bilinear), the 14 x 14 one
                                                                    copied, the 7 x 7 one enlarged (ga_pool_concat_*)
"""
import torch

from . import ops  # noqa: F401
from .engine_vit import MAPViTEngine


class MAPPiTEngine(MAPViTEngine):
    def _drop_path_rates(self):
        """map_pit.py:116-118: drop_path_rate * i / total_block; timm's Block applies it to both residual branches"""
        cfg = self.cfg
        tot, i, out = sum(cfg['depth']), 0, {}
        for s, d in enumerate(cfg['depth']):
            for j in range(d):
                out[f'transformers.{s}.blocks.{j}.#1'] = out[f'transformers.{s}.blocks.{j}.#2'] = cfg['drop_path_rate'] * i / tot
                i += 1
        return out

    def _build(self):
        cfg = self.cfg
        B, T, F, dt, P = self.B, self.training, self.fwd, self.dt, self.P
        self.img = img = self._img
        dims, depth, heads, ps, stride, w0 = cfg['dims'], cfg['depth'], cfg['heads'], cfg['patch_size'], cfg['stride'], cfg['width']
        K0 = 3 * ps * ps
        C0, Mp0 = dims[0], B * w0 * w0
        if T:
            F.zero(self.bn_pool, label='zero.bn_sums')
        # ---------------- conv_embedding + pos_embed ----------------
        self.x_placeholder = torch.zeros(B, 3, img, img, device=self.dev)
        patches = self.patches = self.act('patch.cols', (Mp0, K0))
        F.patchify_strided(self.x_placeholder, patches, ps, stride, dt, label='patch.pack')
        self.pack_call = len(F.calls) - 1
        Wpe = self._w_plain('patch_embed.conv.weight', C0, K0, 1, 1, need_T=False)
        tok = self.tmp('patch.tok', (Mp0, C0))
        posT = self.buf('w.posT', (w0 * w0, C0), torch.float32)
        self.prep.transpose_f32(P['pos_embed'], posT, C0, w0 * w0)
        x0 = self.buf('embed.x0', (Mp0, C0))
        # ---------------- embedding + stages: one pass per forward chain (batch part on its own lane, GAEngine._chains) ----------------
        chains = self._chains()
        for chain in chains:
            self._chain = chain if len(chains) > 1 else None
            (lane, r0, r1, b0, b1), = self._fsplits(w0 * w0)
            F.lane = lane
            nb = b1 - b0
            F.gemm(patches[r0:r1], Wpe, tok[r0:r1], r1 - r0, C0, K0, dt, bias=P['patch_embed.conv.bias'], label='patch.proj')
            F.pos_add_fwd(tok[r0:r1], posT, x0[r0:r1], nb, w0 * w0, C0, dt, label='embed')
            x = x0
            feats = [(x, w0, C0)]
            self.stage_io = []                                # (input map, output map, hw, C) per stage
            hw = w0
            for s_ in range(3):
                C, Ntok = dims[s_], hw * hw
                xin = x
                for j in range(depth[s_]):
                    x = self._vit_block_fwd(f'transformers.{s_}.blocks.{j}.', x, B * Ntok, C, heads[s_], Ntok)
                feats.append((x, hw, C))
                self.stage_io.append((xin, x, hw, C))
                if s_ < 2:
                    ho = (hw - 1) // 2 + 1
                    y = self.buf(f'pool.{s_}.y', (B * ho * ho, dims[s_ + 1]))
                    F.dwpool_fwd(x[b0 * hw * hw:b1 * hw * hw], P[f'pools.{s_}.conv.weight'], P[f'pools.{s_}.conv.bias'],
                                 y[b0 * ho * ho:b1 * ho * ho], nb, hw, hw, C, dims[s_ + 1] // C, dt, label=f'pools.{s_}')
                    x, hw = y, ho
        self._chain = None
        F.lane = 0
        # ---------------- MultiScale at the size of feature `multi_scale_level` (map.py:322-333) ----------------
        Hc = self.Hc = feats[cfg['multi_scale_level']][1]
        M4 = B * Hc * Hc
        ctot = sum(c for _, _, c in feats)
        cat = self.act('ms.cat', (M4, ctot))
        self.agg_segs, off = [], 0
        for fm, fhw, c in feats:
            if fhw == Hc:
                mode = 0
            elif fhw < Hc:
                assert Hc % fhw == 0, 'adaptive_avg_pool2d enlargement by a non-integer factor is not on the registered path'
                mode = 3
            else:
                mode = 2 if fhw % Hc == 0 else 'resize'
            if mode == 'resize':
                F.resize_concat_fwd(fm, cat, B, fhw, fhw, c, Hc, Hc, ctot, off, dt, label=f'agg.{off}')
            else:
                F.pool_concat_fwd(fm, cat, B, fhw, fhw, c, Hc, Hc, ctot, off, mode, dt, label=f'agg.{off}')
            self.agg_segs.append((fm, fhw, c, mode, off))
            off += c
        xh = self._multi_scale_conv_fwd(cat, M4, ctot)
        self._build_map_head(xh, M4, Hc)
        if T:
            self._build_pit_backward(xh, M4, K0)
            if self.async_wgrad:
                self.bwd.join_async()
            self.bwd.flush('end.')
        self.prep.flush('prep.')

    def _build_pit_backward(self, xh, M4, K0):
        Bk, dt, P, cfg, B = self.bwd, self.dt, self.P, self.cfg, self.B
        dims, depth, w0 = cfg['dims'], cfg['depth'], cfg['width']
        dcat = self._build_head_backward(xh, M4)
        ctot = self.ms['ctot']
        seeds = []
        for fm, fhw, c, mode, off in self.agg_segs:
            ds = self.buf(f'agg.d{off}', (B * fhw * fhw, c))
            if mode == 'resize':
                Bk.resize_concat_bwd(dcat, ds, B, fhw, fhw, c, self.Hc, self.Hc, ctot, off, dt, label=f'agg.b{off}')
            else:
                Bk.pool_concat_bwd(dcat, None, ds, B, fhw, fhw, c, self.Hc, self.Hc, ctot, off, mode, dt, label=f'agg.b{off}')
            seeds.append(ds)
        # sequence gradients rotate through three buffers (the asynchronous weight gradients still read the one before); they are
        # sized for the largest stage and viewed per stage
        nmax = max(B * hw * hw * C for _, _, hw, C in self.stage_io)
        rot = [self.buf(f'pit.dx{j}', (nmax,)) for j in range(3)]
        cur = 0
        dy = seeds[3]                                         # gradient wrt the last stage's output
        for s in range(2, -1, -1):
            xin, xout, hw, C = self.stage_io[s]
            Ntok = hw * hw
            M = B * Ntok
            for j in range(depth[s] - 1, -1, -1):
                cur = (cur + 1) % 3
                dx = rot[cur][:M * C].view(M, C)
                self._vit_block_bwd(f'transformers.{s}.blocks.{j}.', dy, dx, M, C, Ntok,
                                    next_pre=f'transformers.{s}.blocks.{j - 1}.' if j > 0 else None)      # (stage seams add a feature seed)
                dy = dx
            if s > 0:
                # conv_head_pooling backward; the stage below's output also fed MultiScale: its seed is added
                _, xprev, hwp, Cp = self.stage_io[s - 1]
                mult = C // Cp
                with self._wlane():
                    Bk.dwpool_bwd_weight(dy, xprev, self.grad(f'pools.{s - 1}.conv.weight'), self.grad(f'pools.{s - 1}.conv.bias'), B, hwp,
                                         hwp, Cp, mult, dt, label=f'pools.{s - 1}.wg')
                dprev = self.buf(f'pool.{s - 1}.dx', (B * hwp * hwp, Cp))
                Bk.dwpool_bwd_data(dy, P[f'pools.{s - 1}.conv.weight'], dprev, B, hwp, hwp, Cp, mult, dt, label=f'pools.{s - 1}.dg')
                Bk.copy2d(seeds[s], hwp * hwp * Cp, dprev, hwp * hwp * Cp, B, hwp * hwp * Cp, dt, accumulate=True, label=f'feat.{s}.b')
                dy = dprev
                # transformers.s and pools.(s-1) are final only after the pooling conv's weight gradient (asynchronous lane) and the
                # deferred unfold jobs of the stage's blocks: join + flush BEFORE the mark the gradient buckets key on
                if self.async_wgrad:
                    Bk.join_async()
                Bk.flush(f'stage{s + 1}.')
                Bk.mark(f'stage{s + 1}')
        # dy: gradient wrt x0 from stage 0; x0 is also feature 0
        Bk.copy2d(seeds[0], w0 * w0 * dims[0], dy, w0 * w0 * dims[0], B, w0 * w0 * dims[0], dt, accumulate=True, label='feat.0.b')
        dposT = self.tmp('dposT', (w0 * w0, dims[0]), torch.float32)
        Bk.pos_add_bwd(dy, dposT, B, w0 * w0, dims[0], dt, label='embedb')
        Bk.transpose_f32(dposT, self.grad('pos_embed'), w0 * w0, dims[0], accumulate=True)
        with self._wlane():
            Bk.wgrad(dy, self.patches, self.grad('patch_embed.conv.weight'), B * w0 * w0, dims[0], K0, dt,
                     dbias=self.grad('patch_embed.conv.bias'), label='patch.wg')
