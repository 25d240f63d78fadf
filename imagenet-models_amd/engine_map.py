"""MAPEngine: the static launch plans (weight prep / forward / backward) of one MAP_ConvNeXt for a fixed
(batch, train|eval, math mode).  The ConvNeXt trunk is engine.GAEngine's (same kernels, the FB parameter names of
/root/reference/MAP/models/map_convnext.py:16-83); the head is restated here from /root/reference/MAP/models/map.py:

  MultiScale :322-333 ....... five maps resized to 14 x 14 (bilinear REDUCTION of the larger ones, adaptive-avg-pool
                              ENLARGEMENT of the 7 x 7 one) into one concat buffer -> conv1x1 -> BN -> GELU
  GramToken :210-234 ........ the n_groups ch_reduction convs as ONE stacked GEMM + BN -> Gram (X.X^T) -> triu / L2-normalise
                              / token interleave (ga_gram_pack_fwd2) -> grouped bp_reduction GEMM + BN -> tokens (+ mean token)
  CABlock :171-184 .......... norm1 over cat(class rows, image rows): the normalised image rows are shared by every group (affine
                              folded into each group's k | v weights, ONE stacked k|v GEMM); multi-token class attention
                              (ga_class_attn_mt); proj; norm2; GroupConvMlp (g = 2, ReLU)
  NormHead :402-412 ......... LayerNorm + Linear; heads on the n_tokens gram tokens, self_dt_heads on the mean token (train)

Train-mode logits buffer: [2 * n_groups][B][NC] = the org heads, then the avg (self-distillation) heads.
nn.Dropout (p = 0.05 on the attention probabilities, the projection output and the MLP hidden layer, map.py:149,464) is
applied through fp32 mask buffers that ga_dropout_mask_sample refills every training step.
"""
import os

import torch

from . import ops  # noqa: F401
from .engine import GAEngine, pad8
from .ops import Plan


class MAPEngine(GAEngine):
    NAMES = dict(stem_conv='downsample_layers.0.0.', stem_ln='downsample_layers.0.1.', ds_ln='downsample_layers.{i}.0.',
                 ds_conv='downsample_layers.{i}.1.', block='stages.{i}.{j}.', dw='dwconv.', fc1='pwconv1.', fc2='pwconv2.')

    def _drop_path_rates(self):
        """map_convnext.py:85 -- linspace over sum(depths) of the four stages"""
        dep, rate = self.cfg['depths'], self.cfg['drop_path_rate']
        pts = torch.linspace(0, rate, sum(dep)).split(list(dep))
        return {f'stages.{i}.{j}.': float(pts[i][j]) for i in range(4) for j in range(dep[i])}

    # ------------------------------------------------------------------------------------------
    def _build(self):
        cfg = self.cfg
        d = cfg['dims']
        B, T, F, dt, P = self.B, self.training, self.fwd, self.dt, self.P
        feats, taps, stage_in, x_stem = self._build_trunk()
        # ---------------- MultiScale (map.py:322-333) ----------------
        Hc = 14
        M4 = B * Hc * Hc
        L = cfg['last_dim']
        srcs = [(x_stem, self.img // 4, d[0])] + [(x, res, d[i]) for i, (x, res) in enumerate(feats)]
        ctot = sum(c for _, _, c in srcs)
        cat = self.act('ms.cat', (M4, ctot))
        off = 0
        self.agg_segs = []
        for src, hw, c in srcs:
            mode = 0 if hw == Hc else (2 if hw > Hc else 3)
            F.pool_concat_fwd(src, cat, B, hw, hw, c, Hc, Hc, ctot, off, mode, dt, label=f'agg.{off}')
            self.agg_segs.append((src, hw, c, mode, off))
            off += c
        x = self._multi_scale_conv_fwd(cat, M4, ctot)
        self._build_map_head(x, M4, Hc)
        if T:
            self._build_backward(feats, stage_in, x, M4)
            if self.async_wgrad:
                self.bwd.join_async()
            self.bwd.flush('end.')
        self.prep.flush('prep.')

    def _multi_scale_conv_fwd(self, cat, M4, ctot):
        """MultiScale.concat_conv (map.py:322-333): conv1x1 -> BN -> GELU on the concat of the resized maps"""
        F, dt, T, L = self.fwd, self.dt, self.training, self.cfg['last_dim']
        mp = 'head.mmcap.multi_scale.concat_conv.'
        Wc = self._w_plain(mp + '0.weight', L, ctot, 1, 1)
        ms = self.ms = dict(cat=cat, ctot=ctot, c=self.act('ms.c', (M4, L)), bn=self._bn_bufs(mp + '1.', L), z=self.act('ms.z', (M4, L)))
        F.gemm(cat, Wc, ms['c'], M4, L, ctot, dt, ldb=pad8(ctot), colsum=ms['bn']['s'] if T else None,
               colsumsq=ms['bn']['q'] if T else None, label=mp + 'conv')
        self._bn_finalize(mp + '1.', ms['bn'], M4, L)
        F.affine_act(ms['c'], ms['bn']['scale'], ms['bn']['shift'], None, ms['z'], M4, L, False, dt, label=mp + 'bn')
        x = self.buf('ms.x', (M4, L))
        F.gelu_fwd(ms['z'], x, M4 * L, dt, label=mp + 'gelu')
        return x

    # ------------------------------------------------------------------------------------------
    def _build_map_head(self, x, M4, Hc):
        cfg, B, T, F, dt, P = self.cfg, self.B, self.training, self.fwd, self.dt, self.P
        L, G, Tn, E, nh = cfg['last_dim'], cfg['n_groups'], cfg['n_tokens'], cfg['ca_dim'], cfg['num_heads']
        bp, NC = cfg['bp_dim'], cfg['num_classes']
        sdt = self.sdt = bool(cfg.get('self_distill_token', True))     # CAP's extra mean token + MAPHead.self_dt_heads (map.py:273-275,490-491)
        hfn = self.head_fn = cfg.get('head_fn', 'norm')                # heads: NormHead | SplitNormHead | nn.Linear (map.py:402-448,485-489)
        assert hfn in ('norm', 'split', 'linear')
        Tq = Tn + (1 if sdt else 0)      # gram tokens (+ the self-distillation mean token)
        HW = Hc * Hc
        hd = E // nh
        assert E % nh == 0 and hd % 8 == 0 and E <= 512 and L % 8 == 0 and bp % 8 == 0 and NC % 8 == 0 and Tq <= 8
        self.G, self.Tq = G, Tq
        H = self.mh = dict(x=x)
        nlog = 2 * G if (T and sdt) else G
        self.logits = self.buf('logits', (nlog, B, NC), torch.float32)
        # ---- dropout masks (training, p > 0): [attn | proj | mlp] per group in ONE fp32 buffer, refilled every step
        pd, pa = (cfg['head_drop'], cfg['head_attn_drop']) if T else (0.0, 0.0)
        N = Tq + HW
        n_attn, n_proj, n_mlp = B * Tq * nh * N, B * Tq * L, B * Tq * 4 * L
        self.drop = None
        if T and (pd > 0 or pa > 0):
            assert abs(pd - pa) < 1e-12, 'one keep probability for the whole mask buffer'
            per = n_attn + n_proj + n_mlp
            buf = torch.ones(G * per, device=self.dev)
            self.drop = dict(buf=buf, keep=1.0 - pd, counter=torch.zeros(1, dtype=torch.int64, device=self.dev))
            self.drop['plan'] = Plan(name='dropout')
            self.drop['plan'].dropout_mask_sample(buf, buf.numel(), 1.0 - pd, torch.initial_seed() + 0x5eed, self.drop['counter'])
            self.drop['views'] = [dict(attn=buf[k * per:k * per + n_attn].view(B, Tq, nh, N),
                                       proj=buf[k * per + n_attn:k * per + n_attn + n_proj].view(B * Tq, L),
                                       mlp=buf[k * per + n_attn + n_proj:(k + 1) * per].view(B * Tq, 4 * L)) for k in range(G)]
        # ---- the image rows of norm1(cat(x_cls, x_img)) are the same for every group: ONE LayerNorm, ONE stacked k|v GEMM
        tk = self.tok = dict(xn=self.act('ca.tok.xn', (M4, L)), rstd=self.act('ca.tok.rstd', (M4,), torch.float32))
        F.layernorm_fwd(x, None, None, tk['xn'], None, tk['rstd'], M4, L, 1e-6, dt, label='ca.tok.ln')
        E2 = 2 * E
        tk['E2'], tk['ld'] = E2, G * E2
        tk['W'] = self.buf('w.ca.kv_all', (G * E2, L))
        tk['WT'] = self.buf('wT.ca.kv_all', (L, G * E2)) if T else None
        tk['b'] = self.buf('w.ca.bkv_all', (G * E2,), torch.float32)
        # dim_mismatch (gram_dim != last_dim, map.py:85-90,165-177): the image rows have their own projections k2 | v2 and norm1_2,
        # the class rows (width gram_dim) q | k1 | v1 and norm1_1, and the attention output REPLACES the class rows
        mm = self.mm = cfg['gram_dim'] != L
        self.kv_img, self.n1_img = ('attn.k2', 'norm1_2') if mm else ('attn.k', 'norm1')
        for k in range(G):
            ap = f'head.mmcap.mmcap.{k}.attention.0.'
            kn, vn = ap + self.kv_img, ap + self.kv_img.replace('.k', '.v')
            pk, pv, bk, bv = P[kn + '.weight'], P[vn + '.weight'], P[kn + '.bias'], P[vn + '.bias']
            assert pv.data_ptr() == pk.data_ptr() + pk.numel() * 4 and bv.data_ptr() == bk.data_ptr() + bk.numel() * 4, \
                'k / v weights (and biases) must be adjacent in the flat buffer'
            self.prep.weight_prep(pk, 1, E2, L, 1, 1, dt, out=tk['W'][k * E2:], ldo=L, outT=tk['WT'][:, k * E2:] if T else None,
                                  ldt=G * E2 if T else 0, cs=P[ap + self.n1_img + '.weight'], t_cols=E2, label='prep.' + ap + 'kv')
            self.prep.bias_fold(pk, bk, None, P[ap + self.n1_img + '.bias'], tk['b'][k * E2:], E2, L)
        tk['kv'] = self.act('ca.kv_all', (M4, G * E2))
        F.gemm(tk['xn'], tk['W'], tk['kv'], M4, G * E2, L, dt, bias=tk['b'], label='ca.kv_all')
        # ---- ch_reduction (conv1x1, no bias) of all groups: ONE stacked GEMM; group k owns columns [k*bp, (k+1)*bp)
        gc = self.gcon = dict(ld=G * bp)
        gc['W'] = self.buf('w.ch_reduction.all', (G * bp, L))
        gc['WT'] = self.buf('wT.ch_reduction.all', (L, G * bp)) if T else None
        gc['s'], gc['q'] = self._bn_pool(G * bp), self._bn_pool(G * bp)
        for k in range(G):
            gp = f'head.mmcap.mmcap.{k}.gram_token_extraction.'
            self.prep.weight_prep(P[gp + 'ch_reduction.0.weight'], 1, bp, L, 1, 1, dt, out=gc['W'][k * bp:], ldo=L,
                                  outT=gc['WT'][:, k * bp:] if T else None, ldt=G * bp if T else 0, t_cols=bp, label='prep.' + gp + 'chr')
        gc['out'] = self.act('ch_reduction.all.out', (M4, G * bp))
        F.gemm(x, gc['W'], gc['out'], M4, G * bp, L, dt, colsum=gc['s'] if T else None, colsumsq=gc['q'] if T else None,
               label='ch_reduction.all')
        # ---- classifiers: heads (on T*L) of all groups as one batched GEMM, self_dt_heads (on L) as another
        fo = self.fc_org = dict(x=self.act('fc.org.x', (G, B, Tn * L)))
        fo['W'] = self.buf('w.fc.org', (G, NC, Tn * L))
        fo['WT'] = self.buf('wT.fc.org', (G, Tn * L, pad8(NC))) if T else None
        fo['b'] = self.buf('w.fc.org.b', (G, NC), torch.float32)
        for k in range(G):
            if hfn == 'split':
                # SplitNormHead (map.py:415-441): sum_t head_t(LayerNorm_t(token t)).  The T LayerNorms run as ONE affine-free
                # LayerNorm over the (B * T) token rows; their affine parts fold into the T linears, which then are one
                # [NC][T * L] operand (column block t = head_t.weight * gamma_t) with bias sum_t (head_t.weight beta_t + bias_t)
                fo['bt'] = self.buf('w.fc.org.bt', (G, Tn, NC), torch.float32)
                wt = [self.buf(f'w.fc.org.split.{k}.{t}', (NC, L)) for t in range(Tn)]      # (weight_prep pads an output row to ldo: it
                for t in range(Tn):                                                          # cannot write a column slice directly)
                    self.prep.weight_prep(P[f'head.heads.{k}.head.{t}.weight'], 1, NC, L, 1, 1, dt, out=wt[t], ldo=L,
                                          outT=fo['WT'][k][t * L:] if T else None, ldt=pad8(NC) if T else 0,
                                          cs=P[f'head.heads.{k}.norm.{t}.weight'], label=f'prep.heads.{k}.{t}')
                    self.prep.bias_fold(P[f'head.heads.{k}.head.{t}.weight'], P[f'head.heads.{k}.head.{t}.bias'], None,
                                        P[f'head.heads.{k}.norm.{t}.bias'], fo['bt'][k, t], NC, L)
                self.prep.flush('prep.split.')
                for t in range(Tn):
                    self.prep.copy2d(wt[t], L, fo['W'][k][:, t * L:], Tn * L, NC, L, dt, label=f'prep.heads.{k}.{t}.place')
                self.prep.zero(fo['b'][k], label=f'prep.heads.{k}.b0')
                for t in range(Tn):          # (one batch per term: the jobs of a batch run concurrently and these share a destination)
                    self.prep.axpy_f32(fo['b'][k], fo['bt'][k, t], 1.0, NC)
                    self.prep.flush(f'prep.split.sum{t}.')
                continue
            wn = f'head.heads.{k}.weight' if hfn == 'linear' else f'head.heads.{k}.head.weight'
            self.prep.weight_prep(P[wn], 1, NC, Tn * L, 1, 1, dt, out=fo['W'][k], ldo=Tn * L,
                                  outT=fo['WT'][k] if T else None, ldt=pad8(NC) if T else 0, label=f'prep.heads.{k}')
            self.prep.bias_fold(None, P[wn[:-6] + 'bias'], None, None, fo['b'][k], NC, Tn * L)
        if T and sdt:
            fa = self.fc_avg = dict(x=self.act('fc.avg.x', (G, B, L)))
            fa['W'] = self.buf('w.fc.avg', (G, NC, L))
            fa['WT'] = self.buf('wT.fc.avg', (G, L, pad8(NC)))
            fa['b'] = self.buf('w.fc.avg.b', (G, NC), torch.float32)
            for k in range(G):
                self.prep.weight_prep(P[f'head.self_dt_heads.{k}.head.weight'], 1, NC, L, 1, 1, dt, out=fa['W'][k], ldo=L,
                                      outT=fa['WT'][k], ldt=pad8(NC), label=f'prep.self_dt_heads.{k}')
                self.prep.bias_fold(None, P[f'head.self_dt_heads.{k}.head.bias'], None, None, fa['b'][k], NC, L)
        # ---- the groups: independent chains of small launches -> side lanes, own transients
        self.head_lanes = int(os.environ.get('GAEXT_HEAD_STREAMS', '4'))
        self.groups = []
        for k in range(G):
            if self.head_lanes > 1:
                F.lane, self.tmp_prefix = 1 + k % self.head_lanes, f'h{k}.'
            self.groups.append(self._group_fwd(k, M4, Hc))
        F.lane, self.tmp_prefix = 0, ''
        F.gemm(fo['x'], fo['W'], self.logits, B, NC, Tn * L, dt, batch=G, strideA=B * Tn * L, strideB=NC * Tn * L, strideC=B * NC,
               bias=fo['b'], strideBias=NC, c_f32=True, label='fc.org')
        if T and sdt:
            fa = self.fc_avg
            F.gemm(fa['x'], fa['W'], self.logits[G:], B, NC, L, dt, batch=G, strideA=B * L, strideB=NC * L, strideC=B * NC,
                   bias=fa['b'], strideBias=NC, c_f32=True, label='fc.avg')

    # ------------------------------------------------------------------------------------------
    def _group_fwd(self, k, M4, Hc):
        """one CAP (map.py:264-278): GramToken -> tokens (+ mean token) -> CABlock -> NormHead inputs"""
        cfg, B, T, F, dt, P = self.cfg, self.B, self.training, self.fwd, self.dt, self.P
        L, G, Tn, E, nh, mg = cfg['last_dim'], cfg['n_groups'], cfg['n_tokens'], cfg['ca_dim'], cfg['num_heads'], cfg['mlp_groups']
        bp, groups, gd = cfg['bp_dim'], cfg['gram_group'], cfg['gram_dim']
        Tq, HW, hd = self.Tq, Hc * Hc, cfg['ca_dim'] // cfg['num_heads']
        R = B * Tq                       # class rows
        h = dict(k=k)
        dm = self.drop['views'][k] if self.drop else {}
        gp = f'head.mmcap.mmcap.{k}.gram_token_extraction.'
        ap = f'head.mmcap.mmcap.{k}.attention.0.'
        gcn = self.gcon
        # --- ch_reduction BN on this group's column slice
        h['gc'] = gcn['out'][:, k * bp:]
        bn = dict(s=gcn['s'][k * bp:(k + 1) * bp], q=gcn['q'][k * bp:(k + 1) * bp],
                  mean=self.buf(gp + 'chr.bmean', (bp,), torch.float32), rstd=self.buf(gp + 'chr.brstd', (bp,), torch.float32),
                  scale=self.buf(gp + 'chr.scale', (bp,), torch.float32), shift=self.buf(gp + 'chr.shift', (bp,), torch.float32))
        h['bn_gc'] = bn
        self._bn_finalize(gp + 'ch_reduction.1.', bn, M4, bp)
        h['g0'] = self.buf(gp + 'g0', (M4, bp))
        F.affine_act(h['gc'], bn['scale'], bn['shift'], None, h['g0'], M4, bp, False, dt, ldx=gcn['ld'], label=gp + 'chr.bn')
        # --- Gram (x / hw on both factors, map.py:217-218) -> packed, normalised, token-interleaved vector
        h['alpha'] = 1.0 / (HW * HW)
        Gm = self.tmp('gramG', (B, bp, bp), torch.float32)
        F.wgrad(h['g0'], h['g0'], Gm, HW, bp, bp, dt, batch=B, strideY=HW * bp, strideX=HW * bp, strideW=bp * bp, split_m=1,
                accumulate=False, alpha=h['alpha'], label=f'gram.{k}')
        ntri = bp * (bp + 1) // 2
        assert ntri % groups == 0 and ntri % Tn == 0 and (gd * Tn) % groups == 0
        Kg = ntri // groups
        Kp = pad8(Kg)
        cg = gd * Tn // groups
        h.update(Kg=Kg, Kp=Kp, cg=cg)
        h['vec'] = self.act(f'gram.{k}.vec', (B, groups * Kp))
        h['inv'] = self.act(f'gram.{k}.inv', (B,), torch.float32)
        F.gram_pack_fwd2(Gm, h['vec'], h['inv'], B, bp, groups, Kp, Tn, dt, label=f'gram.{k}.pack')
        # --- bp_reduction: grouped 1x1 (no bias) + BN on (B, gd*T)
        Wemb = self._w_plain(gp + 'bp_reduction.0.weight', cg, Kg, 1, 1, groups=groups, ldo=Kp)
        h['e'] = self.act(gp + 'e', (B, gd * Tn))
        h['bn_e'] = self._bn_bufs(gp + 'bp_reduction.1.', gd * Tn)
        F.gemm(h['vec'], Wemb, h['e'], B, cg, Kp, dt, lda=groups * Kp, batch=groups, strideA=Kp, strideB=cg * Kp, ldc=gd * Tn,
               strideC=cg, colsum=h['bn_e']['s'] if T else None, colsumsq=h['bn_e']['q'] if T else None, strideCol=cg,
               label=gp + 'bpr')
        self._bn_finalize(gp + 'bp_reduction.1.', h['bn_e'], B, gd * Tn)
        e2 = self.tmp('e2', (B, gd * Tn))
        F.affine_act(h['e'], h['bn_e']['scale'], h['bn_e']['shift'], None, e2, B, gd * Tn, False, dt, label=gp + 'bpr.bn')
        mm = self.mm
        h['cls0'] = self.buf(ap + 'cls0', (R, gd))                 # [B][Tq][gram_dim]: gram tokens, then their mean
        F.map_tokens_fwd(e2, h['cls0'], B, gd, Tn, self.sdt, dt, label=gp + 'tokens')
        # --- CABlock: class rows normalised on their own (the norm's affine part folded into q / k|v)
        n1c = ap + ('norm1_1' if mm else 'norm1')
        g1, b1 = P[n1c + '.weight'], P[n1c + '.bias']
        h['cn'] = self.act(ap + 'cn', (R, gd))
        h['rc'] = self.act(ap + 'rc', (R,), torch.float32)
        F.layernorm_fwd(h['cls0'], None, None, h['cn'], None, h['rc'], R, gd, 1e-6, dt, label=ap + 'ln1c')
        tk = self.tok
        E2 = tk['E2']
        h['kvt'] = tk['kv'][:, k * E2:]
        h['kvc'] = self.act(ap + 'kvc', (R, E2))
        if mm:       # class rows: k1 | v1 (adjacent in the flat buffer) with norm1_1 folded
            pk, bk = P[ap + 'attn.k1.weight'], P[ap + 'attn.k1.bias']
            assert P[ap + 'attn.v1.weight'].data_ptr() == pk.data_ptr() + pk.numel() * 4
            assert P[ap + 'attn.v1.bias'].data_ptr() == bk.data_ptr() + bk.numel() * 4
            h['Wkvc'] = self.buf('w.' + ap + 'kv1', (E2, gd))
            h['WkvcT'] = self.buf('wT.' + ap + 'kv1', (gd, E2)) if T else None
            h['bkvc'] = self.buf('w.' + ap + 'bkv1', (E2,), torch.float32)
            self.prep.weight_prep(pk, 1, E2, gd, 1, 1, dt, out=h['Wkvc'], ldo=gd, outT=h['WkvcT'], ldt=E2 if T else 0, cs=g1,
                                  t_cols=E2, label='prep.' + ap + 'kv1')
            self.prep.bias_fold(pk, bk, None, b1, h['bkvc'], E2, gd)
            F.gemm(h['cn'], h['Wkvc'], h['kvc'], R, E2, gd, dt, bias=h['bkvc'], label=ap + 'kvc')
        else:
            F.gemm(h['cn'], tk['W'][k * E2:], h['kvc'], R, E2, L, dt, bias=tk['b'][k * E2:], label=ap + 'kvc')
        Wq = self._w_plain(ap + 'attn.q.weight', E, gd, 1, 1, cs=g1)
        bq = self.buf('w.' + ap + 'bq', (E,), torch.float32)
        self.prep.bias_fold(P[ap + 'attn.q.weight'], P[ap + 'attn.q.bias'], None, b1, bq, E, gd)
        h['q'] = self.act(ap + 'q', (R, E))
        F.gemm(h['cn'], Wq, h['q'], R, E, gd, dt, bias=bq, label=ap + 'q')
        h['ao'] = self.act(ap + 'ao', (R, E))
        h['P'] = self.act(ap + 'P', (B, Tq, nh, Tq + HW), torch.float32)
        h['scale'] = hd ** -0.5
        if cfg.get('interactive', False):    # head-mixing linears around the softmax (map.py:96-98,130-136): fp32 masters read directly
            F.class_attn_mt_ia_fwd(h['q'], h['kvc'], h['kvt'], tk['ld'], h['ao'], h['P'], dm.get('attn'), P[ap + 'attn.w1.weight'],
                                   P[ap + 'attn.w1.bias'], P[ap + 'attn.w2.weight'], P[ap + 'attn.w2.bias'], B, Tq, Tq + HW, nh, hd,
                                   h['scale'], dt, label=ap + 'attn')
        else:
            F.class_attn_mt_fwd(h['q'], h['kvc'], h['kvt'], tk['ld'], h['ao'], h['P'], dm.get('attn'), B, Tq, Tq + HW, nh, hd, h['scale'], dt,
                                label=ap + 'attn')
        Wpr = self._w_plain(ap + 'attn.proj.weight', L, E, 1, 1)
        h['cls1'] = self.buf(ap + 'cls1', (R, L))
        res0 = None if mm else h['cls0']         # (dim_mismatch: x_cls = attn(...), no residual, map.py:177)
        if 'proj' in dm:
            pr = self.tmp('proj', (R, L))
            F.gemm(h['ao'], Wpr, pr, R, L, E, dt, bias=P[ap + 'attn.proj.bias'], label=ap + 'proj')
            F.mask_mul(pr, dm['proj'], res0, h['cls1'], R * L, dt, label=ap + 'proj.drop')
        else:
            F.gemm(h['ao'], Wpr, h['cls1'], R, L, E, dt, bias=P[ap + 'attn.proj.bias'], R=res0, ldr=L if res0 is not None else 0,
                   label=ap + 'proj')
        h['t'] = self.act(ap + 't', (R, L))
        h['m2'] = self.act(ap + 'm2', (R,), torch.float32)
        h['r2'] = self.act(ap + 'r2', (R,), torch.float32)
        F.layernorm_fwd(h['cls1'], P[ap + 'norm2.weight'], P[ap + 'norm2.bias'], h['t'], h['m2'], h['r2'], R, L, 1e-6, dt,
                        label=ap + 'ln2')
        h['cls2'] = self.buf(ap + 'cls2', (R, L))
        h['mlp'] = self._gmlp_fwd(ap + 'mlp.', h['t'], R, L, mg, h['cls2'], h['cls1'], None, 1, act='relu', drop_mask=dm.get('mlp'))
        # --- NormHead inputs: the T gram tokens flattened (heads) / the mean token (self_dt_heads), each LayerNorm-ed (eps 1e-5)
        hp = f'head.heads.{k}.'
        if self.head_fn == 'linear':         # nn.Linear on the flattened tokens: no norm
            F.copy2d(h['cls2'], Tq * L, self.fc_org['x'][k], Tn * L, B, Tn * L, dt, label=hp + 'org')
        else:
            h['org'] = self.act(hp + 'org', (B, Tn * L))
            F.copy2d(h['cls2'], Tq * L, h['org'], Tn * L, B, Tn * L, dt, label=hp + 'org')
            if self.head_fn == 'split':      # one affine-free LayerNorm per token row (B * Tn rows of L); affine folded into the linears
                h['orr'] = self.act(hp + 'r', (B * Tn,), torch.float32)
                F.layernorm_fwd(h['org'], None, None, self.fc_org['x'][k], None, h['orr'], B * Tn, L, 1e-5, dt, label=hp + 'ln')
            else:
                h['om'], h['orr'] = self.act(hp + 'm', (B,), torch.float32), self.act(hp + 'r', (B,), torch.float32)
                F.layernorm_fwd(h['org'], P[hp + 'norm.weight'], P[hp + 'norm.bias'], self.fc_org['x'][k], h['om'], h['orr'], B, Tn * L, 1e-5,
                                dt, label=hp + 'ln')
        if T and self.sdt:
            sp = f'head.self_dt_heads.{k}.'
            h['avg'] = self.act(sp + 'avg', (B, L))
            F.copy2d(h['cls2'][:, :].view(B, Tq * L)[:, Tn * L:], Tq * L, h['avg'], L, B, L, dt, label=sp + 'avg')
            h['am_'], h['ar'] = self.act(sp + 'm', (B,), torch.float32), self.act(sp + 'r', (B,), torch.float32)
            F.layernorm_fwd(h['avg'], P[sp + 'norm.weight'], P[sp + 'norm.bias'], self.fc_avg['x'][k], h['am_'], h['ar'], B, L, 1e-5,
                            dt, label=sp + 'ln')
        return h

    # ------------------------------------------------------------------------------------------
    def _group_bwd(self, h, M4):
        cfg, B, Bk, dt, P, W = self.cfg, self.B, self.bwd, self.dt, self.P, self.W
        L, G, Tn, E, nh, mg = cfg['last_dim'], cfg['n_groups'], cfg['n_tokens'], cfg['ca_dim'], cfg['num_heads'], cfg['mlp_groups']
        bp, groups, gd = cfg['bp_dim'], cfg['gram_group'], cfg['gram_dim']
        k, Tq = h['k'], self.Tq
        HW = M4 // B
        hd = E // nh
        R = B * Tq
        dm = self.drop['views'][k] if self.drop else {}
        gp = f'head.mmcap.mmcap.{k}.gram_token_extraction.'
        ap = f'head.mmcap.mmcap.{k}.attention.0.'
        hp, sp = f'head.heads.{k}.', f'head.self_dt_heads.{k}.'
        # --- NormHeads -> gradient wrt cls2 [B][Tq][L]
        dcls2 = self.tmp('dcls2', (R, L))
        if self.head_fn == 'linear':
            Bk.copy2d(self.fc_org['dx'][k], Tn * L, dcls2, Tq * L, B, Tn * L, dt, label=hp + 'dorg')
        else:
            dorg = self.tmp('dorg', (B, Tn * L))
            if self.head_fn == 'split':      # fc_org['x'][k] holds xhat (affine-free): LayerNorm backward from the stored normalised rows
                Bk.layernorm_bwd(self.fc_org['dx'][k], self.fc_org['x'][k], None, h['orr'], None, None, dorg, None, None, B * Tn, L, True, dt,
                                 label=hp + 'lnb')
            else:
                Bk.layernorm_bwd(self.fc_org['dx'][k], h['org'], h['om'], h['orr'], P[hp + 'norm.weight'], None, dorg,
                                 self.grad(hp + 'norm.weight'), self.grad(hp + 'norm.bias'), B, Tn * L, False, dt, label=hp + 'lnb')
            Bk.copy2d(dorg, Tn * L, dcls2, Tq * L, B, Tn * L, dt, label=hp + 'dorg')
        if self.sdt:
            davg = self.tmp('davg', (B, L))
            Bk.layernorm_bwd(self.fc_avg['dx'][k], h['avg'], h['am_'], h['ar'], P[sp + 'norm.weight'], None, davg,
                             self.grad(sp + 'norm.weight'), self.grad(sp + 'norm.bias'), B, L, False, dt, label=sp + 'lnb')
            Bk.copy2d(davg, L, dcls2.view(B, Tq * L)[:, Tn * L:], Tq * L, B, L, dt, label=sp + 'davg')
        # --- GroupConvMlp + norm2
        dtk = self.tmp('dt', (R, L))
        self._gmlp_bwd(ap + 'mlp.', h['mlp'], dcls2, h['t'], R, L, mg, dtk)
        dcls1 = self.tmp('dcls1', (R, L))
        Bk.layernorm_bwd(dtk, h['cls1'], h['m2'], h['r2'], P[ap + 'norm2.weight'], dcls2, dcls1, self.grad(ap + 'norm2.weight'),
                         self.grad(ap + 'norm2.bias'), R, L, False, dt, label=ap + 'ln2b')
        # --- projection
        dpz = dcls1
        if 'proj' in dm:
            dpz = self.tmp('dpz', (R, L))
            Bk.mask_mul(dcls1, dm['proj'], None, dpz, R * L, dt, label=ap + 'proj.dropb')
        Bk.wgrad(dpz, h['ao'], self.grad(ap + 'attn.proj.weight'), R, L, E, dt, dbias=self.grad(ap + 'attn.proj.bias'),
                 label=ap + 'proj.wg')
        dao = self.tmp('dao', (R, E))
        Bk.gemm(dpz, W[ap + 'attn.proj.weight.T'], dao, R, E, L, dt, ldb=pad8(L), label=ap + 'proj.dg')
        # --- attention
        tk = self.tok
        E2 = tk['E2']
        dq = self.tmp('dq', (R, E))
        dkvc = self.tmp('dkvc', (R, E2))
        if cfg.get('interactive', False):
            Bk.class_attn_mt_ia_bwd(dao, h['q'], h['kvc'], h['kvt'], tk['ld'], h['P'], dm.get('attn'), P[ap + 'attn.w1.weight'],
                                    P[ap + 'attn.w2.weight'], P[ap + 'attn.w2.bias'], dq, dkvc, tk['dkv'][:, k * E2:], tk['ld'],
                                    self.grad(ap + 'attn.w1.weight'), self.grad(ap + 'attn.w1.bias'), self.grad(ap + 'attn.w2.weight'),
                                    self.grad(ap + 'attn.w2.bias'), B, Tq, Tq + HW, nh, hd, h['scale'], dt, label=ap + 'attnb')
        else:
            Bk.class_attn_mt_bwd(dao, h['q'], h['kvc'], h['kvt'], tk['ld'], h['P'], dm.get('attn'), dq, dkvc, tk['dkv'][:, k * E2:], tk['ld'],
                                 B, Tq, Tq + HW, nh, hd, h['scale'], dt, label=ap + 'attnb')
        mm = self.mm
        n1c = ap + ('norm1_1' if mm else 'norm1')
        g1, b1 = P[n1c + '.weight'], P[n1c + '.bias']
        dcn = self.tmp('dcn', (R, gd))
        if mm:       # class rows own k1 | v1: their weight gradient with the norm1_1 fold undone (both unfolds add into d norm1_1: atomics)
            Gkv, gbkv = self.gbuf((E2, gd)), self.gbuf((E2,))
            Bk.wgrad(dkvc, h['cn'], Gkv, R, E2, gd, dt, dbias=gbkv, label=ap + 'kvc.wg')
            Bk.weight_unfold(Gkv, gd, E2, gd, gb=gbkv, W=P[ap + 'attn.k1.weight'], b=P[ap + 'attn.k1.bias'], cs=g1, v=b1,
                             dW=self.grad(ap + 'attn.k1.weight'), db=self.grad(ap + 'attn.k1.bias'), d_cs=self.grad(n1c + '.weight'),
                             d_v=self.grad(n1c + '.bias'), label=ap + 'kv1.unf')
            Bk.gemm(dkvc, h['WkvcT'], dcn, R, gd, E2, dt, ldb=E2, label=ap + 'kvc.dg')
        else:
            # class rows' share of the effective k|v weight gradient (the image rows' share: one wgrad after the loop)
            Bk.wgrad(dkvc, h['cn'], tk['G'][k * E2:], R, E2, L, dt, dbias=tk['gb'][k * E2:], label=ap + 'kvc.wg')
            Bk.gemm(dkvc, tk['WT'][:, k * E2:], dcn, R, L, E2, dt, ldb=tk['ld'], label=ap + 'kvc.dg')
        Gq, gbq = self.gbuf((E, gd)), self.gbuf((E,))
        Bk.wgrad(dq, h['cn'], Gq, R, E, gd, dt, dbias=gbq, label=ap + 'q.wg')
        Bk.weight_unfold(Gq, gd, E, gd, gb=gbq, W=P[ap + 'attn.q.weight'], b=P[ap + 'attn.q.bias'], cs=g1, v=b1,
                         dW=self.grad(ap + 'attn.q.weight'), db=self.grad(ap + 'attn.q.bias'), d_cs=self.grad(n1c + '.weight'),
                         d_v=self.grad(n1c + '.bias'), label=ap + 'q.unf')
        Bk.gemm(dq, W[ap + 'attn.q.weight.T'], dcn, R, gd, E, dt, ldb=pad8(E), R=dcn, ldr=gd, label=ap + 'q.dg')
        if mm:       # dcls0 = LN'(dcn): the class rows entered the block through norm1_1 only
            dcls0 = self.tmp('dcls0', (R, gd))
            Bk.layernorm_bwd(dcn, h['cn'], None, h['rc'], None, None, dcls0, None, None, R, gd, True, dt, label=ap + 'ln1cb')
        else:        # dcls0 = dcls1 + LN'(dcn)
            dcls0 = dcls1
            Bk.layernorm_bwd(dcn, h['cn'], None, h['rc'], None, dcls1, dcls1, None, None, R, L, True, dt, label=ap + 'ln1cb')
        # --- tokens -> bp_reduction BN + grouped conv
        de2 = self.tmp('de2', (B, gd * Tn))
        Bk.map_tokens_bwd(dcls0, de2, B, gd, Tn, self.sdt, dt, label=gp + 'tokensb')
        de = self.tmp('de', (B, gd * Tn))
        self._bn_bwd(gp + 'bp_reduction.1.', h['bn_e'], de2, None, h['e'], de, B, gd * Tn)
        Kg, Kp, cg = h['Kg'], h['Kp'], h['cg']
        Bk.wgrad(de, h['vec'], self.grad(gp + 'bp_reduction.0.weight'), B, cg, Kg, dt, ldy=gd * Tn, ldx=groups * Kp, ldw=Kg,
                 batch=groups, strideY=cg, strideX=Kp, strideW=cg * Kg, label=gp + 'bpr.wg')
        dvec = self.buf(f'gram.{k}.dvec', (B, groups * Kp), zero=True)      # pad columns stay zero
        Bk.gemm(de, W[gp + 'bp_reduction.0.weight.T'], dvec, B, Kg, cg, dt, lda=gd * Tn, batch=groups, strideA=cg,
                strideB=Kg * pad8(cg), ldb=pad8(cg), ldc=groups * Kp, strideC=Kp, label=gp + 'bpr.dg')
        S = self.tmp('gramS', (B, bp, bp))
        Bk.gram_pack_bwd2(dvec, h['vec'], h['inv'], S, B, bp, groups, Kp, Tn, dt, label=f'gram.{k}.packb')
        dg0 = self.tmp('dg0', (M4, bp))
        Bk.gemm(h['g0'], S, dg0, HW, bp, bp, dt, batch=B, strideA=HW * bp, strideB=bp * bp, strideC=HW * bp, alpha=h['alpha'],
                label=f'gram.{k}.dx')
        gcn = self.gcon
        self._bn_bwd(gp + 'ch_reduction.1.', h['bn_gc'], dg0, None, h['gc'], gcn['dout'][:, k * bp:], M4, bp, ldx=gcn['ld'],
                     lddx=gcn['ld'])

    # ------------------------------------------------------------------------------------------
    def _build_backward(self, feats, stage_in, x, M4):
        """head backward, then the ConvNeXt trunk's"""
        Bk, dt, B = self.bwd, self.dt, self.B
        dcat = self._build_head_backward(x, M4)
        ctot = self.ms['ctot']
        seeds = []
        for src, hw, c, mode, off in self.agg_segs:
            ds = self.buf(f'agg.d{off}', (B * hw * hw, c))
            Bk.pool_concat_bwd(dcat, None, ds, B, hw, hw, c, 14, 14, ctot, off, mode, dt, label=f'agg.b{off}')
            seeds.append(ds)
        self._build_trunk_backward({i: seeds[1 + i] for i in range(4)}, [], [], feats, stage_in, stem_seed=seeds[0])

    def _build_head_backward(self, x, M4):
        """classifiers, the groups, ch_reduction / k|v stacks, MultiScale conv: everything of head.* ; returns dcat [M4, ctot], the
        gradient of the multi-scale concat (the trunk-specific part takes it from there)"""
        Bk, dt, B, P, W, cfg = self.bwd, self.dt, self.B, self.P, self.W, self.cfg
        L, G, Tn, E, NC, bp = cfg['last_dim'], cfg['n_groups'], cfg['n_tokens'], cfg['ca_dim'], cfg['num_classes'], cfg['bp_dim']
        sdt, hfn = self.sdt, self.head_fn
        self.dlogits = self.buf('dlogits', ((2 if sdt else 1) * G, B, NC))
        Bk.zero(self.arena, label='zero.arena')
        # classifiers: one batched wgrad + dgrad for the heads, one for the self_dt_heads
        fcs = [(self.fc_org, self.dlogits[:G], 'heads', Tn * L)] + ([(self.fc_avg, self.dlogits[G:], 'self_dt_heads', L)] if sdt else [])
        for fc, dl, name, cin in fcs:
            Gfc, gbfc = self.gbuf((G, NC, cin)), self.gbuf((G, NC))
            Bk.wgrad(dl, fc['x'], Gfc, B, NC, cin, dt, batch=G, strideY=B * NC, strideX=B * cin, strideW=NC * cin, dbias=gbfc,
                     strideDbias=NC, label=f'fc.{name}.wg')
            for k in range(G):
                if name == 'heads' and hfn == 'split':      # undo the LayerNorm fold per token: dW_t, d(bias_t), d(gamma_t), d(beta_t)
                    for t in range(Tn):
                        hk = f'head.heads.{k}.'
                        Bk.weight_unfold(Gfc[k][:, t * L:], Tn * L, NC, L, gb=gbfc[k], W=P[hk + f'head.{t}.weight'], b=P[hk + f'head.{t}.bias'],
                                         cs=P[hk + f'norm.{t}.weight'], v=P[hk + f'norm.{t}.bias'], dW=self.grad(hk + f'head.{t}.weight'),
                                         db=self.grad(hk + f'head.{t}.bias'), d_cs=self.grad(hk + f'norm.{t}.weight'),
                                         d_v=self.grad(hk + f'norm.{t}.bias'), label=hk + f'{t}.unf')
                    continue
                wn = f'head.{name}.{k}.weight' if (name == 'heads' and hfn == 'linear') else f'head.{name}.{k}.head.weight'
                Bk.axpy_f32(self.grad(wn), Gfc[k], 1.0, NC * cin)
                Bk.axpy_f32(self.grad(wn[:-6] + 'bias'), gbfc[k], 1.0, NC)
            fc['dx'] = self.tmp(f'dfc.{name}', (G, B, cin))
            Bk.gemm(dl, fc['WT'], fc['dx'], B, cin, NC, dt, batch=G, strideA=B * NC, strideB=cin * pad8(NC), ldb=pad8(NC),
                    strideC=B * cin, label=f'fc.{name}.dg')
        tk = self.tok
        tk['dkv'] = self.tmp('dkv_all', (M4, tk['ld']))
        tk['G'], tk['gb'] = self.gbuf((tk['ld'], L)), self.gbuf((tk['ld'],))
        self.gcon['dout'] = self.tmp('dgc_all', (M4, self.gcon['ld']))
        for k in range(G):
            if self.head_lanes > 1:
                Bk.lane, self.tmp_prefix = 1 + k % self.head_lanes, f'h{k}.'
            self._group_bwd(self.groups[k], M4)
        Bk.lane, self.tmp_prefix = 0, ''
        # ch_reduction convs of all groups: one wgrad, one dgrad (first writer of dx)
        gcn = self.gcon
        Gc = self.gbuf((gcn['ld'], L))
        with self._wlane():
            Bk.wgrad(gcn['dout'], x, Gc, M4, gcn['ld'], L, dt, label='ch_reduction.all.wg')
        for k in range(G):
            Bk.axpy_f32(self.grad(f'head.mmcap.mmcap.{k}.gram_token_extraction.ch_reduction.0.weight'), Gc[k * bp:], 1.0, bp * L)
        dx = self.tmp('dx_ms', (M4, L))
        Bk.gemm(gcn['dout'], gcn['WT'], dx, M4, L, gcn['ld'], dt, label='ch_reduction.all.dg')
        # image rows of all groups: effective k|v weight gradients, each group's norm1 fold undone; dx += LN'(...)
        E2 = tk['E2']
        with self._wlane():
            Bk.wgrad(tk['dkv'], tk['xn'], tk['G'], M4, tk['ld'], L, dt, dbias=tk['gb'], label='ca.kv_all.wg')
        for k in range(G):
            ap = f'head.mmcap.mmcap.{k}.attention.0.'
            kn, nn_ = ap + self.kv_img, ap + self.n1_img
            Bk.weight_unfold(tk['G'][k * E2:], L, E2, L, gb=tk['gb'][k * E2:], W=P[kn + '.weight'], b=P[kn + '.bias'],
                             cs=P[nn_ + '.weight'], v=P[nn_ + '.bias'], dW=self.grad(kn + '.weight'),
                             db=self.grad(kn + '.bias'), d_cs=self.grad(nn_ + '.weight'), d_v=self.grad(nn_ + '.bias'),
                             label=ap + 'kv.unf')
        dxt = self.tmp('dxn_tok', (M4, L))
        Bk.gemm(tk['dkv'], tk['WT'], dxt, M4, L, tk['ld'], dt, label='ca.kv_all.dg')
        Bk.layernorm_bwd(dxt, tk['xn'], None, tk['rstd'], None, dx, dx, None, None, M4, L, True, dt, label='ca.tok.lnb')
        # MultiScale: GELU, BN, conv1x1
        ms = self.ms
        mp = 'head.mmcap.multi_scale.concat_conv.'
        dz = self.tmp('dz_ms', (M4, L))
        Bk.gelu_bwd(dx, ms['z'], dz, M4 * L, dt, label=mp + 'gelub')
        dc = self.tmp('dc_ms', (M4, L))
        self._bn_bwd(mp + '1.', ms['bn'], dz, None, ms['c'], dc, M4, L)
        ctot = ms['ctot']
        with self._wlane():
            Bk.wgrad(dc, ms['cat'], self.grad(mp + '0.weight'), M4, L, ctot, dt, label=mp + 'wg')
        dcat = self.tmp('dcat', (M4, ctot))
        Bk.gemm(dc, W[mp + '0.weight.T'], dcat, M4, ctot, L, dt, ldb=pad8(L), label=mp + 'dg')
        if self.async_wgrad:
            Bk.join_async()
        Bk.flush('heads.')
        Bk.mark('heads')      # every gradient of head.* is final here
        return dcat

    # ------------------------------------------------------------------------------------------
    def forward(self, x):
        if self.training and self.drop and not getattr(self, 'fixed_masks', False):
            self.drop['plan'].run()
        return super().forward(x)

    def set_dropout_masks(self, masks):
        """parity tests: masks[k] = dict(attn [B,heads,Tq,N], proj [B,Tq,L], mlp [B,Tq,4L]) in the REFERENCE's layouts
        (the MLP mask in fc1's output-channel order, before channel_shuffle)"""
        for k, m in masks.items():
            v = self.drop['views'][k]
            v['attn'].copy_(m['attn'].permute(0, 2, 1, 3).to(self.dev))
            v['proj'].copy_(m['proj'].reshape(v['proj'].shape).to(self.dev))
            perm = self.groups[k]['mlp']['perm'].long()
            v['mlp'].copy_(m['mlp'].reshape(v['mlp'].shape).to(self.dev)[:, perm])

    def _loss_operands(self):
        """fused MAP loss (MAP/train.py:792-839): org logits of the G groups, their avg logits, and the two gradients"""
        G = self.G
        if not self.sdt:         # plain list of group logits: multi_group_loss reduces to the GA form (MAP/train.py:818-820,824-837)
            return self.logits[:G], None, self.dlogits[:G], None, G
        return self.logits[:G], self.logits[G:], self.dlogits[:G], self.dlogits[G:], G
