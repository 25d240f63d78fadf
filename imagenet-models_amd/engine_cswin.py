"""CSWinEngine: the static launch plans (weight prep / forward / backward) of one GA_CSWinTransformer for a fixed
(batch, train|eval, math mode).  Shares buffers, BatchNorm / GroupConvMlp helpers and the whole GA head with
engine.GAEngine; the trunk is restated here from /root/reference/GA/ga_cswin.py:

  deep stem :463-477 ........ NCHW fp32 -> NHWC8 pack, three 3x3 convs as gather GEMMs (s2, s1, s2), LayerNorm+GELU fused
  CSWinBlock :191-212 ....... LN1 (affine folded into qkv) -> qkv GEMM -> stripe attention + LePE (one kernel for both
                              branches, windows addressed in place) -> proj GEMM (+DropPath, +residual) -> LN2 (folded)
                              -> fc1+GELU(+GELU') -> fc2 (+DropPath, +residual);  GroupConvMlp when mlp_groups > 1
  Merge_Block :253-268 ...... 3x3 s2 gather GEMM (+bias) -> LayerNorm; data gradient as a transposed-conv GEMM
  aggregation :666-669 ...... avg-pool / taps / bilinear x2 into one concat buffer (ga_pool_concat)
  stage5 :531-542 ........... Merge_Block_LCF (1x1) + CSWinBlock, or the SE-Bottleneck
  heads :677-692 ............ grouped (g=8) gram_contraction + BN -> CSWinBlock(192, 6 heads) -> Gram -> grouped embed + BN
                              -> class attention (q/k/v width C/4) -> fc

Saved for backward per CSWinBlock: xn1 (LN1 output without affine), rstd1, qkv, the attention output, xn2, rstd2,
a = gelu(h), g = gelu'(h), the block output.
"""
import os

import torch

from . import ops
from .engine import GAEngine, pad8
from .ops import A_CONV3, A_CONV3S2, A_NEIGH2, ACT_GELU, C_UNPATCH2


def tap_after(nblocks, naggre):
    """ga_cswin.py:659 -- 1-based block counts of stage3 after which a tap is taken"""
    step = nblocks // (naggre + 1)
    taps = []
    for b in range(1, nblocks + 1):
        if b % step == 0 and len(taps) < naggre:
            taps.append(b)
    return taps


class CSWinEngine(GAEngine):
    # ------------------------------------------------------------------------------------------
    def _drop_path_rates(self):
        """ga_cswin.py:486 (linspace over the 4 trunk stages), :538,571 (stage5 and gram layers: dpr[-1]), :541 (Bottleneck:
        drop_path_rate).  A CSWinBlock applies DropPath twice (attention branch, MLP branch: :209-210) -- two sites '#1', '#2'."""
        cfg = self.cfg
        dep, rate = cfg['depth'], cfg['drop_path_rate']
        dpr = torch.linspace(0, rate, sum(dep)).tolist()
        out, i = {}, 0
        for si in range(4):
            for j in range(dep[si]):
                out[f'stage{si + 1}.{j}.#1'] = out[f'stage{si + 1}.{j}.#2'] = dpr[i]
                i += 1
        if cfg['stage5'] == 'CSWin':
            out['stage5.2.#1'] = out['stage5.2.#2'] = dpr[-1]
        else:
            out['stage5.'] = float(rate)
        for k in range(cfg['branches']):
            out[f'gram_layer.{k}.1.#1'] = out[f'gram_layer.{k}.1.#2'] = dpr[-1]
            out[f'ga.{k}.'] = 0.0
        return out

    # ------------------------------------------------------------------------------------------
    # CSWinBlock
    # ------------------------------------------------------------------------------------------
    def _cs_block_fwd(self, pre, x, mod):
        """x [B*reso*reso, C] -> block output (same shape); mod = the block's parameter holder (geometry).  Inside a forward chain
        (GAEngine._chains: the trunk recorded once per batch part, each part on its own lane) the launches cover the chain's rows of
        the same full-batch buffers; weight preparation and the saved state are recorded by the first pass only"""
        F, dt, B, P, T = self.fwd, self.dt, self.B, self.P, self.training
        C, reso, heads, mg = mod.dim, mod.reso, mod.num_heads, mod.mlp_groups
        HW = reso * reso
        M = B * HW
        assert C % 8 == 0
        (lane, r0, r1, b0, b1), = self._fsplits(HW)
        first = pre not in self.blocks
        assert mg == 1 or (r0 == 0 and r1 == M), 'grouped-MLP blocks are recorded for the whole batch'
        if getattr(self, '_chain', None) is not None:      # (outside a chain the caller's lane stands: the heads' gram_layer blocks)
            F.lane = lane
        dp1, dp2 = self.dp_scale.get(pre + '#1'), self.dp_scale.get(pre + '#2')
        dp1c = dp1[b0:b1] if dp1 is not None else None
        dp2c = dp2[b0:b1] if dp2 is not None else None
        st = self.blocks.setdefault(pre, dict(x=x, mod=mod, M=M))
        # --- attention branch
        st['xn1'] = self.blk_act(pre + 'xn1', (M, C))
        st['r1'] = self.blk_act(pre + 'r1', (M,), torch.float32)
        F.layernorm_fwd(x[r0:r1], None, None, st['xn1'][r0:r1], None, st['r1'][r0:r1], r1 - r0, C, 1e-5, dt, label=pre + 'ln1')
        Wqkv = self._w_plain(pre + 'qkv.weight', 3 * C, C, 1, 1, cs=P[pre + 'norm1.weight'])
        bq = self.buf('w.' + pre + 'bqkv', (3 * C,), torch.float32)
        if first:
            self.prep.bias_fold(P[pre + 'qkv.weight'], P.get(pre + 'qkv.bias'), None, P[pre + 'norm1.bias'], bq, 3 * C, C)
        st['qkv'] = self.blk_act(pre + 'qkv', (M, 3 * C))
        F.gemm(st['xn1'][r0:r1], Wqkv, st['qkv'][r0:r1], r1 - r0, 3 * C, C, dt, bias=bq, label=pre + 'qkv')
        st['att'] = self.blk_act(pre + 'att', (M, C))
        lepe = [(P[pre + f'attns.{i}.get_v.weight'], P[pre + f'attns.{i}.get_v.bias']) for i in range(mod.branch_num)]
        if first:      # the backward's descriptor: the whole batch
            st['desc'] = F.cswin_desc(st['qkv'], st['att'], B, reso, C, heads, mod.stripes(), lepe, (C // heads) ** -0.5, dt)
            self._lepe_ws_elems = max(getattr(self, '_lepe_ws_elems', 0), ops.cswin_attn_bwd_workspace(st['desc']) // 4)
        dfw = st['desc'] if (r0 == 0 and r1 == M) else F.cswin_desc(st['qkv'][r0:r1], st['att'][r0:r1], b1 - b0, reso, C, heads,
                                                                       mod.stripes(), lepe, (C // heads) ** -0.5, dt)
        F.cswin_attn_fwd(dfw, label=pre + 'attn')
        Wp = self._w_plain(pre + 'proj.weight', C, C, 1, 1)
        # x1 is re-read by the affine LayerNorm backward of the grouped-MLP form only
        x1 = st['x1'] = self.buf(pre + 'x1', (M, C)) if (mg > 1 and T) else self.tmp('x1', (M, C))
        F.gemm(st['att'][r0:r1], Wp, x1[r0:r1], r1 - r0, C, C, dt, bias=P[pre + 'proj.bias'], rowscale=dp1c, rows_per_scale=HW,
               R=x[r0:r1], ldr=C, label=pre + 'proj')
        # --- MLP branch
        y = st['y'] = self.buf(pre + 'y', (M, C))
        if mg == 1:
            st['xn2'] = self.blk_act(pre + 'xn2', (M, C))
            st['r2'] = self.blk_act(pre + 'r2', (M,), torch.float32)
            F.layernorm_fwd(x1[r0:r1], None, None, st['xn2'][r0:r1], None, st['r2'][r0:r1], r1 - r0, C, 1e-5, dt, label=pre + 'ln2')
            W1 = self._w_plain(pre + 'mlp.fc1.weight', 4 * C, C, 1, 1, cs=P[pre + 'norm2.weight'])
            b1e = self.buf('w.' + pre + 'b1e', (4 * C,), torch.float32)
            if first:
                self.prep.bias_fold(P[pre + 'mlp.fc1.weight'], P[pre + 'mlp.fc1.bias'], None, P[pre + 'norm2.bias'], b1e, 4 * C, C)
            st['a'] = self.blk_act(pre + 'a', (M, 4 * C))
            st['g'] = self.buf(pre + 'g', (M, 4 * C)) if T else None
            F.gemm(st['xn2'][r0:r1], W1, st['a'][r0:r1], r1 - r0, 4 * C, C, dt, bias=b1e, act=ACT_GELU,
                   C2=st['g'][r0:r1] if T else None, c2_mode=2 if T else 0, label=pre + 'fc1')
            W2 = self._w_plain(pre + 'mlp.fc2.weight', C, 4 * C, 1, 1)
            F.gemm(st['a'][r0:r1], W2, y[r0:r1], r1 - r0, C, 4 * C, dt, bias=P[pre + 'mlp.fc2.bias'], rowscale=dp2c, rows_per_scale=HW,
                   R=x1[r0:r1], ldr=C, label=pre + 'fc2')
        else:
            st['t'] = self.blk_act(pre + 't', (M, C))
            st['m2'] = self.blk_act(pre + 'm2', (M,), torch.float32)
            st['r2'] = self.blk_act(pre + 'r2', (M,), torch.float32)
            F.layernorm_fwd(x1, P[pre + 'norm2.weight'], P[pre + 'norm2.bias'], st['t'], st['m2'], st['r2'], M, C, 1e-5, dt,
                            label=pre + 'ln2')
            st['mlp'] = self._gmlp_fwd(pre + 'mlp.', st['t'], M, C, mg, y, x1, dp2, HW)
        return y

    def _cs_block_bwd(self, pre, dy, dx, next_pre=None):
        """dy: gradient wrt the block output; writes dx (a different buffer) = gradient wrt the block input"""
        Bk, dt, B, P, W = self.bwd, self.dt, self.B, self.P, self.W
        st = self.blocks[pre]
        mod, M = st['mod'], st['M']
        C, reso, mg = mod.dim, mod.reso, mod.mlp_groups
        HW = reso * reso
        dp1, dp2 = self.dp_scale.get(pre + '#1'), self.dp_scale.get(pre + '#2')
        # trunk blocks: the weight-gradient launches go to the plan's asynchronous lane (nothing on the dgrad chain reads their
        # results); two sets of the transients they read + three rotating dx buffers (caller), so this block only waits for the
        # asynchronous launches of the block before the previous one (same scheme as GAEngine._block_bwd)
        side = self.async_wgrad and Bk.lane == 0
        par = ''
        if side:
            self._bwd_seq += 1
            par = str(self._bwd_seq & 1)
            Bk.join_async(f'blk{self._bwd_seq - 2}')
        dyz = dy
        if dp2 is not None:
            dyz = self._pre_dyz.pop(pre, None)           # written by the LayerNorm backward of the block before (backward order)
            if dyz is None:
                dyz = self.tmp('dyz' + par, (M, C))
                Bk.rowscale(dy, dp2, dyz, M * C, HW * C, dt, label=pre + 'dp2')
        dx1 = self.tmp('dx1' + par, (M, C))
        dx1z = dx1
        fuse1 = dp1 is not None and self.fuse_dp and mg == 1
        if fuse1:
            dx1z = self.tmp('dx1z' + par, (M, C))
        if mg == 1:
            with self._wlane():
                Bk.wgrad(dyz, st['a'], self.grad(pre + 'mlp.fc2.weight'), M, C, 4 * C, dt, dbias=self.grad(pre + 'mlp.fc2.bias'),
                         label=pre + 'wg2')
            dh = self.tmp('dh' + par, (M, 4 * C))
            gb1 = self.gbuf((4 * C,))
            Bk.gemm(dyz, W[pre + 'mlp.fc2.weight.T'], dh, M, 4 * C, C, dt, ldb=pad8(C), H=st['g'], ldh=4 * C, h_is_deriv=True,
                    colsum=gb1, label=pre + 'dg2')
            G1 = self.gbuf((4 * C, C))
            with self._wlane():
                Bk.wgrad(dh, st['xn2'], G1, M, 4 * C, C, dt, label=pre + 'wg1')
            gx = self.tmp('g', (M, C))
            Bk.gemm(dh, W[pre + 'mlp.fc1.weight.T'], gx, M, C, 4 * C, dt, ldb=pad8(4 * C), label=pre + 'dg1')
            if fuse1:       # the DropPath-scaled copy of dx1 rides on the LayerNorm backward that writes dx1
                Bk.layernorm_bwd(gx, st['xn2'], None, st['r2'], None, dy, dx1, None, None, M, C, True, dt, label=pre + 'ln2b', dx2=dx1z,
                                 scale2=dp1, rows_per_scale=HW)
            else:
                Bk.layernorm_bwd(gx, st['xn2'], None, st['r2'], None, dy, dx1, None, None, M, C, True, dt, label=pre + 'ln2b')
            Bk.weight_unfold(G1, C, 4 * C, C, gb=gb1, W=P[pre + 'mlp.fc1.weight'], b=P[pre + 'mlp.fc1.bias'],
                             cs=P[pre + 'norm2.weight'], v=P[pre + 'norm2.bias'], dW=self.grad(pre + 'mlp.fc1.weight'),
                             db=self.grad(pre + 'mlp.fc1.bias'), d_cs=self.grad(pre + 'norm2.weight'),
                             d_v=self.grad(pre + 'norm2.bias'), label=pre + 'unf1')
        else:
            dtk = self.tmp('dtk', (M, C))
            self._gmlp_bwd(pre + 'mlp.', st['mlp'], dyz, st['t'], M, C, mg, dtk)
            Bk.layernorm_bwd(dtk, st['x1'], st['m2'], st['r2'], P[pre + 'norm2.weight'], dy, dx1, self.grad(pre + 'norm2.weight'),
                             self.grad(pre + 'norm2.bias'), M, C, False, dt, label=pre + 'ln2b')
        # --- attention branch
        if dp1 is not None and not fuse1:
            dx1z = self.tmp('dx1z' + par, (M, C))
            Bk.rowscale(dx1, dp1, dx1z, M * C, HW * C, dt, label=pre + 'dp1')
        with self._wlane():
            Bk.wgrad(dx1z, st['att'], self.grad(pre + 'proj.weight'), M, C, C, dt, dbias=self.grad(pre + 'proj.bias'),
                     label=pre + 'proj.wg')
        datt = self.tmp('datt' + par, (M, C))
        Bk.gemm(dx1z, W[pre + 'proj.weight.T'], datt, M, C, C, dt, ldb=pad8(C), label=pre + 'proj.dg')
        dqkv = self.tmp('dqkv' + par, (M, 3 * C))
        lepe_grads = [(self.grad(pre + f'attns.{i}.get_v.weight'), self.grad(pre + f'attns.{i}.get_v.bias'))
                      for i in range(mod.branch_num)]
        need = ops.cswin_attn_bwd_workspace(st['desc'])
        if need:    # the MFMA kernel leaves the LePE weight-gradient partials of its LDS tiles; the reduce follows in stream order
            lws = self.tmp('lepe_ws', (self._lepe_ws_elems,), torch.float32)
            assert need <= lws.numel() * 4
            Bk.cswin_attn_bwd(st['desc'], datt, dqkv, lepe_ws=lws, label=pre + 'attnb')
            Bk.cswin_lepe_wgrad_reduce(st['desc'], lws, lepe_grads, label=pre + 'lepe.red')
        else:
            Bk.cswin_attn_bwd(st['desc'], datt, dqkv, label=pre + 'attnb')
            with self._wlane():
                Bk.cswin_lepe_wgrad(st['desc'], datt, lepe_grads, label=pre + 'lepe.wg')
        Gq, gbq = self.gbuf((3 * C, C)), self.gbuf((3 * C,))
        with self._wlane():
            Bk.wgrad(dqkv, st['xn1'], Gq, M, 3 * C, C, dt, dbias=gbq, label=pre + 'qkv.wg')
        has_b = (pre + 'qkv.bias') in P
        Bk.weight_unfold(Gq, C, 3 * C, C, gb=gbq, W=P[pre + 'qkv.weight'], b=P.get(pre + 'qkv.bias'), cs=P[pre + 'norm1.weight'],
                         v=P[pre + 'norm1.bias'], dW=self.grad(pre + 'qkv.weight'), db=self.grad(pre + 'qkv.bias') if has_b else None,
                         d_cs=self.grad(pre + 'norm1.weight'), d_v=self.grad(pre + 'norm1.bias'), label=pre + 'qkv.unf')
        gq = self.tmp('g', (M, C))
        Bk.gemm(dqkv, W[pre + 'qkv.weight.T'], gq, M, C, 3 * C, dt, ldb=pad8(3 * C), label=pre + 'qkv.dg')
        ndp = self.dp_scale.get(next_pre + '#2') if next_pre is not None else None
        if ndp is not None and self.fuse_dp:
            npar = str((self._bwd_seq + 1) & 1) if side else ''
            ndyz = self.tmp('dyz' + npar, (M, C))
            if side:           # that buffer was block t-1's: its asynchronous weight gradients must be done before it is rewritten
                Bk.join_async(f'blk{self._bwd_seq - 1}')
            Bk.layernorm_bwd(gq, st['xn1'], None, st['r1'], None, dx1, dx, None, None, M, C, True, dt, label=pre + 'ln1b', dx2=ndyz,
                             scale2=ndp, rows_per_scale=HW)
            self._pre_dyz[next_pre] = ndyz
        else:
            Bk.layernorm_bwd(gq, st['xn1'], None, st['r1'], None, dx1, dx, None, None, M, C, True, dt, label=pre + 'ln1b')
        if side:
            Bk.async_mark(f'blk{self._bwd_seq}')

    # ------------------------------------------------------------------------------------------
    # 3x3 / stride-2 conv (+bias) -> LayerNorm   (Merge_Block; the last stem conv without bias)
    # ------------------------------------------------------------------------------------------
    def _conv3s2_ln_fwd(self, cname, nname, x, Hin, Cin, Cout, bias):
        F, dt, B, P, T = self.fwd, self.dt, self.B, self.P, self.training
        Ho = Hin // 2
        Mo = B * Ho * Ho
        (lane, r0, r1, b0, b1), = self._fsplits(Ho * Ho)           # output rows of this forward chain (whole images)
        i0, i1 = b0 * Hin * Hin, b1 * Hin * Hin
        if getattr(self, '_chain', None) is not None:
            F.lane = lane
        Wf = self._w_plain(cname + 'weight', Cout, Cin, 3, 3, need_T=False)
        st = dict(x=x, Hin=Hin, Cin=Cin, Cout=Cout, cname=cname, nname=nname, bias=bias)
        if T:
            first = ('wD.' + cname) not in self.bufs
            st['Bt'] = self.buf('wD.' + cname, (4 * Cin, pad8(4 * Cout)))
            if first:
                self.prep.conv3s2_dgrad_prep(P[cname + 'weight'], st['Bt'], Cout, Cin, pad8(4 * Cout), dt, label='prep.' + cname + 'dgrad')
        st['c'] = self.act(cname + 'out', (Mo, Cout))
        F.gemm(x[i0:i1], Wf, st['c'][r0:r1], r1 - r0, Cout, 9 * Cin, dt, ldb=pad8(9 * Cin), a_kind=A_CONV3S2, a_dims=(Hin, Hin, Cin),
               bias=P[cname + 'bias'] if bias else None, label=cname + 'conv')
        st['mean'] = self.act(nname + 'mean', (Mo,), torch.float32)
        st['rstd'] = self.act(nname + 'rstd', (Mo,), torch.float32)
        y = self.buf(nname + 'out', (Mo, Cout))
        F.layernorm_fwd(st['c'][r0:r1], P[nname + 'weight'], P[nname + 'bias'], y[r0:r1], st['mean'][r0:r1], st['rstd'][r0:r1], r1 - r0,
                        Cout, 1e-5, dt, label=nname + 'ln')
        return y, st

    def _conv3s2_ln_bwd(self, st, dy, dprev, seed=None):
        """dy: gradient wrt the LayerNorm output; dprev (+= seed) = gradient wrt the conv input"""
        Bk, dt, B, P = self.bwd, self.dt, self.B, self.P
        Hin, Cin, Cout, cname, nname = st['Hin'], st['Cin'], st['Cout'], st['cname'], st['nname']
        Ho = Hin // 2
        Mo, Mi = B * Ho * Ho, B * Hin * Hin
        dc = self.tmp('dconv', (Mo, Cout))
        Bk.layernorm_bwd(dy, st['c'], st['mean'], st['rstd'], P[nname + 'weight'], None, dc, self.grad(nname + 'weight'),
                         self.grad(nname + 'bias'), Mo, Cout, False, dt, label=nname + 'lnb')
        G = self.gbuf((Cout, 9 * Cin))
        with self._wlane():
            Bk.wgrad(dc, st['x'], G, Mo, Cout, 9 * Cin, dt, x_kind=A_CONV3S2, x_dims=(Hin, Hin, Cin),
                     dbias=self.grad(cname + 'bias') if st['bias'] else None, label=cname + 'wg')
        Bk.weight_unfold(G, 9 * Cin, Cout, Cin, 3, 3, dW=self.grad(cname + 'weight'), label=cname + 'unf')
        if dprev is not None:
            Bk.gemm(dc, st['Bt'], dprev, Mo, 4 * Cin, 4 * Cout, dt, ldb=pad8(4 * Cout), a_kind=A_NEIGH2, a_dims=(Ho, Ho, Cout),
                    c_kind=C_UNPATCH2, c_dims=(Hin, Hin, Cin), label=cname + 'dg')
            if seed is not None:
                Bk.affine_act(dprev, None, None, seed, dprev, Mi, Cin, False, dt, label=cname + 'seed')

    # ------------------------------------------------------------------------------------------
    # build
    # ------------------------------------------------------------------------------------------
    def _build(self):
        cfg, m = self.cfg, self.m
        B, T, F, dt, P = self.B, self.training, self.fwd, self.dt, self.P
        e, d, dep = cfg['embed_dim'], cfg['dims'], cfg['depth']
        img = self.img
        if T:
            F.zero(self.bn_pool, label='zero.bn_sums')
        # ---------------- deep stem (ga_cswin.py:463-477) ----------------
        sp = 'stage1_conv_embed.'
        H1 = img // 2
        M1 = B * H1 * H1
        self.x8 = self.buf('stem.x8', (B * img * img, 8))
        self.x_placeholder = torch.zeros(8, device=self.dev)
        F.nchw3_to_nhwc8(self.x_placeholder, self.x8, B, img, img, dt, label='stem.pack')
        self.pack_call = len(F.calls) - 1
        W0 = self.buf('w.' + sp + '0', (e, 72))
        self.prep.convw_pack(P[sp + '0.weight'], W0, e, 3, 9, 8, 72, dt, label='prep.' + sp + '0')
        W1 = self._w_plain(sp + '5.weight', e, e, 3, 3, flip=True)
        S = self.stem = {}
        S['c0'] = self.act(sp + 'c0', (M1, e))
        S['a0'] = self.act(sp + 'a0', (M1, e))
        S['m0'], S['r0'] = self.act(sp + 'm0', (M1,), torch.float32), self.act(sp + 'r0', (M1,), torch.float32)
        S['c1'] = self.act(sp + 'c1', (M1, e))
        S['a1'] = self.act(sp + 'a1', (M1, e))
        S['m1'], S['r1'] = self.act(sp + 'm1', (M1,), torch.float32), self.act(sp + 'r1', (M1,), torch.float32)
        stages = [m.stage1, m.stage2, m.stage3, m.stage4]
        taps_at = tap_after(dep[2], cfg['naggre'])
        # one pass per forward chain (batch part, GAEngine._chains): with GAEXT_FWD_SPLIT > 1 the deep stem and the four stages run as
        # independent chains on side streams -- every op of the trunk is per image, and its launches are too small to fill the chip
        # alone (the step timeline had one kernel running for 23 of 39 ms); every pass names the same full-batch buffers
        chains = self._chains()
        if any(mod.mlp_groups != 1 for st_ in stages for mod in st_):
            chains = [(0, 0, B)]
        for chain in chains:
            self._chain = chain if len(chains) > 1 else None
            (lane, r0, r1, b0, b1), = self._fsplits(H1 * H1)
            F.lane = lane
            F.gemm(self.x8[b0 * img * img:b1 * img * img], W0, S['c0'][r0:r1], r1 - r0, e, 72, dt, a_kind=A_CONV3S2, a_dims=(img, img, 8),
                   label=sp + 'conv0')
            F.layernorm_gelu_fwd(S['c0'][r0:r1], P[sp + '2.weight'], P[sp + '2.bias'], S['a0'][r0:r1], S['m0'][r0:r1], S['r0'][r0:r1],
                                 r1 - r0, e, 1e-5, dt, label=sp + 'ln0')
            F.gemm(S['a0'][r0:r1], W1, S['c1'][r0:r1], r1 - r0, e, 9 * e, dt, ldb=pad8(9 * e), a_kind=A_CONV3, a_dims=(H1, H1, e),
                   label=sp + 'conv1')
            F.layernorm_gelu_fwd(S['c1'][r0:r1], P[sp + '7.weight'], P[sp + '7.bias'], S['a1'][r0:r1], S['m1'][r0:r1], S['r1'][r0:r1],
                                 r1 - r0, e, 1e-5, dt, label=sp + 'ln1')
            x, S['conv2'] = self._conv3s2_ln_fwd(sp + '10.', sp + '12.', S['a1'], H1, e, d[0], bias=False)
            # ---------------- stages 1..4 ----------------
            feats, taps, self.merges = [], [], {}
            for si in range(4):
                if si > 0:
                    x, self.merges[si] = self._conv3s2_ln_fwd(f'merge{si}.conv.', f'merge{si}.norm.', x, stages[si - 1][0].reso,
                                                              d[si - 1], d[si], bias=True)
                for j, mod in enumerate(stages[si]):
                    x = self._cs_block_fwd(f'stage{si + 1}.{j}.', x, mod)
                    if si == 2 and (j + 1) in taps_at:
                        taps.append((x, j))
                feats.append((x, stages[si][0].reso))
        self._chain = None
        F.lane = 0
        # ---------------- aggregate (ga_cswin.py:666-669) ----------------
        Hc = img // 16
        M4 = B * Hc * Hc
        ctot = sum(d) + d[2] * cfg['naggre']
        assert len(taps) == cfg['naggre'], (len(taps), cfg['naggre'])
        cat = self.act('agg.cat', (M4, ctot))
        segs = [(feats[0][0], feats[0][1], d[0], 0), (feats[1][0], feats[1][1], d[1], 0)]
        segs += [(t, feats[2][1], d[2], 0) for t, _ in taps]
        segs += [(feats[2][0], feats[2][1], d[2], 0), (feats[3][0], feats[3][1], d[3], 1)]
        off = 0
        self.agg_segs = []
        for src, hw, c, mode in segs:
            F.pool_concat_fwd(src, cat, B, hw, hw, c, Hc, Hc, ctot, off, mode, dt, label=f'agg.{off}')
            self.agg_segs.append((src, hw, c, mode, off))
            off += c
        assert off == ctot
        # ---------------- stage5 ----------------
        cur = d[3]
        self.cout = cur
        if cfg['stage5'] == 'CSWin':
            lp = 'stage5.1.'
            W5 = self._w_plain(lp + 'conv.weight', cur, ctot, 1, 1)
            L = self.lcf = dict(c=self.act(lp + 'c', (M4, cur)), mean=self.act(lp + 'mean', (M4,), torch.float32),
                                rstd=self.act(lp + 'rstd', (M4,), torch.float32))
            F.gemm(cat, W5, L['c'], M4, cur, ctot, dt, ldb=pad8(ctot), bias=P[lp + 'conv.bias'], label=lp + 'conv')
            t5 = self.buf(lp + 'out', (M4, cur))
            F.layernorm_fwd(L['c'], P[lp + 'norm.weight'], P[lp + 'norm.bias'], t5, L['mean'], L['rstd'], M4, cur, 1e-5, dt,
                            label=lp + 'ln')
            L['t5'] = t5
            x4 = self._cs_block_fwd('stage5.2.', t5, m.stage5[2])
        else:
            self.bott_prefix = 'stage5.'
            x4 = self._bottleneck_fwd(cat, M4, ctot, cur)
        # ---------------- heads ----------------
        self._build_heads(x4, M4, Hc)
        # ---------------- backward ----------------
        if T:
            self._build_backward(feats, taps, x4, M4, ctot, cat)
            if self.async_wgrad:
                self.bwd.join_async()
            self.bwd.flush('end.')
        self.prep.flush('prep.')

    # ------------------------------------------------------------------------------------------
    # grouped gram_contraction of all heads (ga_cswin.py:559-561): ONE batched GEMM, batch = heads x 8 groups
    # ------------------------------------------------------------------------------------------
    def _contract_all_fwd(self, x4, M4):
        cfg, T, F, dt, cout, P = self.cfg, self.training, self.fwd, self.dt, self.cout, self.P
        K, g_, gg = cfg['branches'], cfg['gram_dim'], cfg['gram_groups']
        gpo, cpi = g_ // gg, cout // gg
        assert g_ % gg == 0 and gpo % 8 == 0 and cpi % 8 == 0, 'grouped gram_contraction needs 8-aligned group widths'
        gc = self.gcon = dict(ld=K * g_, gpo=gpo, cpi=cpi, dense=False)
        if dt == ops.GA_BF16 and os.environ.get('GAEXT_GC_DENSE', '1') != '0':
            # the five grouped convs as ONE dense product over block-diagonal weights ([K g_] x cout, zeros off the diagonal blocks):
            # 8 x the MACs of the grouped form, but one ring-form launch with N = 960 instead of 40 products with N = 24 whose 48-byte
            # output rows are partial cache lines (0.196 -> 0.07 ms forward; the data gradient of all heads one launch as well)
            gc['dense'] = True
            gc['Wm'] = self.buf('pad.gram_contraction.bd', (K * g_, cout), torch.float32, zero=True)     # block-diagonal fp32 master
            for k in range(K):
                self.prep.blockdiag_f32(P[f'gram_contraction.{k}.0.weight'], gc['Wm'][k * g_:], g_, gpo, gg, cpi, cout,
                                        label=f'prep.gram_contraction.{k}.bd')
            gc['Wd'] = self.buf('w.gram_contraction.bd', (K * g_, cout))
            gc['WdT'] = self.buf('wT.gram_contraction.bd', (cout, pad8(K * g_))) if T else None
            self.prep.weight_prep(gc['Wm'], 1, K * g_, cout, 1, 1, dt, out=gc['Wd'], ldo=cout, outT=gc['WdT'],
                                  ldt=pad8(K * g_) if T else 0, label='prep.gram_contraction.bd')
            gc['b'] = self.buf('w.gram_contraction.ball', (K * g_,), torch.float32)
            gc['s'], gc['q'] = self._bn_pool(K * g_), self._bn_pool(K * g_)
            for k in range(K):
                self.prep.bias_fold(None, P[f'gram_contraction.{k}.0.bias'], None, None, gc['b'][k * g_:], g_, cpi)
            gc['out'] = self.act('gram_contraction.all.out', (M4, K * g_))
            F.gemm(x4, gc['Wd'], gc['out'], M4, K * g_, cout, dt, bias=gc['b'], colsum=gc['s'] if T else None,
                   colsumsq=gc['q'] if T else None, label='gram_contraction.all')
            return
        gc['W'] = self.buf('w.gram_contraction.all', (K * g_, cpi))
        gc['WT'] = self.buf('wT.gram_contraction.all', (K, gg * cpi, pad8(gpo))) if T else None
        gc['b'] = self.buf('w.gram_contraction.ball', (K * g_,), torch.float32)
        gc['s'], gc['q'] = self._bn_pool(K * g_), self._bn_pool(K * g_)
        for k in range(K):
            pre = f'gram_contraction.{k}.'
            self.prep.weight_prep(P[pre + '0.weight'], gg, gpo, cpi, 1, 1, dt, out=gc['W'][k * g_:], ldo=cpi,
                                  outT=gc['WT'][k] if T else None, ldt=pad8(gpo) if T else 0, label='prep.' + pre + 'w')
            self.prep.bias_fold(None, P[pre + '0.bias'], None, None, gc['b'][k * g_:], g_, cpi)
        gc['out'] = self.act('gram_contraction.all.out', (M4, K * g_))
        F.gemm(x4, gc['W'], gc['out'], M4, gpo, cpi, dt, lda=cout, ldb=cpi, ldc=K * g_, batch=K * gg, a_batch_mod=gg, strideA=cpi,
               strideB=gpo * cpi, strideC=gpo, bias=gc['b'], strideBias=gpo, colsum=gc['s'] if T else None,
               colsumsq=gc['q'] if T else None, strideCol=gpo, label='gram_contraction.all')

    def _contract_all_bwd(self, x4, dx4, M4):
        Bk, dt, cfg, cout = self.bwd, self.dt, self.cfg, self.cout
        K, g_, gg = cfg['branches'], cfg['gram_dim'], cfg['gram_groups']
        gcn = self.gcon
        gpo, cpi = gcn['gpo'], gcn['cpi']
        Gc, gbc = self.gbuf((K * g_, cpi)), self.gbuf((K * g_,))
        if gcn['dense']:
            Gd = self.gbuf((K * g_, cout))
            with self._wlane():
                Bk.wgrad(gcn['dout'], x4, Gd, M4, K * g_, cout, dt, dbias=gbc, label='gram_contraction.all.wg')
                Bk.blockdiag_f32(Gd, Gc, K * g_, gpo, gg, cpi, cout, to_diag=False, label='gram_contraction.all.wg.diag')
            for k in range(K):
                pre = f'gram_contraction.{k}.'
                Bk.axpy_f32(self.grad(pre + '0.weight'), Gc[k * g_:], 1.0, g_ * cpi)
                Bk.axpy_f32(self.grad(pre + '0.bias'), gbc[k * g_:], 1.0, g_)
            Bk.gemm(gcn['dout'], gcn['WdT'], dx4, M4, cout, K * g_, dt, ldb=pad8(K * g_), R=None if self.shared_tok else dx4, ldr=cout,
                    label='gram_contraction.all.dg')
            return
        with self._wlane():
            Bk.wgrad(gcn['dout'], x4, Gc, M4, gpo, cpi, dt, ldy=K * g_, ldx=cout, ldw=cpi, batch=K * gg, strideY=gpo, strideX=cpi,
                     x_batch_mod=gg, strideW=gpo * cpi, dbias=gbc, strideDbias=gpo, label='gram_contraction.all.wg')
        for k in range(K):
            pre = f'gram_contraction.{k}.'
            Bk.axpy_f32(self.grad(pre + '0.weight'), Gc[k * g_:], 1.0, g_ * cpi)
            Bk.axpy_f32(self.grad(pre + '0.bias'), gbc[k * g_:], 1.0, g_)
        for k in range(K):   # grouped data gradient, one batched GEMM per head; the first one is the first writer of dx4
            first = k == 0 and self.shared_tok
            Bk.gemm(gcn['dout'][:, k * g_:], gcn['WT'][k], dx4, M4, cpi, gpo, dt, lda=K * g_, ldb=pad8(gpo), ldc=cout, batch=gg,
                    strideA=gpo, strideB=cpi * pad8(gpo), strideC=cpi, R=None if first else dx4, ldr=cout, strideR=cpi,
                    label=f'gram_contraction.{k}.dg')

    # gram_layer[k] = CSWinBlock(gram_dim, 6 heads) at 14 x 14 (ga_cswin.py:564-574)
    def _gram_layer_fwd(self, h, k, Hc):
        h['blk'] = f'gram_layer.{k}.1.'
        return self._cs_block_fwd(h['blk'], h['g0'], self.m.gram_layer[k][1])

    def _gram_layer_bwd(self, h, dg1, dg0):
        self._cs_block_bwd(h['blk'], dg1, dg0)

    # ------------------------------------------------------------------------------------------
    # whole-network backward plan
    # ------------------------------------------------------------------------------------------
    def _build_backward(self, feats, taps, x4, M4, ctot, cat):
        Bk, dt, B, P, W, cfg, m = self.bwd, self.dt, self.B, self.P, self.W, self.cfg, self.m
        d, dep, e = cfg['dims'], cfg['depth'], cfg['embed_dim']
        cur = self.cout
        dx4 = self._build_heads_backward(x4, M4)
        dcat = self.tmp('dcat', (M4, ctot))
        if cfg['stage5'] == 'CSWin':
            L = self.lcf
            lp = 'stage5.1.'
            dt5 = self.tmp('dt5', (M4, cur))
            self._cs_block_bwd('stage5.2.', dx4, dt5)
            dc5 = self.tmp('dc5', (M4, cur))
            Bk.layernorm_bwd(dt5, L['c'], L['mean'], L['rstd'], P[lp + 'norm.weight'], None, dc5, self.grad(lp + 'norm.weight'),
                             self.grad(lp + 'norm.bias'), M4, cur, False, dt, label=lp + 'lnb')
            with self._wlane():
                Bk.wgrad(dc5, cat, self.grad(lp + 'conv.weight'), M4, cur, ctot, dt, dbias=self.grad(lp + 'conv.bias'),
                         label=lp + 'wg')
            Bk.gemm(dc5, W[lp + 'conv.weight.T'], dcat, M4, ctot, cur, dt, ldb=pad8(cur), label=lp + 'dg')
        else:
            self._bottleneck_bwd(dx4, dcat)
        if self.async_wgrad:
            Bk.join_async()
        Bk.flush('heads.')
        self._unpad_all()
        Bk.mark('heads')      # every gradient of stage5 / gram_* / ga / fc is final here
        # aggregate backward -> gradient seeds of the stage outputs / taps
        seeds = []
        for src, hw, c, mode, off in self.agg_segs:
            ds = self.buf(f'agg.d{off}', (B * hw * hw, c))
            Bk.pool_concat_bwd(dcat, None, ds, B, hw, hw, c, 14, 14, ctot, off, mode, dt, label=f'agg.b{off}')
            seeds.append(ds)
        ntap = len(taps)
        d_taps = {j: seeds[2 + i] for i, (_, j) in enumerate(taps)}
        seed = {0: seeds[0], 1: seeds[1], 2: seeds[2 + ntap], 3: seeds[3 + ntap]}
        stages = [m.stage1, m.stage2, m.stage3, m.stage4]
        dy = seed[3]
        for si in (3, 2, 1, 0):
            reso = stages[si][0].reso
            Mi = B * reso * reso
            pp = [self.tmp(f'dxA{si}', (Mi, d[si])), self.tmp(f'dxB{si}', (Mi, d[si])), self.tmp(f'dxC{si}', (Mi, d[si]))]
            turn = 0
            for j in reversed(range(dep[si])):
                if si == 2 and j in d_taps:
                    Bk.affine_act(dy, None, None, d_taps[j], dy, Mi, d[si], False, dt, label=f'tap.add.{j}')
                dx = pp[turn % 3]          # not this block's dy nor the previous block's (still read by its asynchronous wgrads)
                turn += 1
                # the next block of the chain takes dx unchanged as its dy unless a tap gradient is added in between
                nxt = f'stage{si + 1}.{j - 1}.' if j > 0 and not (si == 2 and (j - 1) in d_taps) else None
                self._cs_block_bwd(f'stage{si + 1}.{j}.', dy, dx, next_pre=nxt)
                dy = dx
            if si > 0:
                Hp = stages[si - 1][0].reso
                dprev = self.tmp(f'dprev{si}', (B * Hp * Hp, d[si - 1]))
                self._conv3s2_ln_bwd(self.merges[si], dy, dprev, seed=seed[si - 1])
                dy = dprev
            if self.async_wgrad:
                Bk.join_async()
            Bk.flush(f'stage{si + 1}.')
            Bk.mark(f'stage{si}')   # gradients of trunk stage si+1 (incl. its merge) are final
        # ---------------- deep stem ----------------
        S = self.stem
        sp = 'stage1_conv_embed.'
        H1 = self.img // 2
        M1 = B * H1 * H1
        da1 = self.tmp('stem.da1', (M1, e))
        self._conv3s2_ln_bwd(S['conv2'], dy, da1)
        dc1 = self.tmp('stem.dc', (M1, e))
        Bk.layernorm_gelu_bwd(da1, S['c1'], S['m1'], S['r1'], P[sp + '7.weight'], P[sp + '7.bias'], dc1, self.grad(sp + '7.weight'),
                              self.grad(sp + '7.bias'), M1, e, dt, label=sp + 'ln1b')
        G1 = self.gbuf((e, 9 * e))
        with self._wlane():
            Bk.wgrad(dc1, S['a0'], G1, M1, e, 9 * e, dt, x_kind=A_CONV3, x_dims=(H1, H1, e), label=sp + 'conv1.wg')
        Bk.weight_unfold(G1, 9 * e, e, e, 3, 3, dW=self.grad(sp + '5.weight'), label=sp + 'conv1.unf')
        da0 = da1   # da1 is dead after the LayerNorm+GELU backward above
        Bk.gemm(dc1, W[sp + '5.weight.T'], da0, M1, e, 9 * e, dt, a_kind=A_CONV3, a_dims=(H1, H1, e), ldb=pad8(9 * e),
                label=sp + 'conv1.dg')
        dc0 = self.tmp('stem.dc0', (M1, e))   # dc1 is still read by the asynchronous conv1 weight gradient
        Bk.layernorm_gelu_bwd(da0, S['c0'], S['m0'], S['r0'], P[sp + '2.weight'], P[sp + '2.bias'], dc0, self.grad(sp + '2.weight'),
                              self.grad(sp + '2.bias'), M1, e, dt, label=sp + 'ln0b')
        G0 = self.gbuf((e, 72))
        with self._wlane():
            Bk.wgrad(dc0, self.x8, G0, M1, e, 72, dt, x_kind=A_CONV3S2, x_dims=(self.img, self.img, 8), label=sp + 'conv0.wg')
            Bk.convw_unpack_grad(G0, self.grad(sp + '0.weight'), e, 3, 9, 8, 72, label=sp + 'conv0.unf')   # same lane: after the wgrad

    # ------------------------------------------------------------------------------------------
    def set_input(self, x):
        x = self._normalize_u8(x)
        assert x.is_cuda and x.dtype == torch.float32 and tuple(x.shape) == (self.B, 3, self.img, self.img), \
            f'input must be a float32 CUDA tensor of shape {(self.B, 3, self.img, self.img)}, got {tuple(x.shape)} {x.dtype}'
        if not x.is_contiguous():
            x = x.contiguous()
        self.x_ref = x
        fn, args, label = self.fwd.calls[self.pack_call]
        self.fwd.calls[self.pack_call] = (fn, (x.data_ptr(),) + tuple(args[1:]), label)
