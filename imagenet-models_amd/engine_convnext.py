"""ConvNeXtEngine: the map_convnext trunk (engine.GAEngine._build_trunk through MAPEngine's parameter names) with the plain head of
/root/reference/MAP/models/map_convnext.py:111-115,134-140 -- global average pool (ga_spatial_sum), LayerNorm(1e-6), Linear
(ga_gemm, fp32 logits); backward: classifier weight gradient / dgrad, LayerNorm backward, ga_rows_bcast as the seed of stage 3."""
import torch

from . import ops  # noqa: F401
from .engine import GAEngine, pad8
from .engine_map import MAPEngine


class ConvNeXtEngine(MAPEngine):
    def _build(self):
        cfg = self.cfg
        d = cfg['dims']
        B, T, F, dt, P = self.B, self.training, self.fwd, self.dt, self.P
        NC = cfg['num_classes']
        assert NC % 8 == 0, 'num_classes must be a multiple of 8 (pad the classifier)'
        self.drop = None
        self.G = 1
        feats, taps, stage_in, x_stem = self._build_trunk()
        x3, res = feats[3]
        HW, C = res * res, d[3]
        hd = self.hd = dict(pool=self.act('head.pool', (B, C)), y=self.act('head.y', (B, C)), mean=self.act('head.mean', (B,), torch.float32),
                            rstd=self.act('head.rstd', (B,), torch.float32))
        pool32 = self.tmp('head.pool32', (B, C), torch.float32)            # ga_spatial_sum reduces into fp32
        F.spatial_sum(x3, None, pool32, B, HW, C, 1.0 / HW, dt, label='head.pool')
        F.cast_from_f32(pool32, hd['pool'], B * C, dt, label='head.pool.cast')
        F.layernorm_fwd(hd['pool'], P['norm.weight'], P['norm.bias'], hd['y'], hd['mean'], hd['rstd'], B, C, 1e-6, dt, label='head.ln')
        Wh = self._w_plain('head.weight', NC, C, 1, 1)
        self.logits = self.buf('logits', (1, B, NC), torch.float32)
        F.gemm(hd['y'], Wh, self.logits[0], B, NC, C, dt, bias=P['head.bias'], c_f32=True, label='head.fc')
        if T:
            Bk = self.bwd
            self.dlogits = self.buf('dlogits', (1, B, NC))
            dl = self.dlogits[0]
            with self._wlane():
                Bk.wgrad(dl, hd['y'], self.grad('head.weight'), B, NC, C, dt, dbias=self.grad('head.bias'), label='head.wg')
            dy = self.tmp('head.dy', (B, C))
            Bk.gemm(dl, self.W['head.weight.T'], dy, B, C, NC, dt, ldb=pad8(NC), label='head.dg')
            dpool = self.tmp('head.dpool', (B, C))
            Bk.layernorm_bwd(dy, hd['pool'], hd['mean'], hd['rstd'], P['norm.weight'], None, dpool, self.grad('norm.weight'),
                             self.grad('norm.bias'), B, C, False, dt, label='head.lnb')
            seed3 = self.buf('head.seed3', (B * HW, C))
            Bk.rows_bcast(dpool, seed3, B, HW, C, 1.0 / HW, dt, label='head.poolb')
            Bk.mark('heads')
            self._build_trunk_backward({3: seed3, 2: None, 1: None, 0: None}, [], [], feats, stage_in)
            if self.async_wgrad:
                Bk.join_async()
            Bk.flush('end.')
        self.prep.flush('prep.')

    def _loss_operands(self):
        return GAEngine._loss_operands(self)
