"""Launch plans over the libgaext C ABI.

A `Plan` is a pre-resolved list of C-ABI calls (ctypes function + fully built argument tuple) over persistent
device buffers.  The engine builds the forward / backward / optimizer plans ONCE per (batch size, mode) and then
replays them on a HIP stream every step: no per-step descriptor construction, no allocation, no host sync --
the MI355X-native replacement for the reference's eager ATen dispatch (and directly capturable in a hipGraph).
Every method records one call; `run()` enqueues them all.  `Plan(eager=True)` also executes each call as it is
recorded (used by the unit tests).
"""
import ctypes as C
import os

import torch

from . import _lib as L
from ._lib import (A_CONV3, A_CONV3S2, A_NEIGH2, A_PATCH2, A_PLAIN, A_STEM4_NCHW, ACT_GELU, ACT_NONE, ACT_RELU, C_PLAIN, C_UNPATCH2,  # noqa: F401
                   GA_BF16, GA_F32)


def ga_dtype(t):
    if isinstance(t, torch.dtype):
        dt = t
    else:
        dt = t.dtype
    if dt == torch.bfloat16:
        return GA_BF16
    if dt == torch.float32:
        return GA_F32
    raise TypeError(f'unsupported dtype {dt}')


def torch_dtype(ga):
    return torch.bfloat16 if ga == GA_BF16 else torch.float32


def _ptr(t):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'libgaext operand must be a torch.Tensor or None, got {type(t).__name__}: {t!r}')
    assert t.is_cuda, 'libgaext operands must live in device memory (no CPU fallback)'
    return t.data_ptr()


def current_stream_ptr():
    return torch.cuda.current_stream().cuda_stream


_SIDE_POOL = {}    # device index -> (side streams, their "done" events, fork event): shared by every plan of the process


def _side_pool(n):
    dev = torch.cuda.current_device()
    streams, done, fork = _SIDE_POOL.get(dev, ([], [], None))
    while len(streams) < n:
        streams.append(torch.cuda.Stream())
        done.append(torch.cuda.Event())
    fork = fork or torch.cuda.Event()
    _SIDE_POOL[dev] = (streams, done, fork)
    return streams, done, fork


_ASYNC_POOL = {}   # device index -> (stream, done event, fork event) of the asynchronous lane


def _async_pool():
    dev = torch.cuda.current_device()
    if dev not in _ASYNC_POOL:
        _ASYNC_POOL[dev] = (torch.cuda.Stream(), torch.cuda.Event(), torch.cuda.Event())
    return _ASYNC_POOL[dev]


def _join_marker(*_):
    """pseudo call recorded by Plan.join_async(); a no-op wherever a plan's calls are replayed one by one"""
    return 0


def _amark_marker(*_):
    """pseudo call recorded by Plan.async_mark(): a named point of the asynchronous lane that join_async(tag) waits for"""
    return 0


def _lsig_marker(*_):
    """pseudo call of Plan.lane_signal(): an event recorded on a side lane's stream"""
    return 0


def _lwait_marker(*_):
    """pseudo call of Plan.lane_wait(): a side lane waits for an event of another lane"""
    return 0


ASYNC_LANE = -1


class Plan:
    def __init__(self, eager=False, name='', defer_small=False):
        self.lib = L.load()
        self.calls = []      # (cfunc, args tuple, label)
        self.keep = []       # tensors / descriptors kept alive
        self.eager = eager
        self.name = name
        self.marks = {}
        # defer_small: weight-prep / un-fold / bias-fold / transpose / axpy jobs are collected and emitted by flush()
        # as ONE batched launch per kind over a device-resident descriptor array
        self.defer = defer_small and not eager
        self._pend = {'wprep': [], 'small': [], 'unfold': []}
        # lanes: calls recorded while lane > 0 form a PARALLEL REGION -- independent chains (the five GA heads) that
        # run() puts on side streams between a fork (side streams wait for the main stream) and a join (the main stream
        # waits for them).  Lane 0 is the stream run() is called on.
        # Lane -1 (ASYNC_LANE) is different: such a call runs on ONE extra stream after everything recorded before it on
        # lane 0, and lane 0 does not wait for it until join_async() (or the end of the range being run).  The trunk's
        # weight-gradient launches go there: nothing on the dgrad chain reads their results.
        # workspaces: ga_wgrad / ga_dwconv7_bwd_weight reduce per-workgroup partial results through CALLER-owned scratch
        # (include/gaext.h): one torch buffer per lane (launches of one lane are stream-ordered and may share it; lanes run
        # concurrently), sized to the largest request of that lane and allocated before the first run
        self._ws_req = []    # (lane, bytes, patch callable(ptr, bytes))
        self._ws = {}        # lane -> uint8 tensor
        self._ws_dirty = False
        self.lane = 0
        self.lanes = []      # lane of every call
        self._side = None    # number of side lanes (streams come from a process-wide pool)
        self._aevents = {}   # async_mark tag -> event

    def flush(self, label=''):
        for kind, fname, cls in (('wprep', 'ga_weight_prep_batch', L.WprepDesc), ('small', 'ga_small_batch', L.SmallDesc),
                                 ('unfold', 'ga_weight_unfold_batch', L.WunfoldDesc)):
            lst = self._pend[kind]
            if not lst:
                continue
            arr = (cls * len(lst))(*lst)
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
            self._add(fname, (dev.data_ptr(), len(lst)), f'{label}{kind}.batch[{len(lst)}]', keep=(dev, arr))
            self._pend[kind] = []

    def _small(self, **kw):
        d = L.SmallDesc()
        for k, v in kw.items():
            setattr(d, k, v)
        self._pend['small'].append(d)
        if os.environ.get('GAEXT_SMALL_SINGLE'):      # diagnosis: one launch per small job, labelled with its geometry
            self.flush(f'single.k{d.kind}.R{d.R}.C{d.C}.n{d.n}.')

    # -- core ---------------------------------------------------------------------------------------
    def _add(self, fname, args, label=None, keep=()):
        fn = getattr(self.lib, fname)
        self.calls.append((fn, args, label or fname))
        self.lanes.append(self.lane)
        self.keep.extend(keep)
        if self.eager:
            L.check(fn(*args, current_stream_ptr()), label or fname)

    def _want_workspace(self, nbytes, patch):
        if self.eager:
            t = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
            self.keep.append(t)
            patch(t.data_ptr(), nbytes)
            return
        self._ws_req.append((self.lane, nbytes, patch))
        self._ws_dirty = True

    def finalize(self):
        """allocate / grow the per-lane workspaces and hand their addresses to the recorded calls"""
        if not self._ws_dirty:
            return
        need = {}
        for lane, nbytes, _ in self._ws_req:
            need[lane] = max(need.get(lane, 0), nbytes)
        for lane, nbytes in need.items():
            if lane not in self._ws or self._ws[lane].numel() < nbytes:
                self._ws[lane] = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
        for lane, nbytes, patch in self._ws_req:
            patch(self._ws[lane].data_ptr(), self._ws[lane].numel())
        self._ws_dirty = False

    def run(self, stream=None):
        self.finalize()
        s = current_stream_ptr() if stream is None else stream
        if os.environ.get('GAEXT_SYNC_DEBUG'):   # localise a faulting launch: label printed before, sync after
            for fn, args, label in self.calls:
                print(f'[gaext] {self.name}:{label}', flush=True)
                L.check(fn(*args, s), f'{self.name}:{label}')
                torch.cuda.synchronize()
            return
        if any(self.lanes):
            return self._run_lanes(0, len(self.calls))
        for fn, args, label in self.calls:
            rc = fn(*args, s)
            if rc != 0:
                L.check(rc, f'{self.name}:{label}')

    def join_async(self, tag=None):
        """lane 0 waits here for everything recorded on the asynchronous lane so far (tag: up to async_mark(tag) only)"""
        self.calls.append((_join_marker, () if tag is None else (tag,), 'join_async'))
        self.lanes.append(0)

    def lane_signal(self, lane):
        """a point of side lane `lane` (inside a parallel region) that lane_wait() of another lane can wait for"""
        ev = torch.cuda.Event()
        self.calls.append((_lsig_marker, (lane, ev), 'lane_signal'))
        self.lanes.append(lane)
        return ev

    def lane_wait(self, lane, ev):
        self.calls.append((_lwait_marker, (lane, ev), 'lane_wait'))
        self.lanes.append(lane)

    def async_mark(self, tag):
        self.calls.append((_amark_marker, (tag,), 'async_mark'))
        self.lanes.append(0)

    def _run_lanes(self, start, end):
        """calls[start:end] with the parallel regions on side streams (the current torch stream is lane 0)"""
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = max(max(self.lanes), 0)
        streams, done, fork = _side_pool(self._side)
        astream, adone, afork = _async_pool()
        apend = False
        aev = {}                                  # named points recorded in THIS run (a range may start after a mark)
        lsig = set()
        s0 = main.cuda_stream
        used = []

        def ajoin():
            adone.record(astream)
            main.wait_event(adone)

        for i in range(start, end):
            fn, args, label = self.calls[i]
            lane = self.lanes[i]
            if fn is _join_marker:
                if args:                          # up to a named point of the asynchronous lane (recorded in this run)
                    ev = aev.get(args[0])
                    if ev is not None:
                        main.wait_event(ev)
                elif apend:
                    ajoin()
                    apend = False
                continue
            if fn is _amark_marker:
                if apend:
                    ev = self._aevents.get(args[0])
                    if ev is None:
                        ev = self._aevents[args[0]] = torch.cuda.Event()
                    ev.record(astream)
                    aev[args[0]] = ev
                continue
            if fn is _lsig_marker:
                if args[0] in used:
                    args[1].record(streams[args[0] - 1])
                    lsig.add(args[1])
                continue
            if fn is _lwait_marker:
                if args[1] in lsig:               # recorded in this run
                    if not used:
                        fork.record(main)
                    if args[0] not in used:
                        streams[args[0] - 1].wait_event(fork)
                        used.append(args[0])
                    streams[args[0] - 1].wait_event(args[1])
                continue
            if lane == ASYNC_LANE:
                if used:                          # the asynchronous lane starts from lane 0's point: close the region first
                    for ln in used:
                        done[ln - 1].record(streams[ln - 1])
                        main.wait_event(done[ln - 1])
                    used = []
                afork.record(main)
                astream.wait_event(afork)
                apend = True
                rc = fn(*args, astream.cuda_stream)
            elif lane == 0:
                if used:                          # join: everything after this point sees the side streams' work
                    for ln in used:
                        done[ln - 1].record(streams[ln - 1])
                        main.wait_event(done[ln - 1])
                    used = []
                rc = fn(*args, s0)
            else:
                if not used:                      # fork: the side streams start from the main stream's current point
                    if apend:
                        ajoin()
                        apend = False
                    fork.record(main)
                if lane not in used:
                    streams[lane - 1].wait_event(fork)
                    used.append(lane)
                rc = fn(*args, streams[lane - 1].cuda_stream)
            if rc != 0:
                L.check(rc, f'{self.name}:{label}')
        for ln in used:                           # a region that runs to the end of the range
            done[ln - 1].record(streams[ln - 1])
            main.wait_event(done[ln - 1])
        if apend:
            ajoin()

    def __len__(self):
        return len(self.calls)

    def mark(self, name):
        """remember the current position (used to run a plan in segments, e.g. to start a gradient all-reduce early)"""
        self.marks[name] = len(self.calls)

    def run_range(self, start, end, stream=None):
        self.finalize()
        s = current_stream_ptr() if stream is None else stream
        if any(self.lanes[start:end]):
            return self._run_lanes(start, end)
        for fn, args, label in self.calls[start:end]:
            rc = fn(*args, s)
            if rc != 0:
                L.check(rc, f'{self.name}:{label}')

    # -- GEMM family --------------------------------------------------------------------------------
    def gemm(self, A, B, Cout, M, N, K, dtype, lda=None, ldb=None, ldc=None, batch=1, strideA=0, strideB=0, strideC=0,
             a_batch_mod=0, a_kind=A_PLAIN, a_dims=(0, 0, 0), a_act=ACT_NONE, c_kind=C_PLAIN, c_dims=(0, 0, 0),
             c_f32=False, alpha=1.0, bias=None, strideBias=0, act=ACT_NONE, H=None, ldh=0, strideH=0, rowscale=None,
             rows_per_scale=1, R=None, ldr=0, strideR=0, relu_after=False, colsum=None, colsumsq=None, strideCol=0,
             C2=None, c2_mode=0, h_is_deriv=False, label=None):
        d = L.GemmDesc()
        d.M, d.N, d.K, d.batch, d.dtype = M, N, K, batch, dtype
        d.A, d.lda, d.strideA, d.a_batch_mod, d.a_kind = _ptr(A), (K if lda is None else lda), strideA, a_batch_mod, a_kind
        d.a_H, d.a_W, d.a_C = a_dims
        d.a_act = a_act
        d.B, d.ldb, d.strideB = _ptr(B), (K if ldb is None else ldb), strideB
        d.C, d.ldc, d.strideC, d.c_kind = _ptr(Cout), (N if ldc is None else ldc), strideC, c_kind
        d.c_H, d.c_W, d.c_C = c_dims
        d.c_f32 = int(c_f32)
        d.C2, d.c2_mode = _ptr(C2), c2_mode
        d.h_is_deriv = int(h_is_deriv)
        d.alpha = alpha
        d.bias, d.strideBias, d.act = _ptr(bias), strideBias, act
        d.H, d.ldh, d.strideH = _ptr(H), ldh, strideH
        d.rowscale, d.rows_per_scale = _ptr(rowscale), rows_per_scale
        d.R, d.ldr, d.strideR, d.relu_after = _ptr(R), ldr, strideR, int(relu_after)
        d.colsum, d.colsumsq, d.strideCol = _ptr(colsum), _ptr(colsumsq), strideCol
        self._add('ga_gemm', (C.byref(d),), label, keep=(d, A, B, Cout, bias, H, rowscale, R, colsum, colsumsq, C2))

    # -- global attention (ViT blocks) ---------------------------------------------------------------
    def attn_desc(self, qkv, out, lse, B, N, H, hd, scale, dtype, ldq=None, ldo=None):
        d = L.AttnDesc()
        d.B, d.N, d.H, d.hd, d.scale, d.dtype = B, N, H, hd, scale, dtype
        d.qkv, d.ldq, d.out, d.ldo, d.lse = _ptr(qkv), (3 * H * hd if ldq is None else ldq), _ptr(out), (H * hd if ldo is None else ldo), _ptr(lse)
        self.keep.extend([d, qkv, out, lse])
        return d

    def attn_fwd(self, d, label=None):
        self._add('ga_attn_fwd', (C.byref(d),), label, keep=(d,))

    def attn_bwd(self, d, dout, dqkv, ws, label=None):
        """ws: fp32 tensor of B*H*N elements (delta)"""
        self._add('ga_attn_bwd', (C.byref(d), _ptr(dout), _ptr(dqkv), _ptr(ws), ws.numel() * ws.element_size()), label,
                  keep=(d, dout, dqkv, ws))

    def patchify(self, x, out, P, dtype, label=None):
        B, CH, H, W = x.shape
        self._add('ga_patchify', (_ptr(x), _ptr(out), B, CH, H, W, P, dtype), label, keep=(x, out))

    def patchify_strided(self, x, out, P, S, dtype, label=None):
        B, CH, H, W = x.shape
        self._add('ga_patchify_strided', (_ptr(x), _ptr(out), B, CH, H, W, P, S, dtype), label, keep=(x, out))

    # -- pooling transformer (PiT) pieces ----------------------------------------------------------
    def pos_add_fwd(self, tok, pos, x0, B, Np, Cdim, dtype, label=None):
        self._add('ga_pos_add_fwd', (_ptr(tok), _ptr(pos), _ptr(x0), B, Np, Cdim, dtype), label, keep=(tok, pos, x0))

    def pos_add_bwd(self, dx0, dpos, B, Np, Cdim, dtype, label=None):
        self._add('ga_pos_add_bwd', (_ptr(dx0), _ptr(dpos), B, Np, Cdim, dtype), label, keep=(dx0, dpos))

    def dwpool_fwd(self, x, w, bias, y, B, H, W_, Cin, mult, dtype, label=None):
        self._add('ga_dwpool_fwd', (_ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, H, W_, Cin, mult, dtype), label, keep=(x, w, bias, y))

    def dwpool_bwd_data(self, dy, w, dx, B, H, W_, Cin, mult, dtype, label=None):
        self._add('ga_dwpool_bwd_data', (_ptr(dy), _ptr(w), _ptr(dx), B, H, W_, Cin, mult, dtype), label, keep=(dy, w, dx))

    def dwpool_bwd_weight(self, dy, x, dw, db, B, H, W_, Cin, mult, dtype, label=None):
        self._add('ga_dwpool_bwd_weight', (_ptr(dy), _ptr(x), _ptr(dw), _ptr(db), B, H, W_, Cin, mult, dtype), label, keep=(dy, x, dw, db))

    def resize_concat_fwd(self, src, dst, B, Hin, Win, Cdim, Hout, Wout, ldd, c_off, dtype, label=None):
        self._add('ga_resize_concat_fwd', (_ptr(src), _ptr(dst), B, Hin, Win, Cdim, Hout, Wout, ldd, c_off, dtype), label, keep=(src, dst))

    def resize_concat_bwd(self, dcat, dsrc, B, Hin, Win, Cdim, Hout, Wout, ldd, c_off, dtype, label=None):
        self._add('ga_resize_concat_bwd', (_ptr(dcat), _ptr(dsrc), B, Hin, Win, Cdim, Hout, Wout, ldd, c_off, dtype), label, keep=(dcat, dsrc))

    def rows_bcast(self, src, dst, B, HW, Cdim, scale, dtype, label=None):
        self._add('ga_rows_bcast', (_ptr(src), _ptr(dst), B, HW, Cdim, scale, dtype), label, keep=(src, dst))

    def vit_embed_fwd(self, tok, cls, pos, x0, B, Np, Cdim, dtype, label=None):
        self._add('ga_vit_embed_fwd', (_ptr(tok), _ptr(cls), _ptr(pos), _ptr(x0), B, Np, Cdim, dtype), label, keep=(tok, cls, pos, x0))

    def vit_embed_bwd(self, dx0, dtok, dcls, dpos, B, Np, Cdim, dtype, label=None):
        self._add('ga_vit_embed_bwd', (_ptr(dx0), _ptr(dtok), _ptr(dcls), _ptr(dpos), B, Np, Cdim, dtype), label, keep=(dx0, dtok, dcls, dpos))

    # -- alignment-free forms (odd-width variants) -------------------------------------------------
    def small_linear_desc(self, A, W, Y, rows, groups, Ng, Kg, dtype, lda, a_gstride, ldy, bias=None, a_perm=None, col_scale=None,
                          rowscale=None, rows_per_scale=1, R=None, ldr=0, Yraw=None):
        d = L.SmallLinearDesc()
        d.rows, d.groups, d.Ng, d.Kg, d.dtype = rows, groups, Ng, Kg, dtype
        d.A, d.lda, d.a_gstride, d.a_perm = _ptr(A), lda, a_gstride, _ptr(a_perm)
        d.W, d.bias, d.col_scale = _ptr(W), _ptr(bias), _ptr(col_scale)
        d.rowscale, d.rows_per_scale, d.R, d.ldr = _ptr(rowscale), rows_per_scale, _ptr(R), ldr
        d.Y, d.ldy, d.Yraw = _ptr(Y), ldy, _ptr(Yraw)
        self.keep.extend([d, A, W, Y, bias, a_perm, col_scale, rowscale, R, Yraw])
        return d

    def small_linear_fwd(self, d, label=None):
        self._add('ga_small_linear_fwd', (C.byref(d),), label, keep=(d,))

    def small_linear_bwd(self, d, dY, dA=None, accumulate_dA=False, dW=None, dbias=None, dcol_scale=None, label=None):
        self._add('ga_small_linear_bwd', (C.byref(d), _ptr(dY), _ptr(dA), int(accumulate_dA), _ptr(dW), _ptr(dbias), _ptr(dcol_scale)),
                  label, keep=(d, dY, dA, dW, dbias, dcol_scale))

    def colstats(self, x, ld, rows, Cdim, s, q, dtype, label=None):
        self._add('ga_colstats', (_ptr(x), ld, rows, Cdim, _ptr(s), _ptr(q), dtype), label, keep=(x, s, q))

    def pad_copy_f32(self, src, dst, rows, cols, lds, ldd, accumulate=False, label=None):
        self._add('ga_pad_copy_f32', (_ptr(src), _ptr(dst), rows, cols, lds, ldd, int(accumulate)), label, keep=(src, dst))

    def stem4_ln_fwd(self, x, W, ldw, bias, gamma, beta, pre, y, mean, rstd, B, H, W_, Cdim, eps, label=None):
        """ConvNeXt stem conv (4 x 4 / 4 from the fp32 NCHW image) + bias + LayerNorm in one bf16 launch"""
        self._add('ga_stem4_ln_fwd', (_ptr(x), _ptr(W), ldw, _ptr(bias), _ptr(gamma), _ptr(beta), _ptr(pre), _ptr(y), _ptr(mean), _ptr(rstd),
                                      B, H, W_, Cdim, eps), label, keep=(x, W, bias, gamma, beta, pre, y, mean, rstd))

    def blockdiag_f32(self, src, dst, R, rg, ng, cg, ld, to_diag=True, accumulate=False, label=None):
        self._add('ga_blockdiag_f32', (_ptr(src), _ptr(dst), R, rg, ng, cg, ld, int(to_diag), int(accumulate)), label, keep=(src, dst))

    def pad_groups_f32(self, src, dst, R, Cdim, RG, RGp, CG, CGp, unpad=False, accumulate=False, label=None):
        self._add('ga_pad_groups_f32', (_ptr(src), _ptr(dst), R, Cdim, RG, RGp, CG, CGp, int(unpad), int(accumulate)), label,
                  keep=(src, dst))

    def pad_copy(self, src, dst, rows, cols, lds, ldd, dtype, accumulate=False, label=None):
        self._add('ga_pad_copy', (_ptr(src), _ptr(dst), rows, cols, lds, ldd, int(accumulate), dtype), label, keep=(src, dst))

    def mlp_fwd(self, X, W1, b1, W2, b2, Y, M, Cdim, dtype, ldw1=None, ldw2=None, R=None, rowscale=None, rows_per_scale=1, label=None):
        """fused fc1 -> GELU -> fc2 (+ DropPath row scale + residual): Y = R + rowscale * (gelu(X W1^T + b1) W2^T + b2)"""
        d = L.MlpDesc()
        H = 4 * Cdim
        d.X, d.ldx, d.W1, d.ldw1, d.b1 = _ptr(X), Cdim, _ptr(W1), (Cdim if ldw1 is None else ldw1), _ptr(b1)
        d.W2, d.ldw2, d.b2 = _ptr(W2), (H if ldw2 is None else ldw2), _ptr(b2)
        d.R, d.ldr, d.rowscale, d.rows_per_scale = _ptr(R), Cdim, _ptr(rowscale), rows_per_scale
        d.Y, d.ldy, d.M, d.C, d.H, d.dtype = _ptr(Y), Cdim, M, Cdim, H, dtype
        self._add('ga_mlp_fwd', (C.byref(d),), label, keep=(d, X, W1, b1, W2, b2, R, rowscale, Y))

    def mlp_bwd(self, X, DY, W1, b1, W2T, W1T, A, DH, DX, M, Cdim, dtype, ldw1=None, ldw2t=None, ldw1t=None, label=None):
        """fused dgrad2 -> dgrad1 with the hidden pre-activation re-computed: A = gelu(X W1^T + b1), DH = (DY W2) gelu'(.), DX = DH W1"""
        d = L.MlpBwdDesc()
        H = 4 * Cdim
        d.X, d.ldx, d.DY, d.lddy = _ptr(X), Cdim, _ptr(DY), Cdim
        d.W1, d.ldw1, d.b1 = _ptr(W1), (Cdim if ldw1 is None else ldw1), _ptr(b1)
        d.W2T, d.ldw2t, d.W1T, d.ldw1t = _ptr(W2T), (Cdim if ldw2t is None else ldw2t), _ptr(W1T), (H if ldw1t is None else ldw1t)
        d.A, d.lda, d.DH, d.lddh, d.DX, d.lddx = _ptr(A), H, _ptr(DH), H, _ptr(DX), Cdim
        d.M, d.C, d.H, d.dtype = M, Cdim, H, dtype
        self._add('ga_mlp_bwd', (C.byref(d),), label, keep=(d, X, DY, W1, b1, W2T, W1T, A, DH, DX))

    def wgrad(self, Y, X, dW, M, N, K, dtype, ldy=None, ldx=None, ldw=None, batch=1, strideY=0, strideX=0, strideW=0,
              x_kind=A_PLAIN, x_dims=(0, 0, 0), x_act=ACT_NONE, dbias=None, strideDbias=0, alpha=1.0, split_m=None,
              accumulate=True, x_batch_mod=0, label=None):
        d = L.WgradDesc()
        d.M, d.N, d.K, d.batch, d.dtype = M, N, K, batch, dtype
        d.Y, d.ldy, d.strideY = _ptr(Y), (N if ldy is None else ldy), strideY
        d.X, d.ldx, d.strideX, d.x_kind = _ptr(X), (K if ldx is None else ldx), strideX, x_kind
        d.x_batch_mod = x_batch_mod
        d.x_H, d.x_W, d.x_C = x_dims
        d.x_act = x_act
        d.dW, d.ldw, d.strideW = _ptr(dW), (K if ldw is None else ldw), strideW
        d.dbias, d.strideDbias = _ptr(dbias), strideDbias
        d.alpha = alpha
        if split_m is None:
            split_m = pick_split_m(M, N, K, batch, dtype)
        d.split_m, d.accumulate = split_m, int(accumulate)
        need = self.lib.ga_wgrad_workspace(C.byref(d))
        if need:
            def patch(ptr, nbytes, d=d):
                d.workspace, d.ws_bytes = ptr, nbytes
            self._want_workspace(int(need), patch)
        self._add('ga_wgrad', (C.byref(d),), label, keep=(d, Y, X, dW, dbias))

    def weight_prep(self, w, G, Co, Ci, KH, KW, dtype, out=None, ldo=0, outT=None, ldt=0, rs=None, cs=None, flip=False,
                    stem=False, row_perm=None, t_cols=0, label=None):
        d = L.WprepDesc()
        d.w, d.G, d.Co, d.Ci, d.KH, d.KW = _ptr(w), G, Co, Ci, KH, KW
        d.rs, d.cs, d.row_perm, d.dtype = _ptr(rs), _ptr(cs), _ptr(row_perm), dtype
        d.out, d.ldo, d.outT, d.ldt, d.flip, d.stem = _ptr(out), ldo, _ptr(outT), ldt, int(flip), int(stem)
        d.t_cols = t_cols
        if self.defer:
            self._pend['wprep'].append(d)
            self.keep.extend((w, out, outT, rs, cs, row_perm))
            return
        self._add('ga_weight_prep', (C.byref(d),), label, keep=(d, w, out, outT, rs, cs, row_perm))

    def bias_fold(self, W, b, rs, v, be, N, Cdim, row_perm=None, label=None):
        if self.defer:
            self._small(kind=2, y=_ptr(be), x=_ptr(W), R=N, C=Cdim, b=_ptr(b), rs=_ptr(rs), v=_ptr(v), row_perm=_ptr(row_perm))
            self.keep.extend((W, b, rs, v, be, row_perm))
            return
        self._add('ga_bias_fold', (_ptr(W), _ptr(b), _ptr(rs), _ptr(v), _ptr(row_perm), _ptr(be), N, Cdim), label,
                  keep=(W, b, rs, v, be, row_perm))

    def weight_unfold(self, G, ldg, N, Ci, KH=1, KW=1, gb=None, W=None, b=None, rs=None, cs=None, v=None, stem=False, dW=None,
                      db=None, d_rs=None, d_cs=None, d_v=None, row_perm=None, label=None):
        d = L.WunfoldDesc()
        d.G, d.ldg, d.gb, d.W, d.b, d.rs, d.cs, d.v = _ptr(G), ldg, _ptr(gb), _ptr(W), _ptr(b), _ptr(rs), _ptr(cs), _ptr(v)
        d.row_perm = _ptr(row_perm)
        d.N, d.Ci, d.KH, d.KW, d.stem = N, Ci, KH, KW, int(stem)
        d.dW, d.db, d.d_rs, d.d_cs, d.d_v = _ptr(dW), _ptr(db), _ptr(d_rs), _ptr(d_cs), _ptr(d_v)
        if self.defer:
            self._pend['unfold'].append(d)
            self.keep.extend((G, gb, W, b, rs, cs, v, dW, db, d_rs, d_cs, d_v, row_perm))
            return
        self._add('ga_weight_unfold', (C.byref(d),), label, keep=(d, G, gb, W, b, rs, cs, v, dW, db, d_rs, d_cs, d_v, row_perm))

    # -- depthwise conv / norms ---------------------------------------------------------------------
    def dwconv7_fwd(self, x, w49, bias, y, B, H, W, Cdim, dtype, label=None):
        self._add('ga_dwconv7_fwd', (_ptr(x), _ptr(w49), _ptr(bias), _ptr(y), B, H, W, Cdim, dtype), label,
                  keep=(x, w49, bias, y))

    def dwconv7_bwd_data(self, dy, w49, res, dx, B, H, W, Cdim, dtype, dx2=None, scale2=None, label=None):
        if dx2 is not None:
            self._add('ga_dwconv7_bwd_data2', (_ptr(dy), _ptr(w49), _ptr(res), _ptr(dx), _ptr(dx2), _ptr(scale2), B, H, W,
                                               Cdim, dtype), label, keep=(dy, w49, res, dx, dx2, scale2))
            return
        self._add('ga_dwconv7_bwd_data', (_ptr(dy), _ptr(w49), _ptr(res), _ptr(dx), B, H, W, Cdim, dtype), label,
                  keep=(dy, w49, res, dx))

    def dwconv7_bwd_weight(self, dy, x, dw49, dbias, B, H, W, Cdim, dtype, label=None):
        need = int(self.lib.ga_dwconv7_bwd_weight_workspace(B, H, W, Cdim, dtype))
        head = (_ptr(dy), _ptr(x), _ptr(dw49), _ptr(dbias), B, H, W, Cdim, dtype)
        box = {}

        def patch(ptr, nbytes):
            box['ws'] = (ptr, nbytes)
            if 'idx' in box:
                fn, _, lb = self.calls[box['idx']]
                self.calls[box['idx']] = (fn, head + (ptr, nbytes), lb)
        self._want_workspace(need, patch)
        self._add('ga_dwconv7_bwd_weight', head + box.get('ws', (None, 0)), label, keep=(dy, x, dw49, dbias))
        box['idx'] = len(self.calls) - 1

    def layernorm_fwd(self, x, w, b, y, mean, rstd, rows, Cdim, eps, dtype, label=None):
        self._add('ga_layernorm_fwd', (_ptr(x), _ptr(w), _ptr(b), _ptr(y), _ptr(mean), _ptr(rstd), rows, Cdim, eps, dtype),
                  label, keep=(x, w, b, y, mean, rstd))

    def layernorm_bwd(self, g, x, mean, rstd, w, dres, dx, dw, db, rows, Cdim, x_is_normalized, dtype, label=None, dx2=None,
                      scale2=None, rows_per_scale=1):
        if dx2 is not None:        # second output: the stored dx times a per-sample scale (the consumer's DropPath factor)
            self._add('ga_layernorm_bwd_dp', (_ptr(g), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(w), _ptr(dres), _ptr(dx), _ptr(dw), _ptr(db),
                                              rows, Cdim, int(x_is_normalized), _ptr(dx2), _ptr(scale2), rows_per_scale, dtype), label,
                      keep=(g, x, mean, rstd, w, dres, dx, dw, db, dx2, scale2))
            return
        self._add('ga_layernorm_bwd', (_ptr(g), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(w), _ptr(dres), _ptr(dx), _ptr(dw),
                                       _ptr(db), rows, Cdim, int(x_is_normalized), dtype), label,
                  keep=(g, x, mean, rstd, w, dres, dx, dw, db))

    def bn_finalize(self, ssum, ssq, n, w, b, eps, momentum, rmean, rvar, mean_out, rstd_out, scale, shift, Cdim,
                    training, label=None):
        self._add('ga_bn_finalize', (_ptr(ssum), _ptr(ssq), n, _ptr(w), _ptr(b), eps, momentum, _ptr(rmean), _ptr(rvar),
                                     _ptr(mean_out), _ptr(rstd_out), _ptr(scale), _ptr(shift), Cdim, int(training)),
                  label, keep=(ssum, ssq, w, b, rmean, rvar, mean_out, rstd_out, scale, shift))

    def affine_act(self, x, scale, shift, res, y, rows, Cdim, relu, dtype, rowscale=None, rows_per_scale=1, ldx=0, label=None):
        self._add('ga_affine_act', (_ptr(x), _ptr(scale), _ptr(shift), _ptr(res), _ptr(rowscale), rows_per_scale, _ptr(y),
                                    rows, Cdim, int(relu), dtype, ldx), label, keep=(x, scale, shift, res, y, rowscale))

    def bn_bwd_reduce(self, dy, y_relu, x, mean, rstd, s1, s2, rows, Cdim, dtype, rowscale=None, rows_per_scale=1, ldx=0,
                      label=None):
        self._add('ga_bn_bwd_reduce', (_ptr(dy), _ptr(y_relu), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(rowscale),
                                       rows_per_scale, _ptr(s1), _ptr(s2), rows, Cdim, dtype, ldx), label,
                  keep=(dy, y_relu, x, mean, rstd, s1, s2, rowscale))

    def bn_bwd_apply(self, dy, y_relu, x, mean, rstd, w, s1, s2, n, dx, rows, Cdim, dtype, rowscale=None,
                     rows_per_scale=1, ldx=0, lddx=0, label=None):
        self._add('ga_bn_bwd_apply', (_ptr(dy), _ptr(y_relu), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(w), _ptr(s1), _ptr(s2),
                                      _ptr(rowscale), rows_per_scale, n, _ptr(dx), rows, Cdim, dtype, ldx, lddx), label,
                  keep=(dy, y_relu, x, mean, rstd, w, s1, s2, dx, rowscale))

    # -- head ---------------------------------------------------------------------------------------
    def pool_concat_fwd(self, src, dst, B, Hin, Win, Cdim, Hout, Wout, ldd, c_off, mode, dtype, label=None):
        self._add('ga_pool_concat_fwd', (_ptr(src), _ptr(dst), B, Hin, Win, Cdim, Hout, Wout, ldd, c_off, mode, dtype),
                  label, keep=(src, dst))

    def pool_concat_bwd(self, dcat, dres, dsrc, B, Hin, Win, Cdim, Hout, Wout, ldd, c_off, mode, dtype, label=None):
        self._add('ga_pool_concat_bwd', (_ptr(dcat), _ptr(dres), _ptr(dsrc), B, Hin, Win, Cdim, Hout, Wout, ldd, c_off,
                                         mode, dtype), label, keep=(dcat, dres, dsrc))

    def spatial_sum(self, a, b2, out, B, HW, Cdim, scale, dtype, label=None):
        self._add('ga_spatial_sum', (_ptr(a), _ptr(b2), _ptr(out), B, HW, Cdim, scale, dtype), label, keep=(a, b2, out))

    def se_mlp_fwd(self, s, W1, b1, W2, b2, hid, gate, B, Cdim, R, label=None):
        self._add('ga_se_mlp_fwd', (_ptr(s), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(hid), _ptr(gate), B, Cdim, R),
                  label, keep=(s, W1, b1, W2, b2, hid, gate))

    def se_mlp_bwd(self, dgate, gate, hid, s, W1, W2, ds, dW1, db1, dW2, db2, B, Cdim, R, ds_scale=1.0, label=None):
        self._add('ga_se_mlp_bwd', (_ptr(dgate), _ptr(gate), _ptr(hid), _ptr(s), _ptr(W1), _ptr(W2), _ptr(ds), ds_scale, _ptr(dW1),
                                    _ptr(db1), _ptr(dW2), _ptr(db2), B, Cdim, R), label,
                  keep=(dgate, gate, hid, s, W1, W2, ds, dW1, db1, dW2, db2))

    def chan_scale(self, x, g, add, y, B, HW, Cdim, dtype, label=None):
        self._add('ga_chan_scale', (_ptr(x), _ptr(g), _ptr(add), _ptr(y), B, HW, Cdim, dtype), label, keep=(x, g, add, y))

    def gram_pack_fwd(self, G, out, inv_norm, B, Cdim, groups, Kp, dtype, label=None):
        self._add('ga_gram_pack_fwd', (_ptr(G), _ptr(out), _ptr(inv_norm), B, Cdim, groups, Kp, dtype), label,
                  keep=(G, out, inv_norm))

    def gram_pack_bwd(self, dvec, vhat, inv_norm, S, B, Cdim, groups, Kp, dtype, label=None):
        self._add('ga_gram_pack_bwd', (_ptr(dvec), _ptr(vhat), _ptr(inv_norm), _ptr(S), B, Cdim, groups, Kp, dtype), label,
                  keep=(dvec, vhat, inv_norm, S))

    def token_cat(self, cls, tok, u, B, N, Cdim, dtype, label=None):
        self._add('ga_token_cat', (_ptr(cls), _ptr(tok), _ptr(u), B, N, Cdim, dtype), label, keep=(cls, tok, u))

    def token_split(self, du, dcls, dtok, B, N, Cdim, acc_cls, acc_tok, dtype, label=None):
        self._add('ga_token_split', (_ptr(du), _ptr(dcls), _ptr(dtok), B, N, Cdim, int(acc_cls), int(acc_tok), dtype),
                  label, keep=(du, dcls, dtok))

    def class_attn_fwd(self, q, kv, out, P, B, N, heads, hd, scale, dtype, label=None):
        self._add('ga_class_attn_fwd', (_ptr(q), _ptr(kv), _ptr(out), _ptr(P), B, N, heads, hd, scale, dtype), label,
                  keep=(q, kv, out, P))

    def class_attn_bwd(self, dout, q, kv, P, dq, dkv, B, N, heads, hd, scale, dtype, label=None):
        self._add('ga_class_attn_bwd', (_ptr(dout), _ptr(q), _ptr(kv), _ptr(P), _ptr(dq), _ptr(dkv), B, N, heads, hd,
                                        scale, dtype), label, keep=(dout, q, kv, P, dq, dkv))

    def class_attn_fwd2(self, q, kv_cls, kv_tok, out, P, B, N, heads, hd, scale, dtype, tok_ld=0, label=None):
        self._add('ga_class_attn_fwd2', (_ptr(q), _ptr(kv_cls), _ptr(kv_tok), tok_ld, _ptr(out), _ptr(P), B, N, heads, hd, scale,
                                         dtype),
                  label, keep=(q, kv_cls, kv_tok, out, P))

    def class_attn_bwd2(self, dout, q, kv_cls, kv_tok, P, dq, dkv_cls, dkv_tok, B, N, heads, hd, scale, dtype, tok_ld=0,
                        label=None):
        self._add('ga_class_attn_bwd2', (_ptr(dout), _ptr(q), _ptr(kv_cls), _ptr(kv_tok), tok_ld, _ptr(P), _ptr(dq), _ptr(dkv_cls),
                                         _ptr(dkv_tok), B, N, heads, hd, scale, dtype), label,
                  keep=(dout, q, kv_cls, kv_tok, P, dq, dkv_cls, dkv_tok))

    # -- GA-CSWin ----------------------------------------------------------------------------------
    def cswin_desc(self, qkv, out, B, reso, Cdim, heads, stripes, lepe, scale, dtype, ldq=None, ldo=None):
        """descriptor of one CSWinBlock's stripe attention: stripes = [(Hs, Ws)] per branch, lepe = [(w [Cb,1,3,3], b [Cb])]"""
        d = L.CswinAttnDesc()
        d.B, d.reso, d.C, d.heads, d.nbranch = B, reso, Cdim, heads, len(stripes)
        for i, ((hs, ws), (w, b)) in enumerate(zip(stripes, lepe)):
            d.Hs[i], d.Ws[i] = hs, ws
            d.lepe_w[i], d.lepe_b[i] = _ptr(w), _ptr(b)
        d.scale, d.dtype = scale, dtype
        d.qkv, d.ldq = _ptr(qkv), (3 * Cdim if ldq is None else ldq)
        d.out, d.ldo = _ptr(out), (Cdim if ldo is None else ldo)
        self.keep.extend([d, qkv, out] + [t for pair in lepe for t in pair])
        return d

    def cswin_attn_fwd(self, d, label=None):
        self._add('ga_cswin_attn_fwd', (C.byref(d),), label, keep=(d,))

    def cswin_attn_bwd(self, d, dout, dqkv, lepe_ws=None, label=None):
        """lepe_ws: fp32 buffer of cswin_attn_bwd_workspace(d) bytes -> the kernel also leaves the LePE weight-gradient partials
        there (cswin_lepe_wgrad_reduce adds them up); None: use cswin_lepe_wgrad"""
        nbytes = 0 if lepe_ws is None else lepe_ws.numel() * lepe_ws.element_size()
        self._add('ga_cswin_attn_bwd', (C.byref(d), _ptr(dout), _ptr(dqkv), _ptr(lepe_ws), nbytes), label, keep=(d, dout, dqkv, lepe_ws))

    def cswin_lepe_wgrad_reduce(self, d, lepe_ws, grads, label=None):
        g = list(grads) + [(None, None)] * (2 - len(grads))
        self._add('ga_cswin_lepe_wgrad_reduce', (C.byref(d), _ptr(lepe_ws), _ptr(g[0][0]), _ptr(g[0][1]), _ptr(g[1][0]), _ptr(g[1][1])),
                  label, keep=(d, lepe_ws) + tuple(t for pair in grads for t in pair))

    def cswin_lepe_wgrad(self, d, dout, grads, label=None):
        """grads = [(dw, db)] per branch (fp32, accumulated into)"""
        g = list(grads) + [(None, None)] * (2 - len(grads))
        self._add('ga_cswin_lepe_wgrad', (C.byref(d), _ptr(dout), _ptr(g[0][0]), _ptr(g[0][1]), _ptr(g[1][0]), _ptr(g[1][1])),
                  label, keep=(d, dout) + tuple(t for pair in grads for t in pair))

    def layernorm_gelu_fwd(self, x, w, b, y, mean, rstd, rows, Cdim, eps, dtype, label=None):
        self._add('ga_layernorm_gelu_fwd', (_ptr(x), _ptr(w), _ptr(b), _ptr(y), _ptr(mean), _ptr(rstd), rows, Cdim, eps, dtype),
                  label, keep=(x, w, b, y, mean, rstd))

    def layernorm_gelu_bwd(self, g, x, mean, rstd, w, b, dx, dw, db, rows, Cdim, dtype, label=None):
        self._add('ga_layernorm_gelu_bwd', (_ptr(g), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(w), _ptr(b), _ptr(dx), _ptr(dw),
                                            _ptr(db), rows, Cdim, dtype), label, keep=(g, x, mean, rstd, w, b, dx, dw, db))

    def nchw3_to_nhwc8(self, x, y, B, H, W, dtype, label=None):
        self._add('ga_nchw3_to_nhwc8', (_ptr(x), _ptr(y), B, H, W, dtype), label, keep=(x, y))

    def convw_pack(self, w, out, Co, Ci, taps, Cp, ldo, dtype, label=None):
        self._add('ga_convw_pack', (_ptr(w), _ptr(out), Co, Ci, taps, Cp, ldo, dtype), label, keep=(w, out))

    def convw_unpack_grad(self, G, dW, Co, Ci, taps, Cp, ldg, label=None):
        self._add('ga_convw_unpack_grad', (_ptr(G), _ptr(dW), Co, Ci, taps, Cp, ldg), label, keep=(G, dW))

    def conv3s2_dgrad_prep(self, w, out, Co, Ci, ldo, dtype, label=None):
        self._add('ga_conv3s2_dgrad_prep', (_ptr(w), _ptr(out), Co, Ci, ldo, dtype), label, keep=(w, out))

    # -- MAP head -----------------------------------------------------------------------------------
    def gram_pack_fwd2(self, G, out, inv_norm, B, Cdim, groups, Kp, ntok, dtype, label=None):
        self._add('ga_gram_pack_fwd2', (_ptr(G), _ptr(out), _ptr(inv_norm), B, Cdim, groups, Kp, ntok, dtype), label,
                  keep=(G, out, inv_norm))

    def gram_pack_bwd2(self, dvec, vhat, inv_norm, S, B, Cdim, groups, Kp, ntok, dtype, label=None):
        self._add('ga_gram_pack_bwd2', (_ptr(dvec), _ptr(vhat), _ptr(inv_norm), _ptr(S), B, Cdim, groups, Kp, ntok, dtype), label,
                  keep=(dvec, vhat, inv_norm, S))

    def map_tokens_fwd(self, e, tok, B, Cdim, T, add_mean, dtype, label=None):
        self._add('ga_map_tokens_fwd', (_ptr(e), _ptr(tok), B, Cdim, T, int(add_mean), dtype), label, keep=(e, tok))

    def map_tokens_bwd(self, dtok, de, B, Cdim, T, add_mean, dtype, label=None):
        self._add('ga_map_tokens_bwd', (_ptr(dtok), _ptr(de), B, Cdim, T, int(add_mean), dtype), label, keep=(dtok, de))

    def class_attn_mt_fwd(self, q, kv_cls, kv_tok, tok_ld, out, P, mask, B, T, N, heads, hd, scale, dtype, label=None):
        self._add('ga_class_attn_mt_fwd', (_ptr(q), _ptr(kv_cls), _ptr(kv_tok), tok_ld, _ptr(out), _ptr(P), _ptr(mask), B, T, N, heads,
                                           hd, scale, dtype), label, keep=(q, kv_cls, kv_tok, out, P, mask))

    def class_attn_mt_bwd(self, dout, q, kv_cls, kv_tok, tok_ld, P, mask, dq, dkv_cls, dkv_tok, dtok_ld, B, T, N, heads, hd, scale,
                          dtype, label=None):
        self._add('ga_class_attn_mt_bwd', (_ptr(dout), _ptr(q), _ptr(kv_cls), _ptr(kv_tok), tok_ld, _ptr(P), _ptr(mask), _ptr(dq),
                                           _ptr(dkv_cls), _ptr(dkv_tok), dtok_ld, B, T, N, heads, hd, scale, dtype), label,
                  keep=(dout, q, kv_cls, kv_tok, P, mask, dq, dkv_cls, dkv_tok))

    def class_attn_mt_ia_fwd(self, q, kv_cls, kv_tok, tok_ld, out, P, mask, W1, b1, W2, b2, B, T, N, heads, hd, scale, dtype, label=None):
        self._add('ga_class_attn_mt_ia_fwd', (_ptr(q), _ptr(kv_cls), _ptr(kv_tok), tok_ld, _ptr(out), _ptr(P), _ptr(mask), _ptr(W1), _ptr(b1),
                                              _ptr(W2), _ptr(b2), B, T, N, heads, hd, scale, dtype), label,
                  keep=(q, kv_cls, kv_tok, out, P, mask, W1, b1, W2, b2))

    def class_attn_mt_ia_bwd(self, dout, q, kv_cls, kv_tok, tok_ld, P, mask, W1, W2, b2, dq, dkv_cls, dkv_tok, dtok_ld, dW1, db1, dW2, db2,
                             B, T, N, heads, hd, scale, dtype, label=None):
        self._add('ga_class_attn_mt_ia_bwd', (_ptr(dout), _ptr(q), _ptr(kv_cls), _ptr(kv_tok), tok_ld, _ptr(P), _ptr(mask), _ptr(W1), _ptr(W2),
                                              _ptr(b2), _ptr(dq), _ptr(dkv_cls), _ptr(dkv_tok), dtok_ld, _ptr(dW1), _ptr(db1), _ptr(dW2),
                                              _ptr(db2), B, T, N, heads, hd, scale, dtype), label,
                  keep=(dout, q, kv_cls, kv_tok, P, mask, W1, W2, b2, dq, dkv_cls, dkv_tok, dW1, db1, dW2, db2))

    def map_loss_fwd_bwd(self, org, avg, target, loss, dorg, davg, K, B, NC, lam, kind, smoothing, grad_scale, dtype, label=None):
        self._add('ga_map_loss_fwd_bwd', (_ptr(org), _ptr(avg), _ptr(target), _ptr(loss), _ptr(dorg), _ptr(davg), K, B, NC, lam, kind,
                                          smoothing, grad_scale, dtype), label, keep=(org, avg, target, loss, dorg, davg))

    def loss_dense_fwd_bwd(self, org, avg, target, dense, loss, dorg, davg, K, B, NC, lam, kind, smoothing, bce_threshold, grad_scale,
                           dtype, label=None):
        """GA (avg None) / MAP loss on class indices (`target`) or on a dense [B, NC] fp32 target (`dense`: mixup / cutmix)"""
        self._add('ga_loss_dense_fwd_bwd', (_ptr(org), _ptr(avg), _ptr(target), _ptr(dense), _ptr(loss), _ptr(dorg), _ptr(davg), K, B, NC,
                                            lam, kind, smoothing, bce_threshold, grad_scale, dtype), label,
                  keep=(org, avg, target, dense, loss, dorg, davg))

    def u8_normalize(self, x, out, mean, std, label=None):
        """uint8 (B, C, H, W) -> fp32, (x - mean[c]) / std[c]; mean / std: python sequences in 0..255 units"""
        B, CH, H, W = x.shape
        m = (C.c_float * CH)(*[float(v) for v in mean])
        s = (C.c_float * CH)(*[float(v) for v in std])
        self._add('ga_u8_normalize', (_ptr(x), _ptr(out), B, CH, H, W, m, s), label, keep=(x, out, m, s))

    def mixup_batch(self, x, out, lam, cutmix=False, box=(0, 0, 0, 0), label=None):
        B, CH, H, W = x.shape
        yl, yh, xl, xh = (int(v) for v in box)
        self._add('ga_mixup_batch', (_ptr(x), _ptr(out), B, CH, H, W, float(lam), int(cutmix), yl, yh, xl, xh), label, keep=(x, out))

    def mixup_target(self, target, out, NC, lam, smoothing, label=None):
        self._add('ga_mixup_target', (_ptr(target), _ptr(out), target.numel(), NC, float(lam), float(smoothing)), label, keep=(target, out))

    def agc_clip(self, params, grads, units, nunits, clip_factor, eps=1e-3, label=None):
        self._add('ga_agc_clip', (_ptr(params), _ptr(grads), _ptr(units), nunits, float(clip_factor), float(eps)), label,
                  keep=(params, grads, units))

    def gelu_fwd(self, x, y, n, dtype, label=None):
        self._add('ga_gelu_fwd', (_ptr(x), _ptr(y), n, dtype), label, keep=(x, y))

    def gelu_bwd(self, dy, x, dx, n, dtype, label=None):
        self._add('ga_gelu_bwd', (_ptr(dy), _ptr(x), _ptr(dx), n, dtype), label, keep=(dy, x, dx))

    def relu_drop(self, a, mask, out, deriv, n, dtype, label=None):
        self._add('ga_relu_drop', (_ptr(a), _ptr(mask), _ptr(out), _ptr(deriv), n, dtype), label, keep=(a, mask, out, deriv))

    def mask_mul(self, x, mask, res, y, n, dtype, label=None):
        self._add('ga_mask_mul', (_ptr(x), _ptr(mask), _ptr(res), _ptr(y), n, dtype), label, keep=(x, mask, res, y))

    def copy2d(self, src, lds, dst, ldd, rows, cols, dtype, accumulate=False, label=None):
        self._add('ga_copy2d', (_ptr(src), lds, _ptr(dst), ldd, rows, cols, int(accumulate), dtype), label, keep=(src, dst))

    # -- loss / metric / optimizer ------------------------------------------------------------------
    def loss_fwd_bwd(self, logits, target, loss, dlogits, K, B, NC, lam, kind, smoothing, grad_scale, dtype, label=None):
        self._add('ga_loss_fwd_bwd', (_ptr(logits), _ptr(target), _ptr(loss), _ptr(dlogits), K, B, NC, lam, kind,
                                      smoothing, grad_scale, dtype), label, keep=(logits, target, loss, dlogits))

    def heads_topk(self, logits, K, B, NC, topk, out_sum, out_idx, label=None):
        self._add('ga_heads_topk', (_ptr(logits), K, B, NC, topk, _ptr(out_sum), _ptr(out_idx)), label,
                  keep=(logits, out_sum, out_idx))

    def sgd_step(self, p, g, buf, hp, n, nesterov, wd_mult, label=None):
        self._add('ga_sgd_step', (_ptr(p), _ptr(g), _ptr(buf), _ptr(hp), n, int(nesterov), wd_mult), label,
                  keep=(p, g, buf, hp))

    def adamw_step(self, p, g, m, v, hp, n, wd_mult, label=None):
        self._add('ga_adamw_step', (_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(hp), n, wd_mult), label,
                  keep=(p, g, m, v, hp))

    def drop_path_sample(self, out, keep, sites, B, seed, counter, label=None):
        self._add('ga_drop_path_sample', (_ptr(out), _ptr(keep), sites, B, int(seed) & ((1 << 64) - 1), _ptr(counter)), label,
                  keep=(out, keep, counter))

    def dropout_mask_sample(self, out, n, keep, seed, counter, label=None):
        self._add('ga_dropout_mask_sample', (_ptr(out), n, float(keep), int(seed) & ((1 << 64) - 1), _ptr(counter)), label,
                  keep=(out, counter))

    # -- utilities ----------------------------------------------------------------------------------
    def transpose_f32(self, src, dst, R, Cdim, accumulate=False, label=None):
        if self.defer:
            self._small(kind=1, y=_ptr(dst), x=_ptr(src), R=R, C=Cdim, accumulate=int(accumulate))
            self.keep.extend((src, dst))
            return
        self._add('ga_transpose_f32', (_ptr(src), _ptr(dst), R, Cdim, int(accumulate)), label, keep=(src, dst))

    def axpy_f32(self, y, x, a, n, label=None):
        if self.defer:
            self._small(kind=0, y=_ptr(y), x=_ptr(x), a=a, n=n)
            self.keep.extend((y, x))
            return
        self._add('ga_axpy_f32', (_ptr(y), _ptr(x), a, n), label, keep=(y, x))

    def lamb_stage1(self, p, g, m, v, u, hp, gsumsq, chunks, nchunks, norms, label=None):
        self._add('ga_lamb_stage1', (_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(u), _ptr(hp), _ptr(gsumsq), _ptr(chunks), nchunks,
                                     _ptr(norms)), label, keep=(p, g, m, v, u, hp, gsumsq, chunks, norms))

    def lamb_stage2(self, p, u, hp, chunks, nchunks, norms, label=None):
        self._add('ga_lamb_stage2', (_ptr(p), _ptr(u), _ptr(hp), _ptr(chunks), nchunks, _ptr(norms)), label,
                  keep=(p, u, hp, chunks, norms))

    def lerp_f32(self, y, x, w, n, label=None):
        self._add('ga_lerp_f32', (_ptr(y), _ptr(x), float(w), n), label, keep=(y, x))

    def sumsq_f32(self, x, n, out, label=None):
        self._add('ga_sumsq_f32', (_ptr(x), n, _ptr(out)), label, keep=(x, out))

    def clip_grad_f32(self, g, n, sumsq, limit, mode, label=None):
        self._add('ga_clip_grad_f32', (_ptr(g), n, _ptr(sumsq), float(limit), int(mode)), label, keep=(g, sumsq))

    def rowscale(self, x, s, y, n, elems_per_scale, dtype, label=None):
        self._add('ga_rowscale', (_ptr(x), _ptr(s), _ptr(y), n, elems_per_scale, dtype), label, keep=(x, s, y))

    def cast_from_f32(self, src, dst, n, dtype, label=None):
        self._add('ga_cast_from_f32', (_ptr(src), _ptr(dst), n, dtype), label, keep=(src, dst))

    def cast_to_f32(self, src, dst, n, dtype, label=None):
        self._add('ga_cast_to_f32', (_ptr(src), _ptr(dst), n, dtype), label, keep=(src, dst))

    def zero(self, t, label=None):
        """memset a persistent buffer (hipMemsetAsync on the plan's stream)."""
        self._add('ga_memset', (_ptr(t), 0, t.numel() * t.element_size()), label or 'zero', keep=(t,))


def mlp_supported(Cdim, H, dtype):
    return bool(L.load().ga_mlp_supported(Cdim, H, dtype))


def cswin_attn_bwd_workspace(d):
    """bytes of the LePE partial buffer Plan.cswin_attn_bwd takes for this descriptor (0: generic form, no fused partials)"""
    return int(L.load().ga_cswin_attn_bwd_workspace(C.byref(d)))


def zero_(t):
    """t.zero_() as a hipMemsetAsync on the current stream through the C ABI"""
    L.check(L.load().ga_memset(_ptr(t), 0, t.numel() * t.element_size(), current_stream_ptr()), 'ga_memset')
    return t


_NUM_CU = None


def num_cus():
    global _NUM_CU
    if _NUM_CU is None:
        n = C.c_int(256)
        L.check(L.load().ga_device_info(C.byref(n), None, None), 'ga_device_info')
        _NUM_CU = n.value
    return _NUM_CU


def pick_split_m(M, N, K, batch, dtype):
    """Number of row-range splits of a wgrad so that ~2 workgroups per CU are in flight, each with >= 4 slabs."""
    slab = 64 if dtype == GA_BF16 else 32
    tiles = ((N + 127) // 128) * ((K + 127) // 128) * batch
    want = max(1, (2 * 256) // tiles)
    return int(max(1, min(want, (M + 4 * slab - 1) // (4 * slab))))
