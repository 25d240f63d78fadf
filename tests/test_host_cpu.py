"""CPU: host logic of the product package -- the C-ABI library loads and exports every symbol include/gaext.h
declares (no compute without a GPU), registry / factories / state_dict surface, flat parameter layout, gradient
buckets + a world-size-2 gloo all-reduce over them, LR schedule, and the "no CPU fallback" contract."""
import os
import re
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from imagenet_models_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, 'include', 'gaext.h')).read()
    declared = sorted(set(re.findall(r'^\s*(?:int|size_t)\s+(ga_\w+)\s*\(', hdr, flags=re.M)))
    assert len(declared) >= 38
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/gaext.h but not exported by libgaext.so'
    assert sorted(_lib.exported_symbols()) == declared, 'ctypes signature table out of sync with the header'
    assert lib.ga_version() >= 100


def test_tuning_knobs_are_a_table_not_the_environment():
    """include/gaext.h ga_set_knob / ga_unset_knob / ga_config_string: no launch reads the environment; a non-default knob shows
    up in the configuration string (what a bench or parity record prints); result-changing debug switches are compiled out"""
    import subprocess
    from imagenet_models_amd import _lib
    lib = _lib.load()
    base = _lib.config_string()
    assert base.startswith('libgaext ') and 'gfx950' in base and 'DEBUG-BUILD' not in base
    assert 'NT_DMA' not in base
    with _lib.knobs(NT_DMA=2, TN2=0):
        s = _lib.config_string()
        assert 'NT_DMA=2(api)' in s and 'TN2=0(api)' in s, s
    assert _lib.config_string() == base
    assert lib.ga_set_knob(b'', 1) != 0 and lib.ga_set_knob(b'X' * 40, 1) != 0
    # getenv appears in the runtime's knob table only; the debug knob of the ping-pong GEMM body is not in a release build
    for f in os.listdir(os.path.join(ROOT, 'imagenet-models_amd', 'csrc')):
        if f.endswith('.hip') and f != 'runtime.hip':
            assert 'getenv' not in open(os.path.join(ROOT, 'imagenet-models_amd', 'csrc', f)).read(), f
    strs = subprocess.run(['strings', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert 'PP_DBG' not in strs


def test_library_never_allocates_device_memory():
    """include/gaext.h: every pointer is caller-owned; the two reducing entry points take a caller workspace"""
    import subprocess
    from imagenet_models_amd import _lib
    syms = subprocess.run(['nm', '-D', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for banned in ('hipMalloc', 'hipFree', 'hipMallocAsync', 'hipDeviceSynchronize', 'hipStreamSynchronize'):
        assert not re.search(r'\bU ' + banned + r'\b', syms), f'libgaext.so references {banned}'
    lib = _lib.load()
    assert lib.ga_dwconv7_bwd_weight_workspace(2, 14, 14, 64, _lib.GA_BF16) > 0


def test_descriptor_structs_match_header_field_order():
    from imagenet_models_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'gaext.h')).read()

    def fields(struct_name):
        end = hdr.index('} ' + struct_name + ';')
        body = hdr[hdr.rindex('typedef struct {', 0, end) + len('typedef struct {'):end]
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        names = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            decl = re.sub(r'^(const\s+)?(void|float|int64_t|int)\s*\*?', '', decl).strip()
            for part in decl.split(','):
                names.append(part.strip().lstrip('*').strip())
        return names

    for cname, cls in (('ga_gemm_desc', _lib.GemmDesc), ('ga_wgrad_desc', _lib.WgradDesc),
                       ('ga_wprep_desc', _lib.WprepDesc), ('ga_wunfold_desc', _lib.WunfoldDesc)):
        assert fields(cname) == [f[0] for f in cls._fields_], cname


def test_registry_and_factories():
    import imagenet_models_amd as A
    import warnings
    from imagenet_models_amd import registry
    names = A.list_models()
    for n in ('ga_convnext_tiny_688', 'ga_convnext_tiny_768', 'ga_convnext_small_688', 'ga_convnext_small_768',
              'ga_convnext_base_976', 'ga_convnext_base_1024'):      # all six registered variants (ga_convnext.py:572-613) run on the engine
        assert n in names and A.is_model(n) and registry.is_supported(n)
    # timm create_model drops None kwargs (GA/train.py:407-420 passes many that are None)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        m = A.create_model('ga_convnext_tiny_688', pretrained=False, num_classes=10, drop_rate=None, drop_path_rate=0.1,
                           drop_block_rate=None, global_pool=None, bn_momentum=None, bn_eps=None, scriptable=None)
    assert m.num_classes == 10 and m.cfg['dims'][-1] == 688 and m.cfg['dim_embed'] == 168
    assert sum(p.numel() for p in A.create_model('ga_convnext_tiny_688').parameters()) == 47821324
    # an entry point the engine cannot run stays constructible but is hidden from list_models() and warns when created
    registry.register_model(lambda pretrained=False, **kw: A.create_model('ga_convnext_tiny_768', **kw), name='_test_unsupported_entry')
    registry.mark_unsupported('_test_unsupported_entry', 'refuses: test entry')
    try:
        assert '_test_unsupported_entry' not in A.list_models() and '_test_unsupported_entry' in A.list_models(include_unsupported=True)
        with pytest.warns(UserWarning, match='refuses'):
            A.create_model('_test_unsupported_entry')
    finally:
        registry._unregister('_test_unsupported_entry')
    with pytest.raises(RuntimeError):
        A.create_model('ga_convnext_tiny_768', pretrained=True)
    with pytest.raises(RuntimeError):
        A.create_model('not_a_model')


def test_no_cpu_fallback_is_loud():
    import imagenet_models_amd as A
    m = A.create_model('ga_convnext_tiny_768')
    with pytest.raises(RuntimeError, match='no CPU'):
        m(torch.zeros(1, 3, 224, 224))
    with pytest.raises(RuntimeError, match='GPU'):
        m.flat_state()
    with pytest.raises(RuntimeError):
        A.ga_loss([torch.zeros(2, 10)] * 5, torch.zeros(2, dtype=torch.long), -0.8)
    from imagenet_models_amd import ops
    with pytest.raises(AssertionError):
        ops.Plan(eager=False).gemm(torch.zeros(8, 8), torch.zeros(8, 8), torch.zeros(8, 8), 8, 8, 8, ops.GA_F32)


def _small():
    import imagenet_models_amd as A
    return A.GA_ConvNeXt(num_classes=40, depths=(1, 1, 6, 1, 1), dims=(16, 32, 64, 128, 128), gram_dim=32, dim_embed=64)


def test_flat_layout_and_buckets_partition_the_gradient_buffer():
    from imagenet_models_amd.trainer import TrainStep
    m = _small()
    m._flatten()   # CPU tensors are fine for the layout logic
    st = m.flat_state()
    total = sum(p.numel() for p in m.parameters())
    assert st['total'] == total
    # timm weight-decay rule: decay segment holds exactly the >=2-D non-bias parameters
    n_decay = sum(p.numel() for n, p in m.named_parameters() if not (p.ndim <= 1 or n.endswith('.bias')))
    assert st['n_decay'] == n_decay
    for n, p in m.named_parameters():
        off, k = st['slices'][n]
        assert p.data_ptr() == st['params'].data_ptr() + 4 * off and p.grad.data_ptr() == st['grads'].data_ptr() + 4 * off
        assert (off < n_decay) == (not m.no_weight_decay_param(n, p))
    from imagenet_models_amd.trainer import make_buckets
    for cap in (8 << 20, 50000):     # one bucket per run / runs cut into <= 50k-element all-reduces
        buckets = make_buckets(st, m.grad_groups(), cap)
        cover = sorted((a, b) for _, a, b in buckets)
        assert cover[0][0] == 0 and cover[-1][1] == total
        for (a0, b0), (a1, b1) in zip(cover, cover[1:]):
            assert b0 == a1                                   # exact partition
        assert all(b - a <= cap for a, b in cover)
        # backward-completion order: the heads' slices first (the end of the decay segment), marks never go backwards
        assert buckets[0][0] == 'heads' and any(mk == 'heads' and b == n_decay for mk, a, b in buckets)
        marks = [mk for mk, _, _ in buckets]
        order = [g for g, _ in m.grad_groups()] + ['end']
        assert [order.index(mk) for mk in marks] == sorted(order.index(mk) for mk in marks)
        owner = {n: next((g for g, pre in m.grad_groups() if n.startswith(tuple(pre))), 'end') for n in st['slices']}
        for n, (off, k) in st['slices'].items():
            assert all(mk == owner[n] for mk, a, b in buckets if a < off + k and off < b), n
    k_off = st['slices']['ga.0.attn.k.weight'][0]
    v_off = st['slices']['ga.0.attn.v.weight'][0]
    assert v_off == k_off + m.ga[0].attn.k.weight.numel()   # engine relies on k/v adjacency


def _gloo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from imagenet_models_amd.trainer import make_buckets
    torch.manual_seed(0)
    m = _small()
    m._flatten()
    st = m.flat_state()
    buckets = make_buckets(st, m.grad_groups(), 60000)
    g = torch.Generator().manual_seed(100 + rank)
    local = torch.randn(st['total'], generator=g)
    st['grads'].copy_(local / world)          # TrainStep folds 1/world into the loss gradient scale
    works = [dist.all_reduce(st['grads'][a:b], async_op=True) for _, a, b in buckets]
    for w in works:
        w.wait()
    want = sum(torch.randn(st['total'], generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
    ok = torch.allclose(st['grads'], want, atol=1e-6)
    # parameter views see the reduced gradient
    p = dict(m.named_parameters())['fc.4.weight']
    off, k = st['slices']['fc.4.weight']
    ok = ok and torch.equal(p.grad.reshape(-1), st['grads'][off:off + k])
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_cosine_schedule_matches_timm_formula():
    import math
    from imagenet_models_amd.optim import CosineLRScheduler

    class _Opt:
        param_groups = [dict(lr=5e-3, initial_lr=5e-3)]
    opt = _Opt()
    sch = CosineLRScheduler(opt, t_initial=300, lr_min=1e-5, warmup_t=3, warmup_lr_init=1e-6)
    assert opt.param_groups[0]['lr'] == 1e-6
    sch.step(1)
    assert abs(opt.param_groups[0]['lr'] - (1e-6 + (5e-3 - 1e-6) / 3)) < 1e-12
    sch.step(150)
    assert abs(opt.param_groups[0]['lr'] - (1e-5 + 0.5 * (5e-3 - 1e-5) * (1 + math.cos(math.pi * 150 / 300)))) < 1e-12


def test_drop_path_schedule_matches_reference():
    # ga_convnext.py:362: linspace over sum(depths) INCLUDING the trailing 1; gram layers take the last point
    from oracle import ga_convnext_oracle as O
    import imagenet_models_amd as A
    from imagenet_models_amd.engine import GAEngine, tap_indices
    assert tap_indices(9, 2) == O.tap_indices(9, 2) == [2, 5]
    assert tap_indices(27, 4) == O.tap_indices(27, 4) == [4, 9, 14, 19]
    m = A.create_model('ga_convnext_tiny_768', drop_path_rate=0.2)
    eng = GAEngine.__new__(GAEngine)
    eng.cfg = m.cfg
    rates = GAEngine._drop_path_rates(eng)
    ref = O.drop_path_rates(O.make_cfg('ga_convnext_tiny_768', drop_path_rate=0.2))
    assert abs(rates['stages.2.blocks.4.'] - ref[2][4]) < 1e-7 and abs(rates['stages.3.blocks.2.'] - ref[3][2]) < 1e-7
    assert abs(rates['gram_layer.3.blocks.0.'] - ref[4][0]) < 1e-7 and rates['stages.4.'] == pytest.approx(0.2)


def test_load_timm_layout_checkpoint_with_args_namespace(tmp_path):
    """timm CheckpointSaver files (GA/train.py:649-651: args=args) hold an argparse.Namespace, 'module.'-prefixed keys when
    saved from DDP, and state_dict_ema: load_checkpoint must read them under weights_only=True"""
    import argparse
    from imagenet_models_amd.checkpoint import load_checkpoint
    m = _small()
    sd = {k: torch.full_like(v, 0.25) if v.is_floating_point() else v.clone() for k, v in m.state_dict().items()}
    ema = {k: torch.full_like(v, 0.5) if v.is_floating_point() else v.clone() for k, v in m.state_dict().items()}
    path = os.path.join(tmp_path, 'checkpoint-3.pth.tar')
    torch.save({'epoch': 3, 'arch': 'ga_convnext_tiny_768', 'state_dict': {'module.' + k: v for k, v in sd.items()},
                'optimizer': {'state': {}, 'param_groups': [{'lr': 0.1}]}, 'version': 2,
                'args': argparse.Namespace(model='ga_convnext_tiny_768', lr=5e-3, opt='lamb'), 'amp_scaler': {'scale': 65536.0},
                'state_dict_ema': ema, 'metric': 81.2}, path)
    load_checkpoint(m, path)
    assert float(m.fc[0].weight.mean()) == 0.25
    load_checkpoint(m, path, use_ema=True)
    assert float(m.fc[0].weight.mean()) == 0.5
    with pytest.raises(RuntimeError, match='size mismatch'):      # a wrong-shaped tensor is rejected, not broadcast
        bad = dict(sd)
        bad['fc.0.bias'] = torch.zeros(1)
        m2 = _small()
        m2._flatten()
        m2.load_state_dict(bad)


def test_second_cuda_call_keeps_flat_buffers_and_stale_holders_raise():
    """model.cuda()/.to() after the optimizer exists must not silently re-create the flat buffers"""
    m = _small()
    m._flatten()
    st = m.flat_state()
    gen = st['gen']
    m.check_flat_generation(gen, 'test')
    m._flatten()                                  # what a parameter move does
    with pytest.raises(RuntimeError, match='re-created'):
        m.check_flat_generation(gen, 'test')


def test_asm_load_audit_flags_a_touched_destination(tmp_path):
    """tools/asm_load_audit.py (run by the Makefile on every rebuild of gemm.hip): an instruction that touches the destination
    registers of an inline-asm buffer load before the s_waitcnt that retires it is a finding; the same code with the wait first is not."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'asm_load_audit.py')
    bad = """_Z9my_kernelv:                           ; @_Z9my_kernelv
\tbuffer_load_dwordx4 v[4:7], v1, s[8:11], 0 offen
\tv_add_u32_e32 v5, v2, v3
\ts_waitcnt vmcnt(0)
\ts_endpgm
.Lfunc_end0:
"""
    good = bad.replace('\tv_add_u32_e32 v5, v2, v3\n\ts_waitcnt vmcnt(0)\n', '\ts_waitcnt vmcnt(0)\n\tv_add_u32_e32 v5, v2, v3\n')
    lds = bad.replace('buffer_load_dwordx4 v[4:7], v1, s[8:11], 0 offen', 'buffer_load_dwordx4 v1, s[8:11], 0 offen lds')
    for text, rc in ((bad, 1), (good, 0), (lds, 0)):
        f = tmp_path / 'k.s'
        f.write_text(text)
        r = subprocess.run([sys.executable, tool, str(f), 'my_kernel'], capture_output=True, text=True)
        assert r.returncode == rc, (text, r.stdout, r.stderr)


def test_asm_store_audit_flags_a_write_behind_a_16_byte_buffer_store(tmp_path):
    """tools/asm_load_audit.py --stores (run by the Makefile on every rebuild of dwconv.hip): a VALU write of a 16-byte buffer
    store's data registers inside the 2 wait states behind the store is a finding (hipcc does not pad this pair when
    the store has an SGPR offset); behind `s_nop 1`, or on other registers, it is not."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'asm_load_audit.py')
    bad = """_Z9my_kernelv:                           ; @_Z9my_kernelv
\tbuffer_store_dwordx4 v[4:7], v1, s[8:11], s2 offen
\tv_pk_mul_f32 v[6:7], s[34:35], v[16:17]
\ts_endpgm
.Lfunc_end0:
"""
    padded = bad.replace('\tv_pk_mul_f32', '\ts_nop 1\n\tv_pk_mul_f32')
    other = bad.replace('v_pk_mul_f32 v[6:7]', 'v_pk_mul_f32 v[8:9]')
    for text, rc in ((bad, 1), (padded, 0), (other, 0)):
        f = tmp_path / 'k.s'
        f.write_text(text)
        r = subprocess.run([sys.executable, tool, '--stores', str(f), 'my_kernel'], capture_output=True, text=True)
        assert r.returncode == rc, (text, r.stdout, r.stderr)
