"""Every registered entry point runs on the engine: one fused train step (TrainStep: forward, loss, backward, optimizer) and one
eval forward at a small batch, finite loss, finite logits of the right shape -- catches shape assumptions the parity tests of the
tiny / narrow configurations do not reach (depth-27 trunks with four taps, the 24322 CSWin, map_convnext_small)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _names():
    import imagenet_models_amd as A
    return A.list_models()


@pytest.mark.parametrize('name', ['ga_convnext_small_768', 'ga_convnext_small_688', 'ga_convnext_base_1024', 'ga_convnext_base_976',
                                  'ga_CSWin_64_24322_small_224', 'map_convnext_small', 'ga_convnext_tiny', 'ga_convnext_small',
                                  'ga_convnext_base', 'map_vit_small_patch16_224', 'map_vit_base_patch16_224', 'map_pit_s', 'convnext_tiny'])
def test_entry_point_trains_and_evaluates(name):
    import imagenet_models_amd as A
    assert name in _names()
    torch.manual_seed(0)
    m = A.create_model(name, pretrained=False, drop_path_rate=0.1).cuda().train()
    opt = A.create_optimizer_v2(m, opt='adamw', lr=1e-4, weight_decay=0.05)
    step = A.TrainStep(m, opt, 4, lam=-0.8)
    img = m.cfg.get('img_size', 224)
    x = torch.randn(4, 3, img, img, device='cuda')
    y = torch.randint(0, 1000, (4,), device='cuda')
    loss = step(x, y)
    loss = step(x, y)
    torch.cuda.synchronize()
    assert torch.isfinite(loss).all(), name
    m.eval()
    with torch.no_grad():
        outs = m(x)
    if isinstance(outs, torch.Tensor):          # the plain ConvNeXt returns one tensor, as the reference class does
        outs = [outs]
    assert all(o.shape == (4, 1000) and torch.isfinite(o).all() for o in outs), name
    assert all(torch.isfinite(p).all() for p in m.parameters()), name


def test_every_listed_model_is_covered_somewhere():
    """list_models() holds exactly the factories of the three families (ga_convnext.py:572-613 + README aliases, the two CSWin
    candidates, map_convnext.py:173-240, map_pit.py:224-251, the builder-defined MAP-ViT compositions)"""
    assert set(_names()) == {'ga_convnext_tiny_688', 'ga_convnext_tiny_768', 'ga_convnext_small_688', 'ga_convnext_small_768',
                             'ga_convnext_base_976', 'ga_convnext_base_1024', 'ga_convnext_tiny', 'ga_convnext_small',
                             'ga_convnext_base', 'ga_CSWin_64_12211_tiny_224', 'ga_CSWin_64_24322_small_224', 'map_convnext_tiny',
                             'map_convnext_small', 'map_vit_base_patch16_384', 'map_vit_base_patch16_224', 'map_vit_small_patch16_224',
                             'map_pit_s', 'convnext_tiny', 'convnext_small'}
