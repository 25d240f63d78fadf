"""Checkpoint I/O in timm's CheckpointSaver layout ({'epoch','arch','state_dict','optimizer','version':2,'metric'},
GA/train.py:649-651,693) so reference `.pth.tar` files load by key.  Loading uses weights_only=True."""
import torch


class ModelEma:
    """timm ModelEmaV2 (GA/train.py:497-502, 774-775): ema = decay * ema + (1 - decay) * model over every state_dict
    value, on the device.  The parameters live in ONE flat fp32 buffer, so their update is a single kernel launch
    (ga_lerp_f32); BatchNorm running statistics are few small tensors, num_batches_tracked is copied."""

    def __init__(self, model, decay=0.9998):
        from .ops import Plan
        self.model, self.decay = model, decay
        st = model.flat_state()
        self.gen = st['gen']
        self.flat = st['params'].clone()
        self.slices = st['slices']
        self.buffers = {n: b.detach().clone() for n, b in model.named_buffers()}
        self.plan = Plan(name='ema')
        self.plan.lerp_f32(self.flat, st['params'], 1.0 - decay, st['total'])
        for n, b in model.named_buffers():
            if b.dtype == torch.float32 and b.numel() > 0:
                self.plan.lerp_f32(self.buffers[n], b, 1.0 - decay, b.numel())

    def update(self, model=None):
        self.model.check_flat_generation(self.gen, 'ModelEma')
        self.plan.run()
        for n, b in self.model.named_buffers():
            if b.dtype != torch.float32:
                self.buffers[n].copy_(b)

    def state_dict(self):
        ref = self.model.state_dict()
        out = {}
        for k, v in ref.items():
            if k in self.slices:
                off, n = self.slices[k]
                out[k] = self.flat[off:off + n].view(v.shape)
            else:
                out[k] = self.buffers[k]
        return out


def save_checkpoint(model, optimizer, epoch, path, metric=None, arch='', model_ema=None):
    ck = {'epoch': epoch, 'arch': arch, 'state_dict': {k: v.detach().cpu() for k, v in model.state_dict().items()},
          'optimizer': {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in optimizer.state_dict().items()}
          if optimizer is not None else None, 'version': 2, 'metric': metric}
    if model_ema is not None:      # timm CheckpointSaver key
        ck['state_dict_ema'] = {k: v.detach().cpu() for k, v in model_ema.state_dict().items()}
    torch.save(ck, path)


def load_checkpoint(model, path, strict=True, use_ema=False):
    """timm-layout checkpoints (`CheckpointSaver` passes args=args, GA/train.py:649-651) carry an argparse.Namespace under
    'args': allow-listed as plain data, everything else stays under weights_only=True (nothing from the file executes)"""
    import argparse
    with torch.serialization.safe_globals([argparse.Namespace]):
        ck = torch.load(path, map_location='cpu', weights_only=True)
    if use_ema and isinstance(ck, dict) and 'state_dict_ema' in ck:
        ck = {'state_dict': ck['state_dict_ema']}
    sd = ck.get('state_dict', ck.get('model', ck)) if isinstance(ck, dict) else ck
    sd = {k[7:] if k.startswith('module.') else k: v for k, v in sd.items()}
    return model.load_state_dict(sd, strict=strict)
