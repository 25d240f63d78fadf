"""Checkpoint I/O in timm's CheckpointSaver layout ({'epoch','arch','state_dict','optimizer','version':2,'metric'},
GA/train.py:649-651,693) so reference `.pth.tar` files load by key.  Loading uses weights_only=True."""
import torch


def save_checkpoint(model, optimizer, epoch, path, metric=None, arch=''):
    torch.save({'epoch': epoch, 'arch': arch, 'state_dict': {k: v.detach().cpu() for k, v in model.state_dict().items()},
                'optimizer': {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in optimizer.state_dict().items()}
                if optimizer is not None else None, 'version': 2, 'metric': metric}, path)


def load_checkpoint(model, path, strict=True):
    ck = torch.load(path, map_location='cpu', weights_only=True)
    sd = ck.get('state_dict', ck.get('model', ck)) if isinstance(ck, dict) else ck
    sd = {k[7:] if k.startswith('module.') else k: v for k, v in sd.items()}
    return model.load_state_dict(sd, strict=strict)
