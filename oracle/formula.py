"""Formula-generated weights and clips (test infrastructure; SURVEY.md 8c golden protocol).

Everything is an exact integer hash -> no libm dependence, identical on every host, so the
golden generator (run once next to the reference) and the tests (run anywhere) see the
same bits without shipping weight files.
"""
import zlib

import numpy as np
import torch

_M32 = np.uint64(0xFFFFFFFF)


def _hash_u32(idx, salt):
    """Elementwise avalanche hash of uint64 indices -> uint32 (as uint64 array)."""
    x = (idx * np.uint64(2654435761) + np.uint64(salt)) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x45D9F3B)) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x45D9F3B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def unit(name, numel, lane=0):
    """float64 values in [-1, 1) determined by (name, element index)."""
    salt = zlib.crc32(name.encode()) + 7919 * lane
    h = _hash_u32(np.arange(numel, dtype=np.uint64), salt)
    return h.astype(np.float64) / 2147483648.0 - 1.0


def fill_state_dict(module):
    """Load closed-form values into every parameter/buffer of `module` (by state_dict key)."""
    sd = module.state_dict()
    new = {}
    for key, t in sd.items():
        n = t.numel()
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            new[key] = torch.zeros_like(t)
            continue
        u = unit(key, n)
        if leaf == "running_var":
            v = 0.6 + 0.4 * np.abs(u)
        elif leaf == "running_mean":
            v = 0.1 * u
        elif t.dim() == 1 and leaf == "weight":      # norm gamma
            v = 1.0 + 0.1 * u
        elif leaf == "bias":
            v = 0.1 * u
        else:                                        # conv / linear weight
            fan_in = max(1, n // t.shape[0])
            v = u * np.sqrt(3.0 / fan_in)
        new[key] = torch.from_numpy(v.reshape(tuple(t.shape))).to(t.dtype)
    module.load_state_dict(new)
    return module


def clip_u8(n, t, h, w, salt=1234):
    """uint8 clip (N,T,H,W,3) as the reference's dataloader hands it over (dataloader.py:17-21)."""
    numel = n * t * h * w * 3
    hsh = _hash_u32(np.arange(numel, dtype=np.uint64), salt)
    return torch.from_numpy((hsh >> np.uint64(9) & np.uint64(0xFF)).astype(np.uint8).reshape(n, t, h, w, 3))


def clip_features(n, t, h, w, dtype=torch.float32, salt=1234):
    """train.py:102-108 transform: (N,T,H,W,3) uint8 -> logical (N,3,T,H,W) /255, physically NDHWC."""
    return clip_u8(n, t, h, w, salt).permute(0, -1, 1, 2, 3).to(dtype) / 255


def labels(n, num_classes, salt=99):
    hsh = _hash_u32(np.arange(n, dtype=np.uint64), salt)
    return torch.from_numpy((hsh % np.uint64(num_classes)).astype(np.int64))


def tensor(name, shape, scale=1.0, dtype=torch.float32):
    """Generic formula tensor in [-scale, scale)."""
    n = int(np.prod(shape))
    return torch.from_numpy((unit(name, n) * scale).reshape(shape)).to(dtype)
