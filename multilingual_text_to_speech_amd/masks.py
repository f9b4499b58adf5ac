"""Dropout-mask source.  Product code asks `provider.keep(name, shape, p, device)` for uint8 keep flags;
tests inject the reference's recorded draws by name so that HIP and oracle see identical masks.

On the GPU the flags come from the library's Philox kernel (mtts_dropout_keep_mask: 1 byte written per element instead of
torch.rand + compare + cast).  Every call draws its 64-bit Philox key from torch's default CPU generator, so
torch.manual_seed() reproduces a run exactly like it does for torch.rand; the data-parallel rank is mixed in."""
import ctypes
import os

import torch


class MaskProvider:
    def __init__(self):
        self.injected = None        # dict name -> uint8 tensor (already in the layout the caller asks for)
        self.generator = None       # a torch.Generator forces the torch.rand path (reproducible against torch streams)

    def _philox(self, n):
        key = int(torch.randint(0, 2 ** 62, (1,)).item())
        return (key + 0x9E3779B97F4A7C15 * int(os.environ.get('RANK', '0'))) & 0xFFFFFFFFFFFFFFFF, 0

    def keep(self, name, shape, p, device):
        if self.injected is not None:
            if name not in self.injected:
                raise KeyError(f'no injected dropout mask named {name!r}')
            m = self.injected[name]
            assert tuple(m.shape) == tuple(shape), (name, tuple(m.shape), tuple(shape))
            return m.to(device=device, dtype=torch.uint8).contiguous()
        if p <= 0.0:
            return None
        dev = torch.device(device)
        if self.generator is not None or dev.type != 'cuda':
            return (torch.rand(shape, device=device, generator=self.generator) >= p).to(torch.uint8)
        from ._C import check, lib, ptr, stream_ptr
        out = torch.empty(shape, dtype=torch.uint8, device=dev)
        n = out.numel()
        seed, off = self._philox(n)
        check(lib().mtts_dropout_keep_mask(ptr(out), ctypes.c_long(n), ctypes.c_float(p), ctypes.c_uint64(seed), ctypes.c_uint64(off),
                                           stream_ptr()), 'mtts_dropout_keep_mask')
        return out

    def teacher(self, T, ratio):
        """`torch.rand([T]) > 1 - ratio` of reference modules/tacotron2.py:171 (host side)."""
        if self.injected is not None and 'teacher' in self.injected:
            return [bool(x) for x in self.injected['teacher']]
        return [bool(x) for x in (torch.rand([T]) > (1 - ratio))]


provider = MaskProvider()
