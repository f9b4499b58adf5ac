"""Dropout-mask source.  Product code asks `provider.keep(name, shape, p, device)` for uint8 keep flags;
tests inject the reference's recorded draws by name so that HIP and oracle see identical masks."""
import torch


class MaskProvider:
    def __init__(self):
        self.injected = None        # dict name -> uint8 tensor (already in the layout the caller asks for)
        self.generator = None

    def keep(self, name, shape, p, device):
        if self.injected is not None:
            if name not in self.injected:
                raise KeyError(f'no injected dropout mask named {name!r}')
            m = self.injected[name]
            assert tuple(m.shape) == tuple(shape), (name, tuple(m.shape), tuple(shape))
            return m.to(device=device, dtype=torch.uint8).contiguous()
        if p <= 0.0:
            return None
        return (torch.rand(shape, device=device, generator=self.generator) >= p).to(torch.uint8)

    def teacher(self, T, ratio):
        """`torch.rand([T]) > 1 - ratio` of reference modules/tacotron2.py:171 (host side)."""
        if self.injected is not None and 'teacher' in self.injected:
            return [bool(x) for x in self.injected['teacher']]
        return [bool(x) for x in (torch.rand([T]) > (1 - ratio))]


provider = MaskProvider()
