"""Closed-form parameters for module pins: the reference module (in tests/golden/make_golden_r2.py) and this build's
module (in the tests) are filled by the SAME function of (position in the state dict, element index), so a fixture only
has to hold inputs and outputs, not a 12 MB state dict.  Also proves that the state-dict keys and shapes agree: the fill
walks them in order."""
import math

import torch


def fill_parameters(module):
    with torch.no_grad():
        for k, (name, p) in enumerate(module.state_dict().items()):
            if not torch.is_floating_point(p):
                continue
            n = p.numel()
            fan_in = n // p.shape[0] if p.dim() > 1 else 1
            bound = 1.0 / math.sqrt(max(fan_in, 1)) if p.dim() > 1 else 0.1
            i = torch.arange(n, dtype=torch.float64)
            v = ((i * 0.6180339887498949 + 0.37 * (k + 1)) % 1.0 - 0.5) * 2.0 * bound
            p.copy_(v.reshape(p.shape).to(p.dtype))
    return module
