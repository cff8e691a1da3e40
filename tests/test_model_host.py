"""CPU tier: host-side behaviour of the mirrored modules that needs no kernel -- construction, initialisation, refusal to
compute on the CPU (the product path has no fallback)."""
import numpy as np
import pytest
import torch

import squeezedet_pytorch_amd as sqd


@pytest.mark.parametrize("arch", ["squeezedet", "squeezedetplus"])
def test_init_weights_distributions(arch):
    """SURVEY 8a row H, src/model/squeezedet.py:89-97: every conv weight ~ N(0, 0.005) except ConvDet ~ N(0, 0.002); every bias 0."""
    from squeezedet_pytorch_amd.model import SqueezeDetBase
    torch.manual_seed(0)
    base = SqueezeDetBase(sqd.make_cfg(arch=arch, device='cpu'))
    n_w = 0
    for name, p in base.named_parameters():
        if name.endswith('.bias'):
            assert float(p.detach().abs().max()) == 0.0, name
            continue
        std = 0.002 if name.startswith('convdet') else 0.005
        v = p.detach().double().reshape(-1)
        n = v.numel()
        assert abs(float(v.mean())) <= 5 * std / np.sqrt(n), name                        # 5 sigma of the sample mean
        assert abs(float(v.std()) - std) <= 5 * std / np.sqrt(2 * n) + 1e-9, (name, float(v.std()))   # 5 sigma of the sample std
        n_w += 1
    assert n_w == 32
    # re-initialising draws again with the same distributions (the reference calls init_weights() in __init__ only; callable)
    w0 = base.convdet.weight.detach().clone()
    base.init_weights()
    assert not torch.equal(w0, base.convdet.weight) and abs(float(base.convdet.weight.std()) - 0.002) < 2e-4


def test_modules_refuse_cpu_input():
    from squeezedet_pytorch_amd.model import SqueezeDet
    m = SqueezeDet(sqd.make_cfg(input_size=(64, 96), device='cpu'))
    with pytest.raises(RuntimeError, match='HIP'):
        m.base(torch.zeros(1, 3, 64, 96))
