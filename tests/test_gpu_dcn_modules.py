"""GPU parity of the HIP DSTA / MVDualAttAlignment modules (reference class surfaces) against golden vectors from the
real reference classes (DCN step = C oracle) on the same seeded weights and inputs."""
import pytest
import torch

from test_oracle_dcn_modules import GOLD, load_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", GOLD, ids=lambda p: p.split("/")[-1][:-4])
def test_hip_modules_match_reference_golden(path):
    kind, sd, inputs, gold = load_case(path)
    if kind == "dsta":
        from ops.attentionlayer import DSTA
        m = DSTA(64)
    else:
        from cdfo_amd.mv_align import MVDualAttAlignment
        m = MVDualAttAlignment(64, 64, 3, padding=1, deformable_groups=16, max_residue_magnitude=10)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        out = m(*[t.cuda() for t in inputs])
    torch.cuda.synchronize()
    assert out.shape == gold.shape
    err = (out.cpu() - gold).abs().max().item()
    assert err <= 1e-3 * max(1.0, gold.abs().max().item()), err
    print(kind, "max-abs", err)


def test_modules_reject_cpu():
    from ops.attentionlayer import DSTA
    with torch.no_grad(), pytest.raises(NotImplementedError):
        DSTA(64)(torch.zeros(1, 64, 40, 40))
