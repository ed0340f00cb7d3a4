"""Where round 1's parity was thinnest ("What's weak"):

  * fp16 range of the default `fp16x2` mode: every earlier case used the Kaiming-0.1 random init (activations O(1)).  Here the
    same function is re-parametrised so that the trunk's activations are 2^10, 2^16 or 2^-14 times larger -- the network
    output, and therefore the reference, is unchanged (all scalings are powers of two and the trunk between `tsa_fusion`
    and `upconv1` is positively homogeneous once its biases are scaled too).  2^10 must pass in fp16 as is (relative
    precision does not depend on the scale inside the normal range); 2^16 overflows fp16 and 2^-14 sinks into its
    subnormals: the range guard must detect both and repeat the forward in split-bf16, still <= 1e-3 against the oracle.
  * the default noise path (uniforms drawn inside the mask kernel): statistics of the draws, reproducibility under
    torch.manual_seed, and parity against the oracle fed with the captured draws.
  * the DCN fast path (split-fp16 operands) at extreme magnitudes: tiny / huge inputs (power-of-two pre-scale), a mask far
    outside [0,1] and an infinite input (device-side overflow flag -> exact-fp32 re-run)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _scaled_state(sd, s):
    """The same function with the trunk's activations multiplied by s (a power of two)."""
    out = {k: v.clone() for k, v in sd.items()}
    out["tsa_fusion.weight"] *= s
    out["tsa_fusion.bias"] *= s
    for k in out:
        if k.startswith("recon_trunk.") and k.endswith(".bias"):
            out[k] *= s
    out["upconv1.weight"] /= s
    return out


def _forward(sd, inp, precision="fp16x2"):
    from arch.SIDECVSR_our import CVSR_V8
    m = CVSR_V8()
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    m.precision = precision
    d = {k: v.cuda() for k, v in inp.items() if k != "gumbel_u"}
    with torch.no_grad():
        out, L1 = m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=[u.cuda() for u in inp["gumbel_u"]])
    torch.cuda.synchronize()
    return m, out.cpu(), L1.cpu()


@pytest.mark.parametrize("log2s", [0, 8, 16, -14])
def test_fp16_range_guard(log2s):
    from arch.SIDECVSR_our import CVSR_V8
    from oracle.cvsr_v8_ref import cvsr_v8_forward, make_inputs, make_state_dict
    sd = make_state_dict(3)
    inp = make_inputs(1, 16, 24, 77)
    taps = {}
    with torch.no_grad():
        ref, L1_ref = cvsr_v8_forward(sd, inp["x"], None, inp["mvs1"], inp["pms"], inp["rms"], inp["ufs"], None, inp["gumbel_u"], taps)
    s = 2.0 ** log2s
    amax = taps["fused"].abs().max().item() * s              # what the guard's probe will see at the trunk's input
    lo, hi = CVSR_V8.FP16_WINDOW
    expect_fallback = not (lo <= amax <= hi)
    assert expect_fallback == (log2s in (16, -14)), f"scenario drifted: max |trunk input| = {amax}"
    with pytest.warns(UserWarning, match="fp16 range") if expect_fallback else _nullcontext():
        m, out, L1 = _forward(_scaled_state(sd, s), inp)
    err, err_l1 = (out - ref).abs().max().item(), (L1 - L1_ref).abs().max().item()
    print(f"trunk activations x 2^{log2s}: range guard {m.last_range}; out {err:.2e} L1_fea {err_l1:.2e}")
    assert m.last_range["fallback"] == expect_fallback
    assert err <= TOL and err_l1 <= TOL
    if expect_fallback:        # without the guard the same weights are outside the bound (or not finite): the guard is what saves it
        m.range_guard = False
        d = {k: v.cuda() for k, v in inp.items() if k != "gumbel_u"}
        with torch.no_grad():
            raw, _ = m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=[u.cuda() for u in inp["gumbel_u"]])
        bad = (raw.cpu() - ref).abs().max().item()
        print(f"   unguarded fp16x2: {bad:.2e}")
        assert not (bad <= 1e-5)       # NaN / inf / visibly wrong: the scaling really leaves fp16's range


def test_fp16_range_guard_deferred_result_check():
    """Round 5: the result's non-finite probe is read at the start of the NEXT forward (or by finish_range_guard / last_range), not
    behind a host sync at the end of this one.  Scenario the trunk-input window cannot see: ONE Block_ body re-parametrised by 2^24
    (body.0 weight and bias x 2^24, body.2 weight x 2^-24: the same function, LeakyReLU is positively homogeneous) -- its fp16
    weights and 256-channel fp16 intermediate overflow while max |trunk input| stays in the window.  The forward returns garbage (the trunk's output is non-finite; the up-sampler's tail turns that
    into finite values, which is why the guard probes the trunk and not the image); settling the guard must warn, recompute in bf16x3 INTO the returned tensors and report the fallback; the next forwards are guarded the same way."""
    from arch.SIDECVSR_our import CVSR_V8
    from oracle.cvsr_v8_ref import cvsr_v8_forward, make_inputs, make_state_dict
    sd = make_state_dict(3)
    inp = make_inputs(1, 16, 24, 78)
    with torch.no_grad():
        ref, L1_ref = cvsr_v8_forward(sd, inp["x"], None, inp["mvs1"], inp["pms"], inp["rms"], inp["ufs"], None, inp["gumbel_u"])
    bad = {k: v.clone() for k, v in sd.items()}
    p = "recon_trunk.body.2.body.1.body."
    bad[p + "0.weight"] *= 2.0 ** 24
    bad[p + "0.bias"] *= 2.0 ** 24
    bad[p + "2.weight"] *= 2.0 ** -24
    m = CVSR_V8()
    m.load_state_dict(bad, strict=True)
    m = m.cuda().eval()
    d = {k: v.cuda() for k, v in inp.items() if k != "gumbel_u"}
    noise = [u.cuda() for u in inp["gumbel_u"]]
    with torch.no_grad():
        out, L1 = m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=noise)
    assert m.__dict__.get("_pending_guard") is not None                     # the trunk-input window saw nothing: the result check is pending
    with pytest.warns(UserWarning, match="fp16 range"):
        m.finish_range_guard()
    torch.cuda.synchronize()
    assert m.last_range["fallback"] and m.last_range["nonfinite"]
    err, err_l1 = (out.cpu() - ref).abs().max().item(), (L1.cpu() - L1_ref).abs().max().item()
    print(f"deferred guard: repaired in place, out {err:.2e} L1_fea {err_l1:.2e}; {m.last_range}")
    assert err <= TOL and err_l1 <= TOL
    # the second forward settles nothing (no pending check) and is itself caught when ITS check is settled through last_range
    with torch.no_grad():
        out2, _ = m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=noise)
    assert m.last_range["fallback"]
    assert (out2.cpu() - ref).abs().max().item() <= TOL
    # range_guard = "sync": the check-before-return behaviour of rounds 2-4
    m.range_guard = "sync"
    with torch.no_grad():
        out3, _ = m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=noise)
    assert m.__dict__.get("_pending_guard") is None and (out3.cpu() - ref).abs().max().item() <= TOL


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def test_default_noise_path_statistics_seeding_and_parity():
    from arch.SIDECVSR_our import CVSR_V8
    from oracle.cvsr_v8_ref import cvsr_v8_forward, make_inputs, make_state_dict
    sd = make_state_dict(4)
    m = CVSR_V8()
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    inp = make_inputs(2, 24, 40, 91)
    d = {k: v.cuda() for k, v in inp.items() if k != "gumbel_u"}

    def run(seed):
        torch.manual_seed(seed)
        m.capture_noise = []
        with torch.no_grad():
            out, L1 = m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"])      # no noise given: drawn in the kernel
        torch.cuda.synchronize()
        cap, m.capture_noise = m.capture_noise, None
        return out.cpu(), L1.cpu(), [u.cpu() for u in cap]

    out, L1, u = run(11)
    assert len(u) == 6 and all(t.shape == (2, 64, 24, 40) for t in u)
    allu = torch.stack(u).double()
    assert allu.min().item() > 0.0 and allu.max().item() < 1.0                      # never 0 (arch.py:2170), never 1
    n = allu.numel()
    assert abs(allu.mean().item() - 0.5) < 4 * (1 / 12) ** 0.5 / n ** 0.5           # uniform mean, 4 sigma
    assert abs(allu.var().item() - 1 / 12) < 0.002
    hist = torch.histc(allu.float(), bins=16, min=0, max=1) / n
    assert (hist - 1 / 16).abs().max().item() < 0.003
    flat = allu.view(6, -1)
    for i in range(6):                                                              # the six draws are different streams
        for j in range(i + 1, 6):
            assert abs(torch.corrcoef(torch.stack([flat[i], flat[j]]))[0, 1].item()) < 0.02
    # neighbouring pixels / channels / images are uncorrelated
    a = allu[0]
    assert abs(torch.corrcoef(torch.stack([a[:, :, :, 1:].reshape(-1), a[:, :, :, :-1].reshape(-1)]))[0, 1].item()) < 0.01
    assert abs(torch.corrcoef(torch.stack([a[:, 1:].reshape(-1), a[:, :-1].reshape(-1)]))[0, 1].item()) < 0.01
    assert abs(torch.corrcoef(torch.stack([a[1:].reshape(-1), a[:-1].reshape(-1)]))[0, 1].item()) < 0.01
    # parity of the default path: the oracle fed with exactly the uniforms the kernels drew
    with torch.no_grad():
        ref, L1_ref = cvsr_v8_forward(sd, inp["x"], None, inp["mvs1"], inp["pms"], inp["rms"], inp["ufs"], None, u)
    err = (out - ref).abs().max().item()
    print(f"default noise path vs oracle on the captured draws: out {err:.2e}")
    assert err <= TOL and (L1 - L1_ref).abs().max().item() <= TOL
    # reproducible under torch.manual_seed, fresh draws otherwise
    out2, _, u2 = run(11)
    assert torch.equal(out, out2) and all(torch.equal(x, y) for x, y in zip(u, u2))
    out3, _, u3 = run(12)
    assert not torch.equal(u[0], u3[0])
    torch.manual_seed(11)
    with torch.no_grad():
        m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"])
        m.capture_noise = []
        m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"])                 # second call after the same seed: new draws
    assert not torch.equal(m.capture_noise[0].cpu(), u[0])
    m.capture_noise = None


@pytest.mark.parametrize("case", ["tiny", "huge", "mask_1e6", "inf_input", "tiny_weights"])
def test_dcn_fast_path_is_range_safe(case):
    from cdfo_amd.dcn import modulated_deform_conv
    from oracle.dcn_modules_ref import dcn_forward_ref
    rs = np.random.RandomState(5)
    B, C, Co, H, W, dg = 2, 64, 64, 20, 28, 16
    x = rs.standard_normal((B, C, H, W)).astype(np.float32)
    w = (rs.standard_normal((Co, C, 3, 3)) / 24).astype(np.float32)
    b = rs.standard_normal((Co,)).astype(np.float32)
    off = (rs.standard_normal((B, 2 * dg * 9, H, W)) * 2).astype(np.float32)
    msk = rs.uniform(0, 1, (B, dg * 9, H, W)).astype(np.float32)
    if case == "tiny":
        x *= np.float32(2.0 ** -40)
        b *= np.float32(2.0 ** -40)
    elif case == "huge":
        x *= np.float32(2.0 ** 30)
    elif case == "mask_1e6":
        msk *= np.float32(1e6)
    elif case == "inf_input":
        x[0, 3, 5, 7] = np.inf
    elif case == "tiny_weights":
        w *= np.float32(2.0 ** -126)
        b[:] = 0
    t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    with torch.no_grad():
        got = modulated_deform_conv(t(x), t(off), t(msk), t(w), t(b), 1, 1, 1, 1, dg).cpu().numpy()
    with np.errstate(all="ignore"):
        want = dcn_forward_ref(x, off, msk, w, b, 1, 1, 1, 1, dg)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin)
    scale = max(float(np.abs(want[fin]).max()), 1e-30)
    err = float(np.abs(got[fin] - want[fin]).max()) / scale
    print(f"dcn fast path, {case}: max |err| / max |out| = {err:.2e} (max |out| {scale:.3g})")
    assert err <= 2e-5
