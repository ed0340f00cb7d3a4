"""Parity at BASELINE.json's full sizes, where the CPU oracle is too slow to run on every clip: size-independent
properties of the forward, anchored on the oracle where one clip is affordable.

  c2  [4,7,1,120,240]   clip 0 of the batch against the CPU ORACLE run on that clip alone (clips never interact in the
                        reference forward), the other clips against their own single-clip forwards
  c3  [8,7,1,272,480]   clip 0 of the batch against the CPU ORACLE run on that clip alone (`out` and `L1_fea` <= 1e-3: the
                        headline configuration checked against the reference's restatement, not only against this code base's
                        own exact mode); a 272x960 strip of the c5 frame size the same way;
                        the default fp16x2 arithmetic against the exact-fp32 MFMA mode of the same kernels (which the small
                        golden cases tie to the reference at 1e-6) -- bound 1e-3 on `out` and `L1_fea`; batch independence;
                        run-to-run bit reproducibility; the cached-feature path against the fresh path on the same window
  c5  [1,7,1,544,960]   (one clip per GPU) the whole clip against the CPU ORACLE (`out`, `L1_fea` <= 1e-3), and fp16x2 against
                        split-bf16 (fp32-grade), bound 1e-3
  c3, other weights     clip 0 at 272x480 against the CPU ORACLE for two further weight seeds (perturbed / plain init)
Tolerance 1e-3 max-abs (BASELINE.json north_star)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _setup(B, H, W, seed, wseed=0, perturb=True):
    from arch.SIDECVSR_our import CVSR_V8
    from oracle.cvsr_v8_ref import make_inputs, make_state_dict
    sd = make_state_dict(wseed, perturb=perturb)
    m = CVSR_V8()
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    inp = make_inputs(B, H, W, seed, pad_rows={272: 2, 544: 4}.get(H, 0))      # 270 / 540 rows zero-padded to a multiple of 8
    return m, sd, inp


def _run(m, inp, sl=slice(None), pre=None, precision=None):
    if precision:
        m.precision = precision
    d = {k: v[sl].cuda() for k, v in inp.items() if k != "gumbel_u"}
    noise = [u[sl].cuda() for u in inp["gumbel_u"]]
    with torch.no_grad():
        out, L1 = m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], pre, gumbel_uniform=noise)
    torch.cuda.synchronize()
    return out, L1


def test_c2_batch_against_the_oracle_and_single_clip_forwards():
    from oracle.cvsr_v8_ref import cvsr_v8_forward
    m, sd, inp = _setup(4, 120, 240, 1001)
    out, L1 = _run(m, inp)
    assert out.shape == (4, 1, 480, 960) and L1.shape == (28, 64, 120, 240)
    one = {k: (v[:1] if k != "gumbel_u" else [u[:1] for u in v]) for k, v in inp.items()}
    torch.set_num_threads(max(1, torch.get_num_threads()))
    with torch.no_grad():
        ref, L1_ref = cvsr_v8_forward(sd, one["x"], None, one["mvs1"], one["pms"], one["rms"], one["ufs"], None, one["gumbel_u"])
    assert (out[:1].cpu() - ref).abs().max().item() <= TOL
    assert (L1[:7].cpu() - L1_ref).abs().max().item() <= TOL
    for b in (1, 3):
        ob, Lb = _run(m, inp, slice(b, b + 1))
        assert (out[b:b + 1] - ob).abs().max().item() <= 2e-5            # summation partitions differ with B, nothing else
        assert (L1[7 * b:7 * b + 7] - Lb).abs().max().item() <= 2e-5


def _clip_vs_oracle(m, sd, inp, clip=0):
    from oracle.cvsr_v8_ref import cvsr_v8_forward
    out, L1 = _run(m, inp)
    one = {k: (v[clip:clip + 1] if k != "gumbel_u" else [u[clip:clip + 1] for u in v]) for k, v in inp.items()}
    with torch.no_grad():
        ref, L1_ref = cvsr_v8_forward(sd, one["x"], None, one["mvs1"], one["pms"], one["rms"], one["ufs"], None, one["gumbel_u"])
    e_out = (out[clip:clip + 1].cpu() - ref).abs().max().item()
    e_l1 = (L1[7 * clip:7 * clip + 7].cpu() - L1_ref).abs().max().item()
    return e_out, e_l1, out


def test_c3_clip0_against_the_oracle():
    """The headline configuration (8 clips of 272x480, default fp16x2 arithmetic) against the CPU oracle on clip 0 -- an
    index bug that only shows above the small golden sizes cannot cancel here, as it could between two modes of one code base."""
    m, sd, inp = _setup(8, 272, 480, 1002)
    e_out, e_l1, out = _clip_vs_oracle(m, sd, inp)
    print(f"c3 B=8 clip 0, fp16x2 vs CPU oracle: out {e_out:.2e}  L1_fea {e_l1:.2e}   range guard {m.last_range}")
    assert out.shape == (8, 1, 1088, 1920) and e_out <= TOL and e_l1 <= TOL
    assert m.last_range is not None and not m.last_range["fallback"]        # the default weights stay inside the fp16 window


@pytest.mark.parametrize("wseed,perturb,iseed", [(1, True, 1012), (2, False, 1022)])
def test_c3_clip0_against_the_oracle_other_weight_seeds(wseed, perturb, iseed):
    """The fp16x2 margin over more than one draw of the weights: two further seeds -- one with the
    LayerNorm affines / temperatures / biases perturbed off their initial values, one plain random init like bench.py's -- at
    the c3 frame size on the grouped three-neighbours-per-launch schedule (B = 3), clip 0 against the CPU oracle."""
    m, sd, inp = _setup(3, 272, 480, iseed, wseed=wseed, perturb=perturb)
    e_out, e_l1, out = _clip_vs_oracle(m, sd, inp)
    print(f"c3 B=3 clip 0, weights seed {wseed} perturb={perturb}, fp16x2 vs CPU oracle: out {e_out:.2e}  L1_fea {e_l1:.2e}   "
          f"range guard {m.last_range}")
    assert e_out <= TOL and e_l1 <= TOL
    assert m.last_range is not None and not m.last_range["fallback"]


def test_c5_clip_against_the_oracle():
    """One full c5 clip (540 rows zero-padded to 544 x 960, test_LD_37.py:24-26 semantics) in the default fp16x2 arithmetic
    against the CPU ORACLE: `out` and `L1_fea` <= 1e-3.  The only check in which a 544-row column attention (H x H map per image
    column) and 960-wide row attention of the same frame meet the reference's restatement rather than another mode of this
    code base.  The oracle forward is ~2 minutes of host CPU."""
    m, sd, inp = _setup(1, 544, 960, 1005)
    e_out, e_l1, out = _clip_vs_oracle(m, sd, inp)
    print(f"c5 1x544x960, fp16x2 vs CPU oracle: out {e_out:.2e}  L1_fea {e_l1:.2e}   range guard {m.last_range}")
    assert out.shape == (1, 1, 2176, 3840) and e_out <= TOL and e_l1 <= TOL
    assert (out[..., 2160:, :].abs().max().item()) < 10.0       # the 16 padded output rows exist and are finite (callers crop them)


def test_c5_width_strip_against_the_oracle():
    """A 272x960 strip (the c5 frame's full width: 960-pixel rows through the row attention, half its height) of one clip
    against the CPU oracle; the full 544x960 oracle forward takes minutes on the host."""
    m, sd, inp = _setup(1, 272, 960, 1004)
    e_out, e_l1, out = _clip_vs_oracle(m, sd, inp)
    print(f"272x960 strip, fp16x2 vs CPU oracle: out {e_out:.2e}  L1_fea {e_l1:.2e}")
    assert out.shape == (1, 1, 1088, 3840) and e_out <= TOL and e_l1 <= TOL


def test_c3_modes_agree_batch_independent_reproducible_and_cached_path():
    m, sd, inp = _setup(8, 272, 480, 1002)
    out, L1 = _run(m, inp, precision="fp16x2")
    out2, L12 = _run(m, inp)
    assert torch.equal(out, out2) and torch.equal(L1, L12)                # no atomics on the path
    exact, L1e = _run(m, inp, precision="f32")
    err, err_l1 = (out - exact).abs().max().item(), (L1 - L1e).abs().max().item()
    print(f"c3 fp16x2 vs exact-fp32 kernels: out {err:.2e}  L1_fea {err_l1:.2e}")
    assert err <= TOL and err_l1 <= TOL
    del exact, L1e, out2, L12
    o5, _ = _run(m, inp, slice(5, 6), precision="fp16x2")
    assert (out[5:6] - o5).abs().max().item() <= 2e-5
    # cached-feature path (arch.py:4420-4427): a window shifted by one frame reuses six of the seven feature maps
    sh = {k: (torch.cat([v[:, 1:], v[:, -1:]], 1) if k in ("x", "pms", "mvs0", "mvs1") else v) for k, v in inp.items()
          if k != "gumbel_u"}
    sh["gumbel_u"] = inp["gumbel_u"]
    fresh, L1f = _run(m, sh)
    cached, L1c = _run(m, sh, pre=L1)
    # frames 0..5 of the shifted window come from the cache: their features were extracted in a different batch position
    assert (L1c - L1f).abs().max().item() <= 2e-5
    assert (cached - fresh).abs().max().item() <= 2e-5


def test_c5_one_clip_fp16x2_against_split_bf16():
    m, sd, inp = _setup(1, 544, 960, 1004)
    torch.cuda.reset_peak_memory_stats()
    out, L1 = _run(m, inp, precision="fp16x2")
    assert out.shape == (1, 1, 2176, 3840)
    ref, L1r = _run(m, inp, precision="bf16x3")
    err, err_l1 = (out - ref).abs().max().item(), (L1 - L1r).abs().max().item()
    print(f"c5 fp16x2 vs bf16x3: out {err:.2e}  L1_fea {err_l1:.2e}  peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    assert err <= TOL and err_l1 <= TOL


def test_v7_c3_frame_size_split_bf16_against_exact_fp32():
    """CVSR_V7 at the c3 frame size, one clip: default split-bf16 arithmetic against the exact-fp32
    kernels, and run-to-run reproducibility."""
    from arch.SIDECVSR_our import CVSR_V7
    from oracle.cvsr_v7_ref import make_inputs_v7, make_state_dict_v7
    m = CVSR_V7()
    m.load_state_dict(make_state_dict_v7(0), strict=True)
    m = m.cuda().eval()
    inp = make_inputs_v7(1, 272, 480, 1002)
    d = {k: v.cuda() for k, v in inp.items() if k != "gumbel_u"}
    noise = [u.cuda() for u in inp["gumbel_u"]]
    outs = {}
    with torch.no_grad():
        for prec in ("bf16x3", "bf16x3", "f32"):
            m.precision = prec
            out, L1 = m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=noise)
            torch.cuda.synchronize()
            if prec in outs:
                assert torch.equal(outs[prec][0], out) and torch.equal(outs[prec][1], L1)
            outs[prec] = (out, L1)
    err = (outs["bf16x3"][0] - outs["f32"][0]).abs().max().item()
    err_l1 = (outs["bf16x3"][1] - outs["f32"][1]).abs().max().item()
    print(f"v7 272x480 bf16x3 vs exact-fp32 kernels: out {err:.2e}  L1_fea {err_l1:.2e}")
    assert out.shape == (1, 1, 1088, 1920) and err <= TOL and err_l1 <= TOL
