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
import concurrent.futures as cf
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-3

# ---- the CPU oracle's full-size forwards (the file's wall time: ~5 minutes of host CPU, during which the GPU used to idle) run
# in two worker threads, started when the session's collection ends (tests/conftest.py::pytest_collection_finish calls prefetch()):
# the GPU tests of the files that run before this one proceed meanwhile, and each test below only waits for ITS reference.
# key -> (B, H, W, input seed, weight seed, perturb, clip)
ORACLE_JOBS = {
    "c2": (4, 120, 240, 1001, 0, True, 0),
    "c3": (8, 272, 480, 1002, 0, True, 0),
    "c3_w1": (3, 272, 480, 1012, 1, True, 0),
    "c3_w2": (3, 272, 480, 1022, 2, False, 0),
    "strip": (1, 272, 960, 1004, 0, True, 0),
    "c5": (1, 544, 960, 1005, 0, True, 0),
    "v7_c3": (3, 272, 480, 2001, 0, True, 0),        # CVSR_V7 (arch.py:4215-4367) at the size bench.py times it at
}
_pool = None
_futures = {}


def _oracle_job(key):
    from oracle.cvsr_v8_ref import cvsr_v8_forward, make_inputs, make_state_dict
    B, H, W, seed, wseed, perturb, clip = ORACLE_JOBS[key]
    if key.startswith("v7"):
        from oracle.cvsr_v7_ref import cvsr_v7_forward, make_inputs_v7, make_state_dict_v7
        inp = make_inputs_v7(B, H, W, seed)
        one = {k: (v[clip:clip + 1] if k != "gumbel_u" else [u[clip:clip + 1] for u in v]) for k, v in inp.items()}
        del inp
        with torch.no_grad():
            return cvsr_v7_forward(make_state_dict_v7(wseed), one["x"], one["mvs0"], one["mvs1"], one["pms"], one["rms"], one["ufs"], None,
                                   one["gumbel_u"])
    sd = make_state_dict(wseed, perturb=perturb)
    inp = make_inputs(B, H, W, seed, pad_rows={272: 2, 544: 4}.get(H, 0))
    one = {k: (v[clip:clip + 1] if k != "gumbel_u" else [u[clip:clip + 1] for u in v]) for k, v in inp.items()}
    del inp
    with torch.no_grad():
        return cvsr_v8_forward(sd, one["x"], None, one["mvs1"], one["pms"], one["rms"], one["ufs"], None, one["gumbel_u"])


def prefetch():
    """Start every oracle forward of this file in the background (idempotent).  Two at a time, each on half the host's threads."""
    global _pool
    if _pool is not None:
        return
    torch.set_num_threads(max(2, (os.cpu_count() or 4) // 2) if torch.get_num_threads() > 4 else torch.get_num_threads())
    _pool = cf.ThreadPoolExecutor(max_workers=2, thread_name_prefix="oracle")
    for key in ("c5", "c3", "c2", "c3_w1", "c3_w2", "strip", "v7_c3"):        # the long one (c5: minutes) first on one worker, the others on the second
        _futures[key] = _pool.submit(_oracle_job, key)


def oracle_reference(key):
    prefetch()
    return _futures[key].result()


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


def _job_setup(key):
    B, H, W, seed, wseed, perturb, _ = ORACLE_JOBS[key]
    return _setup(B, H, W, seed, wseed=wseed, perturb=perturb)


def test_c2_batch_against_the_oracle_and_single_clip_forwards():
    m, sd, inp = _job_setup("c2")
    out, L1 = _run(m, inp)
    assert out.shape == (4, 1, 480, 960) and L1.shape == (28, 64, 120, 240)
    ref, L1_ref = oracle_reference("c2")
    assert (out[:1].cpu() - ref).abs().max().item() <= TOL
    assert (L1[:7].cpu() - L1_ref).abs().max().item() <= TOL
    for b in (1, 3):
        ob, Lb = _run(m, inp, slice(b, b + 1))
        assert (out[b:b + 1] - ob).abs().max().item() <= 2e-5            # summation partitions differ with B, nothing else
        assert (L1[7 * b:7 * b + 7] - Lb).abs().max().item() <= 2e-5


def _clip_vs_oracle(key):
    m, sd, inp = _job_setup(key)
    clip = ORACLE_JOBS[key][6]
    out, L1 = _run(m, inp)
    ref, L1_ref = oracle_reference(key)
    e_out = (out[clip:clip + 1].cpu() - ref).abs().max().item()
    e_l1 = (L1[7 * clip:7 * clip + 7].cpu() - L1_ref).abs().max().item()
    return e_out, e_l1, out, m


def test_c3_clip0_against_the_oracle():
    """The headline configuration (8 clips of 272x480, default fp16x2 arithmetic) against the CPU oracle on clip 0 -- an
    index bug that only shows above the small golden sizes cannot cancel here, as it could between two modes of one code base."""
    e_out, e_l1, out, m = _clip_vs_oracle("c3")
    print(f"c3 B=8 clip 0, fp16x2 vs CPU oracle: out {e_out:.2e}  L1_fea {e_l1:.2e}   range guard {m.last_range}")
    assert out.shape == (8, 1, 1088, 1920) and e_out <= TOL and e_l1 <= TOL
    assert m.last_range is not None and not m.last_range["fallback"]        # the default weights stay inside the fp16 window


@pytest.mark.parametrize("key", ["c3_w1", "c3_w2"])
def test_c3_clip0_against_the_oracle_other_weight_seeds(key):
    """The fp16x2 margin over more than one draw of the weights: two further seeds -- one with the
    LayerNorm affines / temperatures / biases perturbed off their initial values, one plain random init like bench.py's -- at
    the c3 frame size on the grouped three-neighbours-per-launch schedule (B = 3), clip 0 against the CPU oracle."""
    e_out, e_l1, out, m = _clip_vs_oracle(key)
    print(f"c3 B=3 clip 0, weights {key} {ORACLE_JOBS[key][4:6]}, fp16x2 vs CPU oracle: out {e_out:.2e}  L1_fea {e_l1:.2e}   "
          f"range guard {m.last_range}")
    assert e_out <= TOL and e_l1 <= TOL
    assert m.last_range is not None and not m.last_range["fallback"]


def test_c5_clip_against_the_oracle():
    """One full c5 clip (540 rows zero-padded to 544 x 960, test_LD_37.py:24-26 semantics) in the default fp16x2 arithmetic
    against the CPU ORACLE: `out` and `L1_fea` <= 1e-3.  The only check in which a 544-row column attention (H x H map per image
    column) and 960-wide row attention of the same frame meet the reference's restatement rather than another mode of this
    code base.  The oracle forward is ~2 minutes of host CPU."""
    e_out, e_l1, out, m = _clip_vs_oracle("c5")
    print(f"c5 1x544x960, fp16x2 vs CPU oracle: out {e_out:.2e}  L1_fea {e_l1:.2e}   range guard {m.last_range}")
    assert out.shape == (1, 1, 2176, 3840) and e_out <= TOL and e_l1 <= TOL
    assert (out[..., 2160:, :].abs().max().item()) < 10.0       # the 16 padded output rows exist and are finite (callers crop them)


def test_c5_width_strip_against_the_oracle():
    """A 272x960 strip (the c5 frame's full width: 960-pixel rows through the row attention, half its height) of one clip
    against the CPU oracle; the full 544x960 oracle forward takes minutes on the host."""
    e_out, e_l1, out, m = _clip_vs_oracle("strip")
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


def _rescaled(sd, plan):
    """Function-preserving rescaling of single Block_ bodies: body.0 (weight and bias) times s, body.2 weight times 1/s.  LeakyReLU is
    positively homogeneous, so the network computes the same function in exact arithmetic -- but the 256-channel fp16 intermediate
    of those blocks lives s times higher / lower in fp16's range (arch.py:383-387)."""
    out = {k: v.clone() for k, v in sd.items()}
    for (g, b), s in plan.items():
        p = f"recon_trunk.body.{g}.body.{b}.body."
        out[p + "0.weight"] *= s
        out[p + "0.bias"] *= s
        out[p + "2.weight"] /= s
    return out


def test_fp16_margin_on_trained_like_and_rescaled_weights():
    """VERDICT r3 weak #1 / task 8: the fp16x2 margin was only ever measured on random-init weights.  Here: (i) weights after 200 Adam
    steps of the reference's training loop (train_LD_37.py:359-381) on synthetic SR clips, run by this repository's own HIP training
    path (tests/trained_like.py); (ii) the same weights with single blocks' bodies rescaled by 1/16 ... 16 (function-preserving, so
    ONE oracle forward is the reference for all of them).  One clip at the c3 frame size against the CPU oracle, bound 1e-3."""
    from arch.SIDECVSR_our import CVSR_V8
    from oracle.cvsr_v8_ref import cvsr_v8_forward, make_inputs, make_state_dict
    from trained_like import trained_like_state_dict
    sd, losses = trained_like_state_dict(wseed=0, data_seed=4000, steps=200)
    init = make_state_dict(0, perturb=False)
    moved = {k: ((sd[k] - init[k]).norm() / init[k].norm()).item() for k in sd if init[k].numel() > 64 and init[k].norm() > 0}
    trunk_moved = sorted(v for k, v in moved.items() if k.startswith("recon_trunk"))
    print(f"trained-like weights: loss {sum(losses[:10]) / 10:.4f} -> {sum(losses[-10:]) / 10:.4f} over {len(losses)} Adam steps; relative weight "
          f"change: trunk median {trunk_moved[len(trunk_moved) // 2]:.3f}, max over tensors {max(moved.values()):.3f}")
    assert sum(losses[-10:]) < 0.8 * sum(losses[:10])              # the loop learned something: these are not the init weights
    inp = make_inputs(1, 272, 480, 1032, pad_rows=2)
    prefetch()
    fut = _pool.submit(lambda: cvsr_v8_forward(sd, inp["x"], None, inp["mvs1"], inp["pms"], inp["rms"], inp["ufs"], None, inp["gumbel_u"]))
    plans = {"trained": {}, "block(0,0)x16": {(0, 0): 16.0}, "block(3,1)/16": {(3, 1): 1 / 16.0},
             "block(6,2)x4,(2,0)/4": {(6, 2): 4.0, (2, 0): 0.25},
             "all blocks alternating x8 / /8": {(g, b): (8.0 if (g + b) % 2 == 0 else 0.125) for g in range(7) for b in range(3)}}
    outs = {}
    for name, plan in plans.items():
        m = CVSR_V8()
        m.load_state_dict(_rescaled(sd, plan), strict=True)
        m = m.cuda().eval()
        out, L1 = _run(m, inp)
        outs[name] = (out.cpu(), L1.cpu(), dict(m.last_range))
        if name == "trained":
            exact, _ = _run(m, inp, precision="bf16x3")
            outs["trained, bf16x3 (fp32-grade)"] = (exact.cpu(), L1.cpu(), None)
    with torch.no_grad():
        ref, L1_ref = fut.result()
    worst = 0.0
    for name, (o, l1, rng) in outs.items():
        e, el = (o - ref).abs().max().item(), (l1 - L1_ref).abs().max().item()
        print(f"fp16 margin, {name}: out {e:.2e}  L1_fea {el:.2e}  (|out| max {ref.abs().max().item():.2f})  range guard {rng}")
        if "bf16x3" not in name:
            worst = max(worst, e)
            assert rng is not None and not rng["fallback"]
        assert e <= TOL and el <= TOL, name
    print(f"fp16 margin: worst fp16x2 case {worst:.2e} of the {TOL:.0e} bound")


@pytest.mark.parametrize("precision", ["fp16x2", "bf16x3"])
def test_v7_c3_batch_against_the_oracle(precision):
    """CVSR_V7 (arch/SIDECVSR_our.py:4215-4367) where bench.py times it: clip 0 of a 3-clip batch of 272x480 in both 16-bit modes against
    oracle/cvsr_v7_ref.py (whose DCN is the C restatement oracle/dcn_ref.c) run on that clip alone; bound 1e-3 on `out` and `L1_fea`.
    The small goldens (test_gpu_cvsr_v7.py) cannot see an index error that only shows above one tile per level."""
    from arch.SIDECVSR_our import CVSR_V7
    from oracle.cvsr_v7_ref import make_inputs_v7, make_state_dict_v7
    B, H, W, seed, wseed, _, clip = ORACLE_JOBS["v7_c3"]
    m = CVSR_V7()
    m.load_state_dict(make_state_dict_v7(wseed), strict=True)
    m = m.cuda().eval()
    m.precision = precision
    inp = make_inputs_v7(B, H, W, seed)
    dev = {k: v.cuda() for k, v in inp.items() if k != "gumbel_u"}
    with torch.no_grad():
        out, L1 = m(dev["x"], dev["mvs0"], dev["mvs1"], dev["pms"], dev["rms"], dev["ufs"], gumbel_uniform=[u.cuda() for u in inp["gumbel_u"]])
    torch.cuda.synchronize()
    ref, L1_ref = oracle_reference("v7_c3")
    e_out = (out[clip:clip + 1].cpu() - ref).abs().max().item()
    e_l1 = (L1[7 * clip:7 * clip + 7].cpu() - L1_ref).abs().max().item()
    print(f"CVSR_V7 B={B} {H}x{W} clip {clip}, {precision} vs CPU oracle: out {e_out:.2e}  L1_fea {e_l1:.2e}")
    assert out.shape == (B, 1, 4 * H, 4 * W) and L1.shape == (7 * B, 64, H, W)
    assert e_out <= TOL and e_l1 <= TOL
