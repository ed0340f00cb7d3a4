"""`CVSR_V8.capture` (cdfo_amd/graph.py): a whole inference forward replayed from a HIP graph is the eager forward --
same launches, bit-identical outputs -- with fresh Gumbel noise per replay (arch/SIDECVSR_our.py:2169 draws per call),
reproducible under torch.manual_seed, for the fresh and the cached-feature path."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(wseed=21):
    from arch.SIDECVSR_our import CVSR_V8
    from oracle.cvsr_v8_ref import make_state_dict
    m = CVSR_V8()
    m.load_state_dict(make_state_dict(wseed), strict=True)
    return m.cuda().eval()


def _inputs(B, H, W, seed):
    from oracle.cvsr_v8_ref import make_inputs
    a = make_inputs(B, H, W, seed)
    return {k: v.cuda() for k, v in a.items() if k != "gumbel_u"}, [u.cuda() for u in a["gumbel_u"]]


@pytest.mark.parametrize("B,H,W", [(2, 24, 40), (4, 32, 48)])
def test_replay_with_injected_noise_is_bit_identical_to_eager(B, H, W):
    m = _model()
    d0, n0 = _inputs(B, H, W, 500)
    d1, n1 = _inputs(B, H, W, 501)
    with torch.no_grad():
        cap = m.capture(d0["x"], d0["mvs0"], d0["mvs1"], d0["pms"], d0["rms"], d0["ufs"], gumbel_uniform=n0)
        for d, n in ((d0, n0), (d1, n1), (d0, n0)):
            want, want_l1 = m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=n)
            got, got_l1 = cap(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=n)
            torch.cuda.synchronize()
            assert torch.equal(got, want) and torch.equal(got_l1, want_l1)
    assert got.shape == (B, 1, 4 * H, 4 * W) and got_l1.shape == (B * 7, 64, H, W)


def test_replays_draw_fresh_noise_and_follow_the_generator_like_the_eager_path():
    m = _model(22)
    d, _ = _inputs(2, 16, 24, 510)
    args = (d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"])
    with torch.no_grad():
        torch.manual_seed(77)
        eager = [m(*args)[0].clone() for _ in range(3)]
        torch.manual_seed(123)                      # capture itself must not disturb the generator's sequence
        cap = m.capture(*args)
        torch.manual_seed(77)
        replayed = [cap(*args)[0].clone() for _ in range(3)]
    torch.cuda.synchronize()
    for e, r in zip(eager, replayed):
        assert torch.equal(e, r)
    assert not torch.equal(replayed[0], replayed[1])        # per-replay noise (the key is a device word, not a frozen argument)


def test_cached_feature_path_and_input_buffers():
    m = _model(23)
    d0, n0 = _inputs(1, 16, 16, 520)
    d1, n1 = _inputs(1, 16, 16, 521)
    with torch.no_grad():
        _, fea = m(d0["x"], None, d0["mvs1"], d0["pms"], d0["rms"], d0["ufs"], gumbel_uniform=n0)
        want, want_l1 = m(d1["x"], None, d1["mvs1"], d1["pms"], d1["rms"], d1["ufs"], fea, gumbel_uniform=n1)
        cap = m.capture(d0["x"], None, d0["mvs1"], d0["pms"], d0["rms"], d0["ufs"], fea, gumbel_uniform=n0)
        # write the next frame's operands straight into the graph's buffers, then replay without arguments
        for dst, src in zip(cap.inputs, [d1["x"], None, d1["mvs1"], d1["pms"], d1["rms"], d1["ufs"], fea, *n1]):
            if dst is not None:
                dst.copy_(src)
        got, got_l1 = cap.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, want) and torch.equal(got_l1, want_l1)
    with pytest.raises(ValueError):
        cap(torch.zeros(1, 7, 1, 24, 16, device="cuda"))


def test_capture_is_the_inference_schedule_and_refuses_out_of_range_examples():
    m = _model(24)
    d, n = _inputs(1, 16, 16, 530)
    args = (d["x"], None, d["mvs1"], d["pms"], d["rms"], d["ufs"])
    cap = m.capture(*args, gumbel_uniform=n)               # grad mode on, parameters require grad: still the no_grad schedule
    out, _ = cap.replay()
    assert not out.requires_grad
    with torch.no_grad():
        want, _ = m(*args, gumbel_uniform=n)
    assert torch.equal(out, want)
    m.FP16_WINDOW = (1e-30, 1e-29)                          # every workload now "leaves the fp16 range": the eager forward falls back
    with pytest.warns(UserWarning), pytest.raises(RuntimeError):
        m.capture(*args, gumbel_uniform=n)
    m.precision = "bf16x3"                                  # ... and the advice of the error message works
    cap = m.capture(*args, gumbel_uniform=n)
    with torch.no_grad():
        want, _ = m(*args, gumbel_uniform=n)
    assert torch.equal(cap.replay()[0], want)



def test_captured_forward_carries_the_fp16_range_guard():
    """Round 5: the guard's two probes are nodes of the captured graph and replay() reads them back.  A replay on operands that
    push the trunk's input out of the fp16 window is repeated eagerly in bf16x3 INTO the graph's output buffers -- the result the
    eager, guarded forward returns for the same operands and noise, bit for bit -- and the next replay on ordinary operands is the
    fp16x2 graph again."""
    m = _model(25)
    d, n = _inputs(1, 16, 24, 540)
    rest = (d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"])
    xs = d["x"] * 2.0 ** 13
    with torch.no_grad():
        cap = m.capture(d["x"], *rest, gumbel_uniform=n)
        assert cap.guarded
        want0, _ = m(d["x"], *rest, gumbel_uniform=n)
        got0, _ = cap.replay()
        assert cap.last_range["fallback"] is False and cap.last_range["trunk_input_amax"] > 0
        assert torch.equal(got0, want0)
        with pytest.warns(UserWarning, match="fp16 range"):
            want, want_l1 = m(xs, *rest, gumbel_uniform=n)
            m.finish_range_guard()
        assert m.last_range["fallback"], f"scenario drifted: {m.last_range}"
        want, want_l1 = want.clone(), want_l1.clone()
        m._warned_range = False
        with pytest.warns(UserWarning, match="fp16 range"):
            got, got_l1 = cap(xs, *rest, gumbel_uniform=n)
        torch.cuda.synchronize()
        assert cap.last_range["fallback"]
        assert torch.equal(got, want) and torch.equal(got_l1, want_l1)
        # ordinary operands again: the graph's own result stands
        got2, _ = cap(d["x"], *rest, gumbel_uniform=n)
        assert cap.last_range["fallback"] is False and torch.equal(got2, want0)
        # sync=False: the check is deferred to finish_range_guard() / last_range / the next replay
        cap.load(xs)
        out3, _ = cap.replay(sync=False)
        assert cap._pending
        assert cap.last_range["fallback"] and torch.equal(out3, want)



def test_pipelined_forward_settles_the_previous_forward_after_launching_the_next():
    """cdfo_amd.graph.PipelinedForward: two alternating graphs; submit(k + 1) launches first and reads forward k's probes afterwards.
    Every result -- ordinary operands, operands the guard rejects (repaired in bf16x3 into the tensors submit returned), ordinary
    again -- equals the eager, guarded forward bit for bit once the next submit / drain has returned."""
    m = _model(26)
    d, n = _inputs(1, 16, 24, 550)
    d2, n2 = _inputs(1, 16, 24, 551)
    rest = (d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"])
    rest2 = (d2["mvs0"], d2["mvs1"], d2["pms"], d2["rms"], d2["ufs"])
    xs = d["x"] * 2.0 ** 13
    seq = [(d["x"], rest, n), (xs, rest, n), (d2["x"], rest2, n2), (d["x"], rest, n), (xs, rest, n)]
    import warnings
    with torch.no_grad():
        pipe = m.capture_pipelined(d["x"], *rest, gumbel_uniform=n)
        assert pipe.guarded
        want = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for x, r, nn in seq:
                o, l1 = m(x, *r, gumbel_uniform=nn)
                m.finish_range_guard()
                want.append((o.clone(), l1.clone(), bool(m.last_range["fallback"])))
        assert [w[2] for w in want] == [False, True, False, False, True], "scenario drifted"
        m._warned_range = False
        held = []
        with pytest.warns(UserWarning, match="fp16 range"):
            for k, (x, r, nn) in enumerate(seq):
                held.append(pipe.submit(x, *r, gumbel_uniform=nn))
                if k:       # the PREVIOUS result is final now (and not yet overwritten: that happens at the submit after this one)
                    torch.cuda.synchronize()
                    assert torch.equal(held[k - 1][0], want[k - 1][0]) and torch.equal(held[k - 1][1], want[k - 1][1])
            pipe.drain()
        torch.cuda.synchronize()
        assert pipe.last_range["fallback"]
        assert torch.equal(held[-1][0], want[-1][0]) and torch.equal(held[-1][1], want[-1][1])


def test_v7_replay_with_injected_noise_matches_eager():
    """`CVSR_V7.capture` (round 5): the ~1 000 launches of a V7 forward (three pyramid levels x twelve neighbour pipelines on side
    streams) replayed from one HIP graph give the eager forward's result on the captured and on new operands."""
    from arch.SIDECVSR_our import CVSR_V7
    from oracle.cvsr_v7_ref import make_inputs_v7, make_state_dict_v7
    m = CVSR_V7()
    m.load_state_dict(make_state_dict_v7(5), strict=True)
    m = m.cuda().eval()

    def inputs(seed):
        a = make_inputs_v7(2, 16, 24, seed)
        return {k: v.cuda() for k, v in a.items() if k != "gumbel_u"}, [u.cuda() for u in a["gumbel_u"]]

    d0, n0 = inputs(600)
    d1, n1 = inputs(601)
    with torch.no_grad():
        cap = m.capture(d0["x"], d0["mvs0"], d0["mvs1"], d0["pms"], d0["rms"], d0["ufs"], gumbel_uniform=n0)
        for d, n in ((d0, n0), (d1, n1)):
            want, want_l1 = m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=n)
            got, got_l1 = cap(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=n)
            torch.cuda.synchronize()
            # the attention Gram sums of the alignment are atomics-free and fixed-order: the replay is the eager arithmetic
            assert (got - want).abs().max().item() <= 1e-6 and (got_l1 - want_l1).abs().max().item() <= 1e-6
