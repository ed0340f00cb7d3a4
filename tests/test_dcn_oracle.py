"""Pins the C oracle of the DCN forward (oracle/dcn_ref.c): the reference's own known-answer test
(ops/dcn/simple_check.py:8-22) plus derived identities against torch-cpu F.conv2d."""
import numpy as np
import torch
import torch.nn.functional as F

from dcn_oracle import dcn_forward_ref


def test_reference_simple_check_known_answer():
    # DeformConv(2, 1, 3, padding=1, deformable_groups=2), weights == 1, input arange(18).view(1,2,3,3); the offsets make
    # tap (i,j) sample the centre pixel, so every output = 9 * (x0 + x1).  Expected vector from simple_check.py:19.
    off = np.array([1, 1, 1, 0, 1, -1, 0, 1, 0, 0, 0, -1, -1, 1, -1, 0, -1, -1], np.float32)
    offset = np.tile(off.reshape(1, 18, 1, 1), (1, 2, 3, 3))
    x = np.arange(18, dtype=np.float32).reshape(1, 2, 3, 3)
    w = np.ones((1, 2, 3, 3), np.float32)
    out = dcn_forward_ref(x, offset, None, w, None, pad=1, dg=2)
    gt = np.array([81, 99, 117, 135, 153, 171, 189, 207, 225], np.float32)
    assert np.abs(gt - out.reshape(-1)).sum() < 1e-8


def _rand(shape, seed):
    return np.random.RandomState(seed).standard_normal(shape).astype(np.float32)


def test_zero_offset_unit_mask_is_conv2d():
    for (C, Co, k, s, p, d, g, dg) in [(8, 6, 3, 1, 1, 1, 1, 4), (8, 8, 3, 2, 1, 1, 2, 2), (4, 4, 3, 1, 2, 2, 1, 1),
                                       (6, 3, 1, 1, 0, 1, 3, 3)]:
        x, w, b = _rand((2, C, 9, 11), 1), _rand((Co, C // g, k, k), 2), _rand((Co,), 3)
        ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), s, p, d, g).numpy()
        Ho, Wo = ref.shape[2:]
        off = np.zeros((2, 2 * dg * k * k, Ho, Wo), np.float32)
        msk = np.ones((2, dg * k * k, Ho, Wo), np.float32)
        out = dcn_forward_ref(x, off, msk, w, b, s, p, d, g, dg)
        assert np.abs(out - ref).max() < 1e-5
        out1 = dcn_forward_ref(x, off, None, w, b, s, p, d, g, dg)      # mask == 1  <=>  DCNv1
        assert np.array_equal(out, out1)


def test_integer_offsets_are_a_shifted_conv_and_far_offsets_give_zero():
    x, w = _rand((1, 4, 10, 12), 5), _rand((5, 4, 3, 3), 6)
    off = np.zeros((1, 18, 10, 12), np.float32)
    off[:, 0::2] = 2.0   # every tap 2 rows down
    off[:, 1::2] = -1.0  # and 1 column left
    out = dcn_forward_ref(x, off, None, w, None, pad=1)
    xs = np.zeros_like(x)
    xs[:, :, :-2, 1:] = x[:, :, 2:, :-1]          # xs[y][x] = x[y+2][x-1]
    ref = F.conv2d(torch.from_numpy(xs), torch.from_numpy(w), None, 1, 1).numpy()
    assert np.abs(out[:, :, 1:-1, 1:-1] - ref[:, :, 1:-1, 1:-1]).max() < 1e-5
    off[:] = 1000.0
    assert np.abs(dcn_forward_ref(x, off, None, w, None, pad=1)).max() == 0.0


def test_mask_scales_linearly_and_fractional_offsets_interpolate():
    x, w = _rand((1, 2, 6, 6), 7), _rand((3, 2, 3, 3), 8)
    off = np.full((1, 18, 6, 6), 0.5, np.float32)
    m1 = np.random.RandomState(9).uniform(0, 1, (1, 9, 6, 6)).astype(np.float32)
    a = dcn_forward_ref(x, off, m1, w, None, pad=1)
    b = dcn_forward_ref(x, off, 2 * m1, w, None, pad=1)
    assert np.abs(2 * a - b).max() < 1e-5
    # half-pixel offsets == conv over the 2x2-averaged image (interior only; borders differ by the zero rule)
    xa = 0.25 * (x[:, :, :-1, :-1] + x[:, :, 1:, :-1] + x[:, :, :-1, 1:] + x[:, :, 1:, 1:])
    ref = F.conv2d(torch.from_numpy(xa), torch.from_numpy(w), None).numpy()       # valid conv on the 5x5 average
    full = dcn_forward_ref(x, off, None, w, None, pad=1)
    assert np.abs(full[:, :, 1:4, 1:4] - ref).max() < 1e-5
