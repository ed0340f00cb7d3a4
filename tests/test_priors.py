"""File readers of the evaluation loop (cdfo_amd/priors.py).  The reference reads with cv2.imread /
np.load (test_LD_22_FPS.py:20-97); cv2 is absent here, so the PNG decoder is checked against PNG files written by an
independent encoder path (all five row filters) and the directory loader against the reference's naming rules."""
import os

import numpy as np
import pytest

from cdfo_amd.priors import load_sequence, read_gray_png, write_gray_png


@pytest.mark.parametrize("filter_type", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("shape", [(1, 1), (7, 5), (30, 48)])
def test_png_roundtrip_every_row_filter(tmp_path, filter_type, shape):
    rs = np.random.RandomState(filter_type * 10 + shape[0])
    img = rs.randint(0, 256, size=shape).astype(np.uint8)
    img[0, 0], img[-1, -1] = 0, 255
    p = str(tmp_path / "a.png")
    write_gray_png(p, img, filter_type)
    assert np.array_equal(read_gray_png(p), img)


def test_png_decoder_against_an_independent_codec(tmp_path):
    """cv2 (what test_LD_22_FPS.py:20-74 reads with) is absent; Pillow (libpng) is an independent PNG codec that IS here:
    files it writes (adaptive per-row filters, several compression settings, grey + alpha) decode to the same pixels, and
    files written by write_gray_png decode identically in Pillow."""
    Image = pytest.importorskip("PIL.Image")
    rs = np.random.RandomState(11)
    smooth = (np.add.outer(np.arange(37) * 3, np.arange(53) * 2) % 256).astype(np.uint8)      # makes libpng pick Sub/Up/Avg/Paeth
    for k, img in enumerate((rs.randint(0, 256, (30, 48)).astype(np.uint8), smooth, np.full((5, 9), 200, np.uint8))):
        for j, kw in enumerate((dict(), dict(optimize=True), dict(compress_level=1), dict(compress_level=9))):
            p = str(tmp_path / f"pil_{k}_{j}.png")
            Image.fromarray(img, mode="L").save(p, **kw)
            assert np.array_equal(read_gray_png(p), img)
        la = np.stack([img, 255 - img], -1)
        p = str(tmp_path / f"pil_la_{k}.png")
        Image.fromarray(la, mode="LA").save(p)
        assert np.array_equal(read_gray_png(p), img)                      # alpha dropped, like cv2.imread(path, 0)
        for ft in range(5):
            p = str(tmp_path / f"own_{k}_{ft}.png")
            write_gray_png(p, img, ft)
            assert np.array_equal(np.asarray(Image.open(p)), img)


def test_png_rejects_what_it_does_not_decode(tmp_path):
    p = str(tmp_path / "bad.png")
    open(p, "wb").write(b"not a png")
    with pytest.raises(ValueError):
        read_gray_png(p)
    import struct, zlib
    def chunk(k, b):
        return struct.pack(">I", len(b)) + k + b + struct.pack(">I", zlib.crc32(k + b) & 0xFFFFFFFF)
    rgb = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 1, 1, 8, 2, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(b"\x00\x01\x02\x03")) + chunk(b"IEND", b"")
    open(p, "wb").write(rgb)
    with pytest.raises(NotImplementedError):
        read_gray_png(p)


def _make_sequence(root, T, H, W, seed):
    rs = np.random.RandomState(seed)
    lr_dir, side = os.path.join(root, "lr"), os.path.join(root, "side")
    for d in ("part_m", "res", "unfiltered", "mvl0", "mvl1"):
        os.makedirs(os.path.join(side, d))
    os.makedirs(lr_dir)
    data = dict(lr=rs.randint(0, 256, (T, H, W)).astype(np.uint8), pms=rs.randint(0, 256, (T, H, W)).astype(np.uint8),
                ufs=rs.randint(0, 256, (T, H, W)).astype(np.uint8), rms=rs.randint(-128, 128, (T, H, W, 3)).astype(np.int8),
                mvl0=rs.randint(-64, 64, (T, H, W, 3)).astype(np.int16), mvl1=rs.randint(-64, 64, (T, H, W, 3)).astype(np.int16))
    for t in range(T):
        write_gray_png(os.path.join(lr_dir, "f%03d.png" % t), data["lr"][t], t % 5)
        if t >= 1:                                   # the reference's side-info files start at 00001
            i = "%05d" % t
            write_gray_png(os.path.join(side, "part_m", i + "_M_mask.png"), data["pms"][t], (t + 1) % 5)
            write_gray_png(os.path.join(side, "unfiltered", i + "_unflt.png"), data["ufs"][t], (t + 2) % 5)
            np.save(os.path.join(side, "res", i + "_res.npy"), data["rms"][t])
            np.save(os.path.join(side, "mvl0", i + "_mvl0.npy"), data["mvl0"][t])
            np.save(os.path.join(side, "mvl1", i + "_mvl1.npy"), data["mvl1"][t])
    return lr_dir, side, data


def test_load_sequence_follows_the_reference_naming(tmp_path):
    lr_dir, side, data = _make_sequence(str(tmp_path), 5, 12, 16, 3)
    seq = load_sequence(lr_dir, side)
    assert np.array_equal(seq["lr"], data["lr"])
    for k in ("pms", "ufs"):
        assert np.array_equal(seq[k][1:], data[k][1:]) and np.array_equal(seq[k][0], data[k][1])      # ii = max(1, i)
    assert np.array_equal(seq["rms"][1:], data["rms"][1:, :, :, 0]) and np.array_equal(seq["rms"][0], data["rms"][1, :, :, 0])
    assert np.array_equal(seq["mvl1"][2], data["mvl1"][2]) and seq["mvl0"].shape == (5, 12, 16, 3)


@pytest.mark.gpu
def test_streaming_from_files_equals_streaming_from_arrays(tmp_path):
    """The whole reader -> device-resident streaming loop chain against the same loop fed with the arrays directly."""
    import torch
    from arch.SIDECVSR_our import CVSR_V8
    from cdfo_amd.streaming import StreamingSR
    from oracle.cvsr_v8_ref import make_state_dict
    lr_dir, side, data = _make_sequence(str(tmp_path), 4, 16, 24, 5)
    seq = load_sequence(lr_dir, side)
    m = CVSR_V8()
    m.load_state_dict(make_state_dict(4), strict=True)
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(1)
    noise = [[torch.rand(1, 64, 16, 24, generator=g).clamp_min(1e-6) for _ in range(6)] for _ in range(4)]
    a = StreamingSR(m, seq["lr"], seq["pms"], seq["rms"], seq["ufs"], seq["mvl0"], seq["mvl1"], gumbel_uniform=noise).run()
    fix = lambda k: np.concatenate([data[k][1:2], data[k][1:]])      # noqa: E731  entry 0 <- entry 1
    b = StreamingSR(m, data["lr"], fix("pms"), fix("rms")[..., 0], fix("ufs"), fix("mvl0"), fix("mvl1"),
                    gumbel_uniform=noise).run()
    for x, y in zip(a, b):
        assert x.shape == (1, 1, 64, 96) and torch.equal(x, y)
