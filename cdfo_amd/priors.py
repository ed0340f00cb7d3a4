"""File readers of the reference's evaluation loop: the 8-bit grey PNG frames and coding priors and
the NPY residual / motion-vector dumps that ``test_LD_22_FPS.py:20-97,155-176`` (= ``train_LD_37.py:56-126``) reads frame
by frame with ``cv2.imread(path, 0)`` and ``np.load``.  Here a sequence is read ONCE into uint8 / int arrays that
``cdfo_amd.streaming.StreamingSR`` uploads and keeps resident in HBM; the ``/255``, the 270 -> 272 zero padding
(``:24-26``) and the seven-frame window gathering happen on the device.

``cv2`` is not available in this image, so the PNG decoder is a small stdlib one (zlib + the five PNG row filters) for
the formats the data set uses: 8-bit greyscale (colour type 0; an alpha channel is dropped), non-interlaced.

Directory layout and the reference's naming (``test_LD_22_FPS.py:143-170``):
    <lr_dir>/<any sorted file names>.png                      LR luma frames, index i = position in the sorted list
    <side_dir>/part_m/%05d_M_mask.png   partition maps        (index max(1, i): the priors of frame 0 are frame 1's)
    <side_dir>/res/%05d_res.npy         residual maps, [:,:,0]
    <side_dir>/unfiltered/%05d_unflt.png
    <side_dir>/mvl0/%05d_mvl0.npy, <side_dir>/mvl1/%05d_mvl1.npy     decoder motion fields [H,W,3]"""
from __future__ import annotations

import os
import struct
import zlib
from typing import Dict

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"


def read_gray_png(path: str) -> np.ndarray:
    """8-bit greyscale PNG -> uint8 [H,W] (what ``cv2.imread(path, 0)`` returns for such a file)."""
    data = open(path, "rb").read()
    if data[:8] != _SIG:
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    if hdr is None:
        raise ValueError(f"{path}: no IHDR chunk")
    W, H, depth, ctype, _, _, interlace = hdr
    if depth != 8 or ctype not in (0, 4) or interlace:
        raise NotImplementedError(f"{path}: only 8-bit non-interlaced greyscale PNGs are supported "
                                  f"(bit depth {depth}, colour type {ctype}, interlace {interlace})")
    bpp = 1 if ctype == 0 else 2
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8)
    stride = W * bpp
    if raw.size != H * (stride + 1):
        raise ValueError(f"{path}: truncated image data")
    rows = raw.reshape(H, stride + 1)
    out = np.zeros((H, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.uint8)
    for y in range(H):
        f, line = int(rows[y, 0]), rows[y, 1:]
        if f == 0:
            cur = line.copy()
        elif f == 2:                                           # Up
            cur = line + prev
        elif f == 1:                                           # Sub: running sum along the row, per byte lane
            cur = line.copy()
            for c in range(bpp):
                cur[c::bpp] = np.cumsum(line[c::bpp], dtype=np.uint64).astype(np.uint8)
        elif f in (3, 4):                                      # Average / Paeth: sequential in x
            cur = np.empty(stride, dtype=np.uint8)
            ln, pv = line.tolist(), prev.tolist()
            res = [0] * stride
            for x in range(stride):
                a = res[x - bpp] if x >= bpp else 0
                b = pv[x]
                if f == 3:
                    pred = (a + b) >> 1
                else:
                    c = pv[x - bpp] if x >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                res[x] = (ln[x] + pred) & 255
            cur[:] = res
        else:
            raise ValueError(f"{path}: bad filter type {f} in row {y}")
        out[y] = cur
        prev = cur
    return out[:, ::bpp].copy() if bpp == 2 else out


def write_gray_png(path: str, img: np.ndarray, filter_type: int = 0) -> None:
    """uint8 [H,W] -> 8-bit greyscale PNG with every row filtered by ``filter_type`` (0-4).  For tests and tools."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape
    rows = bytearray()
    prev = np.zeros(W, dtype=np.int32)
    for y in range(H):
        cur = img[y].astype(np.int32)
        left = np.concatenate([[0], cur[:-1]])
        ul = np.concatenate([[0], prev[:-1]])
        if filter_type == 0:
            enc = cur
        elif filter_type == 1:
            enc = cur - left
        elif filter_type == 2:
            enc = cur - prev
        elif filter_type == 3:
            enc = cur - ((left + prev) >> 1)
        else:
            p = left + prev - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            enc = cur - pred
        rows.append(filter_type)
        rows += (enc & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(bytes(rows), 6)) + chunk(b"IEND", b""))


def load_sequence(lr_dir: str, side_dir: str) -> Dict[str, np.ndarray]:
    """One sequence of the reference's test layout -> the arrays ``StreamingSR`` takes:
    lr, pms, ufs uint8 [T,H,W]; rms [T,H,W] (dtype of the ``*_res.npy`` files); mvl0, mvl1 [T,H,W,3].
    Index t = file index t; entry 0 of the priors (which the reference never reads: ``ii = max(1, i)``) repeats entry 1."""
    names = sorted(n for n in os.listdir(lr_dir) if n.lower().endswith(".png"))
    if not names:
        raise FileNotFoundError(f"no PNG frames in {lr_dir}")
    T = len(names)
    lr = np.stack([read_gray_png(os.path.join(lr_dir, n)) for n in names])

    def per_frame(fn):
        items = [fn("%05d" % max(1, t)) for t in range(T)] if T > 1 else [fn("%05d" % 1)]
        return np.stack(items)

    pms = per_frame(lambda i: read_gray_png(os.path.join(side_dir, "part_m", i + "_M_mask.png")))
    ufs = per_frame(lambda i: read_gray_png(os.path.join(side_dir, "unfiltered", i + "_unflt.png")))
    rms = per_frame(lambda i: np.load(os.path.join(side_dir, "res", i + "_res.npy"))[:, :, 0])
    mvl0 = per_frame(lambda i: np.load(os.path.join(side_dir, "mvl0", i + "_mvl0.npy")))
    mvl1 = per_frame(lambda i: np.load(os.path.join(side_dir, "mvl1", i + "_mvl1.npy")))
    for name, arr in (("pms", pms), ("ufs", ufs), ("rms", rms)):
        if arr.shape != lr.shape:
            raise ValueError(f"{name} planes are {arr.shape[1:]} but the LR frames are {lr.shape[1:]}")
    return dict(lr=lr, pms=pms, rms=rms, ufs=ufs, mvl0=mvl0, mvl1=mvl1)
