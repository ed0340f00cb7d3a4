"""Streaming evaluation of one sequence on the device: the caller side of ``CVSR_V8.forward(..., pre_L1_fea)``.

Mirrors the reference's evaluation loop (``test_LD_22_FPS.py:155-189``) with the sequence resident in HBM: the LR
frames and the coding priors are uploaded ONCE, each step gathers its seven-frame window by index on the device (the
reference re-reads and re-uploads six repeated frames per step), converts the decoder's motion field to the seven
per-slot flows (``mv2mvs``, ``:100-122``), applies the sequence-boundary fix-ups (``modify_mv_for_end_frames``,
``:200-225``) and calls the model with the feature cache returned by the previous step, so that only ONE new frame
goes through feature extraction (``arch/SIDECVSR_our.py:4420-4427``).  Quirks of the reference loop that are kept:
window indices are clipped to the sequence (``generate_input_index``, ``:14-17``); the priors and motion fields of
frame 0 are read from entry 1 (``ii = max(1, i)``, ``:36,53,66,168``); ``x / 0`` in ``mv2mvs`` stays ``inf`` (only NaN
becomes 0).
"""
from __future__ import annotations

import time
from typing import List, Optional, Sequence

import torch

NFRAMES = 7


def generate_input_index(center_index: int, frame_number: int, max_index: int) -> torch.Tensor:
    """test_LD_22_FPS.py:14-17."""
    return (torch.arange(frame_number) - (frame_number // 2) + center_index).clamp_(0, max_index)


def mv2mvs(mv: torch.Tensor) -> torch.Tensor:
    """test_LD_22_FPS.py:100-122.  mv: [H,W,3] (any float/int dtype, any device) -> flows [7,2,H,W] in pixels, ready for
    ``mvs.unsqueeze(0)`` (the reference's ``permute(0,1,4,2,3)`` is folded in)."""
    m = mv.to(torch.float32)
    d = m[..., 2] * -1.0
    fx, fy = m[..., 1] / d, m[..., 0] / d                       # components swapped (:103)
    f = torch.stack([torch.where(torch.isnan(fx), torch.zeros_like(fx), fx),
                     torch.where(torch.isnan(fy), torch.zeros_like(fy), fy)], 0)      # [2,H,W]
    scale = torch.tensor([3.0, 2.0, 1.0, 0.0, -1.0, -2.0, -3.0], device=m.device).view(7, 1, 1, 1)
    out = f.unsqueeze(0) * scale
    out[3] = 0.0                                                # slot 3 stays zero (0 * inf would be NaN)
    return out / (4.0 * 32.0)


def modify_mv_for_end_frames(i: int, mvs: torch.Tensor, max_idx: int) -> torch.Tensor:
    """test_LD_22_FPS.py:200-225 on mvs [B,7,2,H,W], in place; max_idx = number of frames in the sequence."""
    if i == 0:
        mvs[:, 0:3] = 0.0
    if i == 1:
        mvs[:, 0] = mvs[:, 2]
        mvs[:, 1] = mvs[:, 2]
    if i == 2:
        mvs[:, 0] = mvs[:, 1]
    if i == max_idx - 1:
        mvs[:, 4:7] = 0.0
    if i == max_idx - 2:
        mvs[:, 5] = mvs[:, 4]
        mvs[:, 6] = mvs[:, 4]
    if i == max_idx - 3:
        mvs[:, 6] = mvs[:, 5]
    return mvs


class StreamingSR:
    """One sequence, device resident.

    lr, pms, ufs : [T,H,W] pixel planes in file units (0..255), rms : [T,H,W] residual map in file units
    (``*_res.npy[:,:,0]``), mvl0 / mvl1 : [T,H,W,3] decoder motion fields (``*_mvl0.npy``); index t = file index t.
    H, W are padded with zero rows / columns to multiples of 8 (``test_LD_37.py:24-26`` pads 270 -> 272) and the
    output is cropped back to 4H x 4W.
    """

    def __init__(self, model, lr, pms, rms, ufs, mvl0, mvl1, device: Optional[torch.device] = None,
                 gumbel_uniform: Optional[Sequence] = None, use_graph: bool = False):
        dev = torch.device(device) if device is not None else next(model.parameters()).device
        if dev.type != "cuda":
            raise NotImplementedError("StreamingSR needs the model on a GPU (HIP path, no CPU fallback)")
        self.model, self.dev = model, dev
        as_dev = lambda t: torch.as_tensor(t).to(dev)
        lr = as_dev(lr)
        self.T, self.H, self.W = int(lr.shape[0]), int(lr.shape[1]), int(lr.shape[2])
        self.Hp, self.Wp = (self.H + 7) // 8 * 8, (self.W + 7) // 8 * 8

        def plane(t):                        # [T,H,W] file units -> float32 / 255, zero padded
            out = torch.zeros((self.T, self.Hp, self.Wp), dtype=torch.float32, device=dev)
            out[:, :self.H, :self.W] = as_dev(t).to(torch.float32) / 255.0
            return out

        self.lr, self.pms, self.rms, self.ufs = plane(lr), plane(pms), plane(rms), plane(ufs)
        self.mvl0, self.mvl1 = as_dev(mvl0), as_dev(mvl1)
        self.noise = gumbel_uniform
        self.fea = None
        self.seconds = 0.0
        # use_graph: the cached-path forward (frames >= 1: identical shapes every step, ~700 kernel launches of a few
        # microseconds each at one clip) is captured once into a HIP graph and replayed from static buffers
        self.use_graph = use_graph
        self._graph = None

    def _mvs(self, mvl: torch.Tensor, i: int) -> torch.Tensor:
        m = torch.zeros((NFRAMES, 2, self.Hp, self.Wp), dtype=torch.float32, device=self.dev)
        m[:, :, :self.H, :self.W] = mv2mvs(mvl[max(1, i) if self.T > 1 else 0])
        return modify_mv_for_end_frames(i, m.unsqueeze(0), self.T)

    def step(self, i: int) -> torch.Tensor:
        """Super-resolve frame i (frames must be visited in order: the feature cache slides by one frame per step)."""
        if (self.fea is None) != (i == 0):
            raise ValueError("StreamingSR.step: frames must be processed in order, starting at 0")
        o = generate_input_index(i, NFRAMES, self.T - 1).to(self.dev)
        po = o.clamp_min(1) if self.T > 1 else o                 # priors of frame 0 come from entry 1 (:36,53,66)
        win = lambda t, idx: t.index_select(0, idx)[None, :, None]            # [1,7,1,H,W]
        x, p, r, u = win(self.lr, o), win(self.pms, po), win(self.rms, po), win(self.ufs, po)
        m0, m1 = self._mvs(self.mvl0, i), self._mvs(self.mvl1, i)
        noise = None if self.noise is None else self.noise[i]
        if self.use_graph and i >= 1:
            return self._graph_step(x, m0, m1, p, r, u, noise)
        torch.cuda.synchronize(self.dev)
        t0 = time.perf_counter()
        with torch.no_grad():
            out, self.fea = self.model(x, m0, m1, p, r, u, self.fea, gumbel_uniform=noise)
        torch.cuda.synchronize(self.dev)
        self.seconds += time.perf_counter() - t0
        return out[..., :4 * self.H, :4 * self.W]

    def _graph_step(self, x, m0, m1, p, r, u, noise):
        """Frames >= 1 from a HIP graph of the cached-path forward (cdfo_amd.graph.CapturedForward: device-side Philox key
        refreshed per replay, no range guard inside the graph -- the eager first frame of the sequence ran with it)."""
        fea = self.fea.contiguous()
        if self._graph is None:
            from .graph import CapturedForward
            self._graph = CapturedForward(self.model, x, m0, m1, p, r, u, fea, noise, check_range=False)
        torch.cuda.synchronize(self.dev)
        t0 = time.perf_counter()
        out, new_fea = self._graph(x, m0, m1, p, r, u, fea, noise)
        torch.cuda.synchronize(self.dev)
        self.seconds += time.perf_counter() - t0
        self.fea = new_fea.clone()                                   # the graph's output buffers are overwritten next step
        return out[..., :4 * self.H, :4 * self.W].clone()

    def _inputs(self, i: int):
        o = generate_input_index(i, NFRAMES, self.T - 1).to(self.dev)
        po = o.clamp_min(1) if self.T > 1 else o
        win = lambda t, idx: t.index_select(0, idx)[None, :, None]            # [1,7,1,H,W]
        return (win(self.lr, o), self._mvs(self.mvl0, i), self._mvs(self.mvl1, i), win(self.pms, po), win(self.rms, po),
                win(self.ufs, po))

    def run_pipelined(self) -> List[torch.Tensor]:
        """All frames in order, software-pipelined over two streams: frame i + 1's front half (the new frame's feature
        extraction, the six neighbour pipelines, temporal fusion -- launches on one to three frames that leave much of the GPU
        idle at one clip) runs beside frame i's reconstruction trunk (matrix-core bound).  Frame i + 1 only needs frame i's
        feature cache, which its front half produced.  Same kernels, same arithmetic, same noise keys in the same order as
        `run()`: the outputs are identical.  `self.fps` afterwards = frames / wall time of the whole loop (a THROUGHPUT; the
        per-frame latency is that of `run()`).  The fp16 range guard of `forward` (one host readback per call) is not part
        of this schedule, as in the graph mode."""
        if not hasattr(self.model, "forward_front"):
            raise NotImplementedError("run_pipelined needs a model with forward_front / forward_back (CVSR_V8)")
        front, back = torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev)
        cur = torch.cuda.current_stream(self.dev)
        front.wait_stream(cur)
        back.wait_stream(cur)
        outs, fea = [], None
        torch.cuda.synchronize(self.dev)
        t0 = time.perf_counter()
        with torch.no_grad():
            for i in range(self.T):
                with torch.cuda.stream(front):
                    x, m0, m1, p, r, u = self._inputs(i)
                    noise = None if self.noise is None else self.noise[i]
                    state, fea = self.model.forward_front(x, m0, m1, p, r, u, fea, gumbel_uniform=noise)
                    ready = torch.cuda.Event()
                    ready.record(front)
                with torch.cuda.stream(back):
                    back.wait_event(ready)
                    for t in state:
                        t.record_stream(back)
                    out = self.model.forward_back(state)
                    outs.append(out[..., :4 * self.H, :4 * self.W])
        torch.cuda.synchronize(self.dev)
        self.seconds = time.perf_counter() - t0
        self.fea = fea
        cur.wait_stream(back)
        return outs

    def run(self) -> List[torch.Tensor]:
        """All frames in order; ``self.fps`` afterwards = frames / summed forward time (test_LD_22_FPS.py:192)."""
        self.fea, self.seconds = None, 0.0
        return [self.step(i) for i in range(self.T)]

    @property
    def fps(self) -> float:
        return self.T / self.seconds if self.seconds > 0 else float("nan")
