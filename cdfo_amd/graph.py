"""Whole-forward HIP graphs for fixed shapes: ``CVSR_V8.capture(...)`` -> :class:`CapturedForward`.

One inference forward is ~270 kernel launches at eight clips and ~700 at one clip (three HIP streams); at small shapes the
inter-kernel gaps and the host-side launch cost are a visible share of the step (c2: 15.1 ms for ~270 launches).  A captured
forward replays the same launches, on the same streams' fork / join structure, from one ``hipGraphLaunch``.

What makes a capture of THIS model correct (the pieces were built for ``StreamingSR`` in round 3):

* the Gumbel noise of ``LLongRangAttention`` is drawn per call in the reference (arch/SIDECVSR_our.py:2169).  A captured mask
  kernel therefore reads its Philox key from a device word (``cdfo_rdab_prep_rng_dev``) that ``refresh_noise_key`` rewrites
  before every replay -- torch's default generator advances once per forward exactly as in the eager loop, so a run under
  ``torch.manual_seed`` produces the same frames eagerly and from the graph;
* the fp16 range guard: ``capture`` runs ONE eager, guarded forward on the example inputs first and refuses to capture a workload
  whose activations leave the fp16 window (use ``precision="bf16x3"``).  Round 5: the two probes of the guard (trunk input, trunk
  output) are graph nodes too -- zeroed, filled by the probe kernels, copied to pinned host memory behind the last kernel -- and
  ``replay()`` reads them back: a replay on operands that leave the window is repeated eagerly in bf16x3 INTO the graph's output
  buffers, exactly what the eager forward does.  ``replay(sync=False)`` returns without waiting (pipelined callers); the check
  then runs at the start of the next replay, in ``finish_range_guard()`` or when ``last_range`` is read;
* the caller sees graph-owned buffers: ``inputs`` (write the next batch there, or pass tensors to ``__call__`` and they are
  copied in) and the returned ``(out, L1_fea)``, which the NEXT replay overwrites -- ``clone()`` what must outlive it.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch


class CapturedForward:
    """A HIP graph of ``model(x, mvs0, mvs1, pms, rms, ufs, pre_L1_fea, gumbel_uniform=...)`` at the example's shapes."""

    def __init__(self, model, x, mvs0, mvs1, pms, rms, ufs, pre_L1_fea=None, gumbel_uniform: Optional[Sequence] = None,
                 check_range: bool = True):
        if not x.is_cuda:
            raise NotImplementedError("CapturedForward (HIP): CUDA tensors expected")
        # Graphs are an inference tool: warm-up, capture and every replay are the torch.no_grad() schedule whatever the
        # caller's grad mode is (the autograd path cannot be captured -- CVSR_V8.forward says why).
        self.model, self.dev = model, x.device
        self.draws_noise = gumbel_uniform is None and getattr(model, "gumbel_uniform", None) is None
        fixed = [x, mvs0, mvs1, pms, rms, ufs]
        self._nnoise = 0 if gumbel_uniform is None else len(gumbel_uniform)
        ins: List[Optional[torch.Tensor]] = [None if t is None else t.clone() for t in fixed]
        ins.append(None if pre_L1_fea is None else pre_L1_fea.contiguous().clone())
        ins += [] if gumbel_uniform is None else [u.clone() for u in gumbel_uniform]
        self.inputs = ins

        def call(guard: bool):
            old = getattr(model, "range_guard", False)
            model.range_guard = guard
            try:
                return model(ins[0], ins[1], ins[2], ins[3], ins[4], ins[5], ins[6],
                             gumbel_uniform=None if self._nnoise == 0 else ins[7:])
            finally:
                model.range_guard = old

        # Warm-up and capture must not consume the generator: replay k then draws the key the k-th eager forward would have
        rng_state = torch.cuda.get_rng_state(self.dev)
        cur = torch.cuda.current_stream(self.dev)
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(cur)
        model.last_range = None
        with torch.cuda.stream(side), torch.no_grad():          # warm-up on a side stream, as graph capture requires
            call(guard=check_range)
            if check_range and (getattr(model, "last_range", None) or {}).get("fallback"):
                raise RuntimeError("CapturedForward: the example's activations leave the fp16 range (the eager forward fell back to "
                                   "bf16x3); a captured forward has no range guard -- set model.precision = 'bf16x3' and capture again")
        cur.wait_stream(side)
        # CVSR_V8 draws its noise in-kernel from a device-side Philox key; CVSR_V7 draws with torch.rand, whose graph-safe generator
        # state torch advances per replay by itself
        self._refresh = getattr(model, "refresh_noise_key", None) if self.draws_noise else None
        if self._refresh is not None:
            self._refresh(self.dev)
        # the guard's probes inside the graph (CVSR_V8.forward's capturing branch reads model._graph_probe)
        self.guarded = bool(check_range and getattr(model, "precision", None) == "fp16x2" and getattr(model, "range_guard", False)
                            and hasattr(model, "_range_fallback"))
        self._pending, self._seed, self.last_range_seen = False, None, None
        if self.guarded:
            self._probe = torch.zeros(4, dtype=torch.int32, device=self.dev)
            self._probe_host = torch.zeros(4, dtype=torch.int32).pin_memory()
            self._ev = torch.cuda.Event()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: only THIS thread's calls are checked against the capture -- a process group's watchdog thread (event queries
        # while bench.py's ranks capture) must not invalidate it
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"), torch.no_grad():
            if self.guarded:
                model._graph_probe = self._probe
            try:
                self.out, self.L1_fea = call(guard=True if self.guarded else False)
            finally:
                if self.guarded:
                    model._graph_probe = None
            if self.guarded:
                self._probe_host.copy_(self._probe, non_blocking=True)
        torch.cuda.set_rng_state(rng_state, self.dev)

    def load(self, x=None, mvs0=None, mvs1=None, pms=None, rms=None, ufs=None, pre_L1_fea=None, gumbel_uniform=None) -> None:
        """Copy new operands into the graph's input buffers (None = keep what is there; tensors that ARE the buffers are skipped)."""
        new = [x, mvs0, mvs1, pms, rms, ufs, pre_L1_fea] + ([None] * self._nnoise if gumbel_uniform is None else list(gumbel_uniform))
        if len(new) != len(self.inputs):
            raise ValueError(f"captured with {self._nnoise} noise tensors, got {len(new) - 7}")
        for dst, src in zip(self.inputs, new):
            if src is None or src is dst:
                continue
            if dst is None:
                raise ValueError("an operand that was None at capture time cannot be supplied at replay")
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"captured for shape {tuple(dst.shape)}, got {tuple(src.shape)}")
            dst.copy_(src)

    def replay(self, sync: bool = True):
        """One forward on whatever the input buffers hold.  Returns the graph-owned (out, L1_fea).  With the range guard in the
        graph (fp16x2): sync=True waits for the forward and settles the guard before returning -- what is returned is final;
        sync=False returns at once, the guard of THIS replay is settled by the next replay / finish_range_guard() / last_range."""
        self.finish_range_guard()
        if self._refresh is not None:
            self._seed = self._refresh(self.dev)
        self.graph.replay()
        if self.guarded:
            self._ev.record()
            self._pending = True
            if sync:
                self.finish_range_guard()
        return self.out, self.L1_fea

    def finish_range_guard(self) -> None:
        """Read the probes of the last replay (waits for it); operands outside the fp16 window, or a non-finite trunk result: the
        forward is repeated eagerly in bf16x3 on the graph's input buffers (same noise key) into the graph's output buffers."""
        if not getattr(self, "_pending", False):
            return
        self._pending = False
        self._ev.synchronize()
        h = self._probe_host
        amax = h[0:1].view(torch.float32).item()
        nonfinite = bool(h[1].item()) or bool(h[3].item())
        m = self.model
        self.last_range_seen = {"trunk_input_amax": amax, "nonfinite": nonfinite, "fallback": False}
        lo, hi = m.FP16_WINDOW
        if not nonfinite and (amax == 0.0 or lo <= amax <= hi):
            return
        ins = self.inputs
        m._last_range = self.last_range_seen
        args = (ins[0], ins[1], ins[2], ins[3], ins[4], ins[5], ins[6], None if self._nnoise == 0 else ins[7:],
                self._seed if self._seed is not None else 0)
        with torch.cuda.device(self.dev):
            m._range_fallback(args, self.out, self.L1_fea)

    @property
    def last_range(self):
        """What the in-graph range guard saw in the most recent replay (settles it first); None without a guard."""
        self.finish_range_guard()
        return self.last_range_seen

    def __call__(self, x=None, mvs0=None, mvs1=None, pms=None, rms=None, ufs=None, pre_L1_fea=None, gumbel_uniform=None):
        self.load(x, mvs0, mvs1, pms, rms, ufs, pre_L1_fea, gumbel_uniform)
        return self.replay()


class PipelinedForward:
    """Two captured forwards of the same model and shapes, used alternately on one stream.  Each has its OWN memory pool: with a
    shared pool the second graph's outputs land in memory the first graph uses for intermediates, and a result would be overwritten
    by the next forward before it has been checked and consumed (measured: tests/probe_pipe.py) -- twice the activation memory of one
    forward (2 x ~20 GB at eight clips of 272x480) is the price.

    ``CapturedForward.replay()`` either waits for the forward before it returns (the range guard's probes are a host readback), or
    leaves the check to the next replay -- which then waits before it launches.  Either way the GPU idles for the launch latency of a
    ~700-node graph once per forward (~1.4 ms at eight clips).  With two graphs, forward k + 1 is launched FIRST and forward k's probes
    are read afterwards; k's operands and results live in the other graph's buffers, so a forward the guard rejects is still repeated
    in bf16x3 into the tensors ``submit`` returned for it.

    ``submit(...)`` returns graph-owned ``(out, L1_fea)`` of the forward it launched; they are FINAL once the next ``submit`` (or
    ``drain()``) has returned, and are overwritten by the submit after that."""

    def __init__(self, model, x, mvs0, mvs1, pms, rms, ufs, pre_L1_fea=None, gumbel_uniform: Optional[Sequence] = None,
                 check_range: bool = True):
        a = CapturedForward(model, x, mvs0, mvs1, pms, rms, ufs, pre_L1_fea, gumbel_uniform, check_range)
        b = CapturedForward(model, x, mvs0, mvs1, pms, rms, ufs, pre_L1_fea, gumbel_uniform, check_range)
        self.caps, self.k = (a, b), 0

    @property
    def guarded(self) -> bool:
        return self.caps[0].guarded

    def submit(self, x=None, mvs0=None, mvs1=None, pms=None, rms=None, ufs=None, pre_L1_fea=None, gumbel_uniform=None):
        cap, prev = self.caps[self.k & 1], self.caps[(self.k + 1) & 1]
        self.k += 1
        cap.load(x, mvs0, mvs1, pms, rms, ufs, pre_L1_fea, gumbel_uniform)
        res = cap.replay(sync=False)          # (its own previous forward was settled two submits ago)
        prev.finish_range_guard()             # the forward before this one: its probes are read with this one already queued
        return res

    def drain(self) -> None:
        """Settle the range guard of every forward submitted so far (waits for them)."""
        for c in self.caps:
            c.finish_range_guard()

    @property
    def last_range(self):
        """The guard's verdict on the most recently submitted forward (settles it first)."""
        return self.caps[(self.k + 1) & 1].last_range if self.k else None
