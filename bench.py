#!/usr/bin/env python3
"""Headline benchmark: x4 SR frames/s of the CVSR_V8 forward on synthetic JCT-VC ClassB-shape clips.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 8] [--height 270 --width 480] [--no-cpu-baseline]

A "step" is one forward of the hot path over one batch of clips already resident in HBM: 8 clips per GPU of
7 x 1 x 272 x 480 luma (270 rows zero-padded to 272) + MV / residual / partition / unfiltered priors ->
8 HR frames of 1088 x 1920 (crop to 1080).  For N > 1 there is one process per GPU (the driver starts them with torchrun;
a bare `python bench.py --gpus N` starts them itself, see `launch_ranks`); clips are sharded by batch (weak scaling:
8 clips per GPU), there is no data-path collective, and ONE all_gather over RCCL carries the per-rank metrics.
Rank 0 prints one JSON line.

What is timed (`value`): K forwards of the DEFAULT path -- Gumbel noise drawn per call like the reference does
(arch/SIDECVSR_our.py:2169), here inside the mask kernel -- with no per-launch instrumentation.  The per-kernel roofline
numbers come from a second pass over the same K steps with one HIP-event pair per launch.  Parity (gated: the run fails
above the 1e-3 bound) is checked on clip 0 of the timed batch against the CPU oracle, replaying the noise that forward
drew; the same oracle run is the `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FLOPs per clip-forward, counted on the reference itself: F = P*(73.55e6 + 1536*(W+H))
EXACT_FLOPS = {(272, 480): 9_753_744_609_072, (120, 240): 2_134_302_625_536, (64, 64): 302_119_013_712,
               (544, 960): 39_617_571_644_688}
PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0, "fp16x2": 2500.0}   # /opt/skills/guides/MI355X_MICROARCH.md dense MFMA peaks
MFMA_PASSES = {"f32": 1, "bf16x3": 3, "bf16": 1, "fp16x2": 2}   # tiled kernel; the Block_ kernels (ws, ring) are 1-pass fp16                    # bf16 MFMA MACs issued per algorithmic MAC
KID_NAMES = ["conv3x3_wide", "conv3x3_narrow", "conv1x1", "conv3x3_s2", "stem", "layernorm", "dwconv", "flow_warp",
             "resample", "scale", "conv_last", "small_conv", "spatial_gate", "chan_sum", "gram", "fold", "rdab_prep",
             "colconv9", "attn_row", "attn_col", "attn_win", "layout", "pack", "dcn", "conv3x3_ws", "conv3x3_ring", "conv3x3_ring4", "conv3x3_ws_res", "dcn_bwd", "conv3x3_wino"]
# kernel families on the 16-bit matrix cores -> (kernel symbol in the rocprofv3 stats, MFMA passes per algorithmic MAC)
# conv3x3_wide (the tiled kernel) runs the --precision mode's passes: fp16x2 = 2 (3 for the split-bf16 feature cache)
# conv3x3_wino (Block_.body[0] since round 5): Winograd F(2,3) along x issues 12 MFMA MACs per 18 algorithmic ones
MFMA16 = {"conv3x3_wide": ("conv3x3_mma16_kernel", None), "conv3x3_ws": ("conv3x3_c64_wsq_kernel", 1),
          "conv3x3_wino": ("conv3x3_c64_wino_kernel", 0.667),
          "conv3x3_ws_res": ("conv3x3_c64_ws_kernel<0, true, 8>", 1),
          "conv3x3_ring": ("conv3x3_ring_kernel<false", 1), "conv3x3_ring4": ("conv3x3_ring_split_kernel<true", 1)}


# HBM-bound kernel families of the event profiler -> substrings of the kernel symbols they launch (rocprofv3 names); used to
# pair a family's algorithmic bytes with the PMC traffic of the same launches (tools/pmc_summary.py --families)
HBM_FAMILIES = {"conv1x1": ["conv1x1_stream_kernel", "conv1x1_bf16x3_kernel"], "dwconv": ["qkv_dw_kernel", "qkv_dw2_kernel", "dwconv3x3_kernel"],
                "attn_row": ["seq_attn_mfma_kernel<0"], "attn_col": ["seq_attn_mfma_kernel<1"], "attn_win": ["seq_attn_mfma_kernel<2"],
                "rdab_prep": ["rdab_prep_kernel"], "resample": ["block_pro_kernel", "resample2_kernel"], "stem": ["stem_conv"],
                "flow_warp": ["flow_warp_kernel"], "colconv9": ["colconv9_kernel"], "chan_sum": ["chan_sum_partial_kernel"],
                "gram": ["gram_partial_kernel"], "layout": ["swap_outer_kernel", "to_cp16_kernel", "nchw_to_nhwc", "nhwc_to_nchw"],
                "scale": ["scale_channels_kernel"], "small_conv": ["small_conv3x3_kernel", "udsa_head_kernel"],
                "spatial_gate": ["spatial_gate16_kernel"], "conv_last": ["conv_last"], "layernorm": ["layernorm64"],
                "conv3x3_narrow": ["conv3x3_c64_n16_kernel"]}
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
# kernel family -> the cdfo_amd/csrc files its kernels live in.  tools/pmc_summary.py stores their SHA-256 next to the PMC traffic it
# writes into profiles/traffic_*.json; a replayed figure whose sources have changed since is reported with "traffic_stale": true
FAMILY_SOURCES = {"conv3x3_wide": ["conv3x3_bf16.hip"], "conv3x3_ws": ["conv3x3_ws.hip"], "conv3x3_wino": ["conv3x3_wino.hip"],
                  "conv3x3_ws_res": ["conv3x3_ws.hip"], "conv3x3_ring": ["conv3x3_ring.hip"], "conv3x3_ring4": ["conv3x3_ring.hip"],
                  "conv1x1": ["conv1x1_stream.hip", "conv1x1_bf16x3.hip"], "dwconv": ["qkv_dw.hip", "pointwise.hip"],
                  "attn_row": ["attention.hip"], "attn_col": ["attention.hip"], "attn_win": ["attention.hip"], "rdab_prep": ["attention.hip"],
                  "colconv9": ["attention.hip"], "resample": ["block_pro.hip", "pointwise.hip"], "stem": ["pointwise.hip"],
                  "flow_warp": ["pointwise.hip"], "chan_sum": ["stats.hip"], "gram": ["stats.hip"], "layout": ["layout.hip", "conv3x3_ws.hip"],
                  "scale": ["pointwise.hip"], "small_conv": ["smallconv.hip"], "spatial_gate": ["smallconv.hip"], "conv_last": ["pointwise.hip"],
                  "layernorm": ["pointwise.hip"], "conv3x3_narrow": ["conv3x3_n16.hip"], "dcn": ["dcn_win.hip", "dcn_fast.hip", "dcn.hip"]}


def source_hashes(family):
    """{file: sha256} of a kernel family's sources in THIS tree (no git needed: the GPU box has none)."""
    import hashlib
    out = {}
    for f in FAMILY_SOURCES.get(family, []):
        q = os.path.join(ROOT, "cdfo_amd", "csrc", f)
        out[f] = hashlib.sha256(open(q, "rb").read()).hexdigest() if os.path.exists(q) else None
    return out


def traffic_stale(family, recorded):
    """True when the PMC figure of `family` was collected on other kernel sources than this tree's (or carries no hashes at all:
    files of rounds 1-4), False when every source file still hashes to what tools/pmc_summary.py recorded."""
    if not recorded:
        return True
    return any(recorded.get(f) != h for f, h in source_hashes(family).items())


def load_traffic(precision):
    """profiles/traffic_rNN.json of the newest round: HBM bytes per launch from rocprofv3 PMC passes (separate FETCH_SIZE /
    WRITE_SIZE runs of this script, tools/profile_run.sh), with the commit they were taken at.  Replayed, not measured here."""
    for r in range(9, 0, -1):
        q = os.path.join(ROOT, "profiles", f"traffic_r{r:02d}.json")
        if not os.path.exists(q):
            continue
        try:
            tj = json.load(open(q))
        except Exception:
            return {}
        if tj.get("precision") != precision:
            return {}
        fams = dict(tj.get("families") or {})
        if "kernel" in tj and tj["kernel"] not in fams:        # rounds 1-2: one kernel per file
            fams[tj["kernel"]] = {"hbm_bytes_per_launch": tj.get("hbm_bytes_per_launch")}
        return {"file": os.path.relpath(q, ROOT), "commit": tj.get("commit"), "families": fams}
    return {}


def flops_per_clip(H, W):
    return EXACT_FLOPS.get((H, W), H * W * (73.55e6 + 1536.0 * (W + H)))


PARITY_BOUND = 1e-3      # north_star: max-abs vs the fp32 reference path


def launch_command(n: int, argv, port: int):
    """The torchrun command line that starts the N ranks of this script (one process per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without torchrun: start the N ranks as children BEFORE this process touches the GPU
    (a process that has initialised the GPU must never exec or fork GPU work), relay their output (rank 0 prints the JSON
    line) and return their exit status."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(launch_command(n, argv, port), env=env).returncode


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU")
    ap.add_argument("--height", type=int, default=270)
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--launch", default="eager", choices=["graph", "eager"],
                    help="how the timed step is launched: eager (default) = one host launch per kernel; graph = two alternating captured "
                         "forwards (CVSR_V8.capture_pipelined): per step the operands are copied into a graph's buffers, a fresh noise key is "
                         "written, the HIP graph is replayed and the previous step's range-guard probes are read back.  Same kernels, "
                         "bit-identical outputs; at eight clips the two are within +-0.5 % of each other (the default line reports the replay "
                         "as the extra `hip_graph_replay`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the c1 parity spot-check (profiling runs: keeps every launch at the workload's size)")
    ap.add_argument("--precision", default="fp16x2", choices=["f32", "bf16x3", "bf16", "fp16x2"])
    ap.add_argument("--streaming", action="store_true", help="time the cached-feature path (pre_L1_fea given: one new frame per clip, test_LD_22_FPS.py:183-189) instead of the fresh path")
    ap.add_argument("--neighbour-streams", type=int, default=0, help="issue the six neighbour-frame pipelines on this many HIP streams (0 = the model's choice: 3 at 8 clips)")
    ap.add_argument("--breakdown", type=str, default="", help="write the per-kernel-family event timings to this file")
    ap.add_argument("--injected-noise", action="store_true", help="time the forward with pre-made Gumbel noise tensors (the parity tests' path) instead of the default in-kernel draws")
    ap.add_argument("--no-full-size-parity", action="store_true", help="N > 1: skip rank 0's clip-0 check at the timed size (every rank still checks c1)")
    ap.add_argument("--no-extra-modes", action="store_true", help="skip the extra measurements (bf16x3 mode, injected-noise path, DCN / V7 / streaming / training lines)")
    ap.add_argument("--process-group", action="store_true", help="N = 1: still initialise the process group (RCCL, world size 1) so that the barriers and the metric all_gather of the N > 1 path run over RCCL; automatic under torchrun")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: become the launcher.  Nothing above has initialised the GPU in this process.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU (torchrun --nproc-per-node "
                         f"{args.gpus}), or run `python bench.py --gpus {args.gpus}` outside torchrun and let it start them")
    # CDFO_BENCH_BACKEND=gloo rehearses the N > 1 control flow on a box with fewer GPUs than ranks (ranks then share GPUs and
    # the one all_gather goes over gloo on the CPU); the measured configuration is "nccl" = RCCL over xGMI, one GPU per rank
    backend = os.environ.get("CDFO_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # The process group exists whenever there is a rendezvous to join: always at N > 1, and at N = 1 under torchrun
    # (`torchrun --nproc-per-node 1 bench.py`) or with --process-group -- the one-rank run then takes the same barriers and
    # the same all_gather over RCCL as the multi-GPU run (the way to prove that line on a one-GPU box).
    import torch.distributed as dist
    under_torchrun = all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"))
    use_pg = world > 1 or under_torchrun or args.process_group
    if use_pg:
        if not under_torchrun:                               # --process-group outside torchrun: a one-rank rendezvous on localhost
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1")
        # explicit collective timeout: at N > 1 rank 0 runs the CPU oracle on one full-size clip (tens of seconds; minutes on a loaded
        # host) while the other ranks already wait in the metric all_gather -- the wait must never be mistaken for a hang
        import datetime
        pg_timeout = datetime.timedelta(seconds=int(os.environ.get("CDFO_BENCH_PG_TIMEOUT_S", "3600")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)

    from arch.SIDECVSR_our import CVSR_V8
    from cdfo_amd import _lib
    from oracle.cvsr_v8_ref import cvsr_v8_forward, make_inputs, make_state_dict, psnr_y

    lib = _lib.lib()
    sd = make_state_dict(0, perturb=False)                # random init of the reference architecture
    model = CVSR_V8()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    model.precision = args.precision
    model.neighbour_streams = args.neighbour_streams

    if world > 1:                                          # N ranks share the host's cores for CPU-side checks
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))

    # ---- workload: B clips, H padded to a multiple of 8 (test_LD_37.py:24-26 semantics)
    B = args.batch
    Hp = (args.height + 7) // 8 * 8
    Wp = (args.width + 7) // 8 * 8
    inp = make_inputs(B, Hp, Wp, 1002 + rank, pad_rows=Hp - args.height)
    d = {k: v.to(dev) for k, v in inp.items() if k != "gumbel_u"}
    injected = [u.to(dev) for u in inp["gumbel_u"]] if args.injected_noise else None

    pre = None
    if args.streaming:
        with torch.no_grad():
            _, pre = model(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=injected)

    graph = {"cap": None}     # the captured forward of the timed configuration (--launch graph), built behind the parity forward

    def step(noise=injected, eager=False):
        with torch.no_grad():
            if graph["cap"] is not None and not eager and noise is injected:
                # copies the operands into the buffers of one of two alternating graphs, refreshes the noise key, replays it, THEN
                # reads the previous step's range-guard probes back (PipelinedForward: no graph-launch latency between steps)
                return graph["cap"].submit(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], pre, injected)
            return model(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], pre, gumbel_uniform=noise)

    def barrier():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(nsteps, noise=injected, eager=False):
        barrier()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            o, _ = step(noise, eager)
        if graph["cap"] is not None and not eager:
            graph["cap"].drain()              # the last step's range guard is settled inside the timed region
        barrier()
        return time.perf_counter() - t0, o

    # ---- parity against the CPU oracle, on the path that is timed.  N = 1: clip 0 of the timed batch at the timed size
    # (the oracle forward on one 272x480 clip is also the cpu_baseline measurement); N > 1: every rank checks the
    # reference's own CPU-runnable config (c1: 64x64, B=1) -- N oracle forwards at full size would share one host -- and
    # rank 0 ALSO checks clip 0 of its timed batch at the timed size (the grouped, two-stream, full-size schedule that c1's
    # one-frame-per-stream small-shape kernels do not exercise) while the other ranks wait at the barrier.
    max_abs, psnr, cpu = float("nan"), float("nan"), None
    max_abs_full, ref_out_full, cap_full, pending_full = float("nan"), None, None, None
    parity_cfg = "skipped (--no-parity)"
    if not args.no_parity:
        full = not args.streaming and (world == 1 or (rank == 0 and not args.no_full_size_parity))
        if full:
            cap = []
            model.capture_noise = None if args.injected_noise else cap
            got, _ = step()
            model.capture_noise = None
            cap_full = injected if args.injected_noise else cap
            # The CPU oracle on this clip (tens of seconds of host work) runs AFTER the timed region: a GPU that sat idle through
            # it clocks down, and the first timed steps after a one-step warm-up would be measured on a cold device
            pending_full = (got[0:1].cpu(), [u[0:1].cpu() for u in cap_full],
                            {k: (v[0:1].cpu() if v is not None else None) for k, v in d.items()})
            parity_cfg = f"clip 0 of the timed batch ({Hp}x{Wp}, noise as drawn by the timed path) vs CPU oracle"
            if world > 1 or args.no_extra_modes or args.precision == "bf16":
                cap_full = None                         # only the bf16_plain extra replays this noise
            del got
        if world > 1 or args.streaming:
            c1 = make_inputs(1, 64, 64, 1000)
            cap = []
            model.capture_noise = cap
            with torch.no_grad():
                g0, _ = model(c1["x"].to(dev), None, c1["mvs1"].to(dev), c1["pms"].to(dev), c1["rms"].to(dev), c1["ufs"].to(dev))
                model.capture_noise = None
                ref_out, _ = cvsr_v8_forward(sd, c1["x"], None, c1["mvs1"], c1["pms"], c1["rms"], c1["ufs"], None,
                                             [u.cpu() for u in cap])
            g0 = g0.cpu()
            max_abs = (g0 - ref_out).abs().max().item()
            psnr = psnr_y(g0, ref_out)
            small = "c1 64x64 B=1 (noise as drawn by the default path) vs CPU oracle, on every rank"
            if args.streaming:
                parity_cfg = small + " -- small-config only (the streaming path's timed size is not checked here)"
            elif full:
                parity_cfg = small + f"; rank 0 also: clip 0 of its timed batch at {Hp}x{Wp} (the grouped, two-stream, full-size schedule)"
            else:
                parity_cfg = small + " -- small-config only (--no-full-size-parity)"

    # ---- --launch graph: the forward of the timed configuration captured once (cdfo_amd/graph.py); a replay must BE the eager forward:
    # both run once under the same generator seed and their images are compared bit for bit before anything is timed
    launch_info = {"mode": "eager"}
    if args.launch == "graph" and args.precision != "f32":
        try:
            with torch.no_grad():
                cap_fwd = model.capture_pipelined(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], pre, gumbel_uniform=injected)
            torch.manual_seed(4242 + rank)
            e_out, _ = step(eager=True)
            e_out = e_out.clone()
            graph["cap"] = cap_fwd
            same = True
            for _ in range(2):                                # both graphs of the pair
                torch.manual_seed(4242 + rank)
                g_out, _ = step()
                cap_fwd.drain()
                torch.cuda.synchronize()
                same = same and bool(torch.equal(e_out, g_out))
            del e_out, g_out
            if not same:
                raise SystemExit("bench.py: the replayed HIP graph and the eager forward differ under the same seed")
            launch_info = {"mode": "hip_graph", "replay_equals_eager_bitwise": True, "range_guard_in_graph": bool(cap_fwd.guarded),
                           "per_step": "operands copied into the buffers of one of two alternating graphs (cdfo_amd.graph.PipelinedForward), "
                                       "fresh Philox key, hipGraphLaunch, then the PREVIOUS step's fp16 range-guard probes are read back; "
                                       "the last step's inside the timed region"}
        except SystemExit:
            raise
        except Exception as e:                                # no graph on this box / shape: the eager path is timed, and the line says so
            graph["cap"] = None
            launch_info = {"mode": "eager", "graph_capture_failed": repr(e)[:300]}

    for _ in range(args.warmup):
        step()
    elapsed, out = timed(args.steps)                      # the headline: no per-launch instrumentation
    range_info = (graph["cap"].last_range if graph["cap"] is not None and graph["cap"].guarded else getattr(model, "last_range", None))
    if graph["cap"] is not None:                          # the same steps launched kernel by kernel, for the record
        step(eager=True)
        elapsed_eager, _ = timed(args.steps, eager=True)
        launch_info["eager_ms_per_step"] = round(1e3 * elapsed_eager / args.steps, 3)
        launch_info["eager_frames_per_s"] = round(B * args.steps / elapsed_eager, 3)

    # ---- second pass over the same steps with one HIP-event pair per launch: per-kernel durations for the roofline
    nk = lib.cdfo_prof_kid_count()
    cap_records = 4000 * max(1, args.steps)
    _lib.check(lib.cdfo_prof_begin(cap_records), "cdfo_prof_begin")
    elapsed_prof, _ = timed(args.steps, eager=True)       # (events are recorded around host launches: the eager path)
    launches = (C.c_int * nk)()
    ms = (C.c_double * nk)()
    fl = (C.c_double * nk)()
    by = (C.c_double * nk)()
    nrec = lib.cdfo_prof_end(launches, ms, fl, by, nk)
    if nrec < 0:
        raise SystemExit(f"cdfo_prof_end failed: {nrec}")
    if nrec >= cap_records:
        raise SystemExit(f"the event profiler ran out of records ({nrec} >= {cap_records}): per-kernel numbers would be truncated")

    # ---- extra measurements (N = 1 only, outside the headline): the fp32-grade mode and the injected-noise path
    extra = {}
    if world == 1 and not args.no_extra_modes and not args.streaming:
        nst = max(2, min(args.steps, 5))
        if graph["cap"] is None and args.precision != "f32":
            # the same batch replayed from two alternating captured forwards (--launch graph's step), outside the headline
            pipe = None
            try:
                with torch.no_grad():
                    pipe = model.capture_pipelined(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], pre, gumbel_uniform=injected)
                torch.manual_seed(4242 + rank)
                e_out, _ = step(eager=True)
                e_out = e_out.clone()
                graph["cap"] = pipe
                same = True
                for _ in range(2):
                    torch.manual_seed(4242 + rank)
                    g_out, _ = step()
                    pipe.drain()
                    torch.cuda.synchronize()
                    same = same and bool(torch.equal(e_out, g_out))
                del e_out, g_out
                step()
                tg, _ = timed(nst)
                extra["hip_graph_replay"] = {"frames_per_s": round(B * nst / tg, 3), "ms_per_step": round(1e3 * tg / nst, 3), "steps": nst,
                                             "replay_equals_eager_bitwise": same, "range_guard_in_graph": bool(pipe.guarded),
                                             "note": "cdfo_amd.graph.PipelinedForward: operands copied in, fresh noise key, hipGraphLaunch, the "
                                                     "previous step's range-guard probes read back"}
            except Exception as e:
                extra["hip_graph_replay"] = {"error": repr(e)[:200]}
            finally:
                graph["cap"] = None
                del pipe
                torch.cuda.empty_cache()
        other_noise = None if args.injected_noise else [u.to(dev) for u in inp["gumbel_u"]]
        step(other_noise, eager=True)
        t, _ = timed(nst, other_noise, eager=True)
        extra["injected_noise_path" if not args.injected_noise else "default_noise_path"] = {
            "frames_per_s": round(B * nst / t, 3), "ms_per_step": round(1e3 * t / nst, 3), "steps": nst}
        del other_noise
        if args.precision != "bf16x3":
            model.precision = "bf16x3"
            step(eager=True)
            t, _ = timed(nst, eager=True)
            extra["bf16x3_fp32_grade"] = {"frames_per_s": round(B * nst / t, 3), "ms_per_step": round(1e3 * t / nst, 3),
                                          "steps": nst, "note": "split-bf16 3-pass MFMA everywhere: <= 1.2e-5 max-abs vs the fp32 reference"}
            model.precision = args.precision
        if args.precision != "bf16" and cap_full is not None and pending_full is not None:
            # BASELINE's literal c3 dtype: plain bf16 MFMA (one rounding of activations and weights, fp32 accumulate) on the same
            # batch, replaying the noise of the parity forward so that clip 0 compares with the same oracle output (below)
            model.precision = "bf16"
            gb, _ = step(cap_full, eager=True)
            t, _ = timed(nst, cap_full, eager=True)
            g0_bf16 = gb[0:1].cpu()
            extra["bf16_plain"] = {"frames_per_s": round(B * nst / t, 3), "ms_per_step": round(1e3 * t / nst, 3), "steps": nst}
            model.precision = args.precision
            del gb
        cap_full = None

    # ---- the deferred full-size parity check: CPU oracle on clip 0 (also the cpu_baseline measurement at N = 1)
    if pending_full is not None:
        g0f, noise0, clip0 = pending_full
        ref_out_full, cpu = oracle_clip(sd, clip0, noise0, Hp, Wp, time_it=(world == 1 and not args.no_cpu_baseline))
        max_abs_full = (g0f - ref_out_full).abs().max().item()
        if world == 1:
            max_abs, psnr = max_abs_full, psnr_y(g0f, ref_out_full)
        if "bf16_plain" in extra:
            eb = (g0_bf16 - ref_out_full).abs().max().item()
            extra["bf16_plain"].update({"max_abs": eb, "psnr_y_db": psnr_y(g0_bf16, ref_out_full), "within_1e-3": bool(eb <= PARITY_BOUND),
                                        "note": "precision='bf16' (BASELINE c3's dtype), noise replayed from the parity forward; outside "
                                                "the 1e-3 bound by design of the format (8-bit mantissa), which is why it is not the default"})
        pending_full = None

    # ---- the single collective: all_gather of per-rank metrics (time, checksum, parity)
    from cdfo_amd.dist import gather_metrics
    ident = device_identity(local)               # 8 numbers: which physical GPU this rank drove (duplicates fail the run below)
    allm = gather_metrics([elapsed, out.double().mean().item(), max_abs, psnr if math.isfinite(psnr) or psnr != psnr else 999.0,
                           elapsed_prof, max_abs_full, float(local), *ident], dev if backend == "nccl" else None)
    t_max = allm[:, 0].max().item()
    # parity gate: a numerically broken build must not print a valid-looking line
    parity_ok = args.no_parity or (all(math.isfinite(v) and v <= PARITY_BOUND for v in allm[:, 2].tolist())
                                   and all(v != v or v <= PARITY_BOUND for v in allm[:, 5].tolist()))
    # one GPU per rank is what "n_gpus" claims: two ranks on one physical device (a wrong LOCAL_RANK / HIP_VISIBLE_DEVICES
    # mapping) would print a valid-looking scaling line.  Only the explicit gloo rehearsal may share devices.
    distinct = len({tuple(allm[r, 7:15].tolist()) for r in range(world)})
    devices_ok = backend != "nccl" or distinct == world

    if rank == 0:
        clips_total = B * world * args.steps
        value = clips_total / t_max
        F = flops_per_clip(Hp, Wp)
        dom = max(range(nk), key=lambda k: ms[k])
        dom_avg_ms = ms[dom] / max(1, launches[dom])
        achieved = (fl[dom] / max(1, launches[dom])) / (dom_avg_ms * 1e-3) / 1e12 if dom_avg_ms > 0 else 0.0
        def fam(k):
            name = KID_NAMES[k]
            avg_ms = ms[k] / max(1, launches[k])
            ach = (fl[k] / max(1, launches[k])) / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            if name in MFMA16:
                symbol, passes = MFMA16[name]
                peak = PEAK_TFLOPS[args.precision if passes is None else "fp16x2"]
                passes = MFMA_PASSES[args.precision] if passes is None else passes
            else:
                symbol, passes, peak = name, 1, PEAK_TFLOPS["f32"]
            return {"kernel": name, "kernel_symbol": symbol, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "mfma_passes_per_mac": passes,
                    "algorithmic_flop_per_launch": round(fl[k] / max(1, launches[k]), 0),
                    "launches_per_step": launches[k] // max(1, args.steps), "avg_launch_ms": round(avg_ms, 4),
                    "share_of_gpu_time": round(ms[k] / max(1e-9, sum(ms)), 4)}
        tr = load_traffic(args.precision)
        tfam = tr.get("families", {})
        traffic = (tfam.get(KID_NAMES[dom]) or {}).get("hbm_bytes_per_launch")
        traffic_is_stale = traffic_stale(KID_NAMES[dom], (tfam.get(KID_NAMES[dom]) or {}).get("source_sha256")) if traffic else None

        def hbm_fam(k):
            name = KID_NAMES[k]
            avg_ms = ms[k] / max(1, launches[k])
            bpl = by[k] / max(1, launches[k])
            gbs = bpl / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            pmc = (tfam.get(name) or {}).get("hbm_bytes_per_launch")
            return {"kernel": name, "kernel_symbols": HBM_FAMILIES.get(name, [name]), "bound": "hbm", "achieved": round(gbs, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "algorithmic_bytes_per_launch": round(bpl), "launches_per_step": launches[k] // max(1, args.steps),
                    "avg_launch_ms": round(avg_ms, 4), "share_of_gpu_time": round(ms[k] / max(1e-9, sum(ms)), 4),
                    "traffic": pmc, "traffic_ratio": (round(pmc / bpl, 3) if pmc and bpl else None),
                    "traffic_stale": (traffic_stale(name, (tfam.get(name) or {}).get("source_sha256")) if pmc else None)}
        hbm_rows = sorted((k for k in range(nk) if KID_NAMES[k] in HBM_FAMILIES and launches[k] and by[k] > 0), key=lambda k: -ms[k])[:6]
        wf = F * B * args.steps / t_max / 1e12
        roofline = {"bound": "mfma", **fam(dom), "traffic": traffic, "traffic_stale": traffic_is_stale,
                    "whole_forward_tflops": round(wf, 2), "whole_forward_frac": round(wf / PEAK_TFLOPS[args.precision], 4),
                    "measured_in": f"second pass over the same {args.steps} steps, launched eagerly, with one HIP-event pair per launch on the launch "
                                   f"stream ({round(1e3 * allm[:, 4].max().item() / args.steps, 3)} ms per step with the events)",
                    # the other matrix-core convolution kernels of the step, same definitions (not the headline entry)
                    "other_mfma_kernels": [fam(k) for k in range(nk) if KID_NAMES[k] in MFMA16 and k != dom and launches[k]],
                    # the step's largest HBM-bound kernel families, live HIP-event durations, in situ (the
                    # neighbour phase runs two streams, so its families' durations overlap); `traffic` as for the headline entry
                    "hbm_kernels": [hbm_fam(k) for k in hbm_rows],
                    "traffic_source": ({"file": tr.get("file"), "profiles_commit": tr.get("commit"),
                                        "note": "replayed from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this script "
                                                "(tools/profile_run.sh) taken at profiles_commit, not measured in this run"}
                                       if tr else None)}
        if cpu is None and not args.no_cpu_baseline and world == 1:      # reported on rank 0 at N = 1 only
            cpu = cpu_baseline_scaled(sd, Hp, Wp)       # (--no-parity / --streaming runs: bounded half-size sample)
        res = {
            "metric": "x4 SR frames/sec, 7-frame 270x480->1080p", "value": round(value, 4), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * t_max / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x3": "bf16x3 (split-bf16 MFMA, fp32 accumulate; fp32-grade: <= 1.2e-5 max-abs vs the fp32 reference)", "bf16": "bf16", "fp16x2": "fp16 MFMA, fp32 accumulate (1 pass inside Block_, 2 passes = fp16 hi+lo activations elsewhere, split-bf16 for the returned feature cache; 2-4e-4 max-abs vs the fp32 reference, bound 1e-3)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": f"JCT-VC ClassB-shape synthetic clips: {B} clips/GPU x 7x1x{args.height}x{args.width} "
                                   f"luma (padded to {Hp}x{Wp}) + MV/residual/partition/unfiltered priors -> "
                                   f"{4 * args.height}x{4 * args.width}, " + ("streaming path (pre_L1_fea cache hit)" if args.streaming else "fresh path (pre_L1_fea=None)") + f", precision={args.precision}, "
                                   + ("pre-made Gumbel noise tensors" if args.injected_noise else "Gumbel noise drawn per forward inside the mask kernel (the reference's default behaviour)"),
                       "clips_per_gpu": B, "lr_padded": [Hp, Wp], "parallelism": f"batch-shard x{world}",
                       "weights": "random init (seed 0)", "launch": launch_info["mode"]},
            "launch": launch_info,
            "parity": {"config": parity_cfg, "max_abs": max_abs, "psnr_y_db": psnr, "bound": PARITY_BOUND,
                       "verified": bool(parity_ok and not args.no_parity),
                       "per_rank_max_abs": [float(v) for v in allm[:, 2]],
                       "full_size_max_abs": (None if max_abs_full != max_abs_full else max_abs_full)},
            "world": world, "backend": (dist.get_backend() if use_pg else None),
            "ranks": [{"rank": r, "ms_per_step": round(1e3 * allm[r, 0].item() / args.steps, 3), "local_device": int(allm[r, 6].item()),
                       "device": format_identity(allm[r, 7:15].tolist())} for r in range(world)],
            "distinct_devices": distinct,
            "roofline": roofline, "cpu_baseline": cpu, "fp16_range_guard": range_info, **extra,
        }
        if world == 1 and not args.no_parity and not args.no_extra_modes:
            # the path's other operator row, outside the timed region: the fused DCNv2 forward at the
            # alignment module's shape against its HBM roofline (same definitions as tools/bench_dcn.py)
            # BASELINE configs 2 and 5 (the single-GPU share of c5), outside the timed region
            res["other_configs"] = other_configs_line(model, sd, dev)
            res["dcn_forward"] = dcn_forward_line(dev, Hp, Wp, B)
            # SURVEY section 8f n3, also outside the timed region: the DCN-aligned CVSR_V7 on the same synthetic clips
            res["cvsr_v7"] = cvsr_v7_line(dev, d, Hp, Wp, B)
            if res["cvsr_v7"].get("parity_ok") is False:
                parity_ok = False
            torch.cuda.empty_cache()
            # SURVEY section 8f n1 / n2, outside the timed region: the reference's real evaluation loop (one sequence, one new
            # frame per forward, test_LD_22_FPS.py:183-192) and its training step (train_LD_37.py:376-381)
            res["streaming_b1"] = streaming_b1_line(dev, args.height, args.width)
            res["train_step"] = train_step_line(dev)
        print(json.dumps(res))
        if args.breakdown:
            with open(args.breakdown, "w") as f:
                tot = sum(ms)
                f.write(f"# HIP-event timing per kernel family, {args.steps} steps, B={B}, {Hp}x{Wp}; wall {t_max*1e3:.1f} ms\n")
                f.write("family launches total_ms share avg_ms TFLOP/s GB/s(algorithmic)\n")
                for k in sorted(range(nk), key=lambda k: -ms[k]):
                    if launches[k] == 0:
                        continue
                    f.write(f"{KID_NAMES[k]} {launches[k]} {ms[k]:.3f} {ms[k]/tot:.4f} {ms[k]/launches[k]:.4f} "
                            f"{fl[k]/ms[k]/1e9:.2f} {by[k]/ms[k]/1e6:.1f}\n")
    if use_pg:
        dist.destroy_process_group()
    if not devices_ok:
        sys.stderr.write(f"bench.py: {world} ranks drove only {distinct} distinct GPU(s) over backend {backend}: not a {world}-GPU measurement\n")
        sys.exit(4)
    if not parity_ok:
        sys.stderr.write(f"bench.py: PARITY FAILED: max-abs vs the CPU oracle {allm[:, 2].tolist()} (bound {PARITY_BOUND})\n")
        sys.exit(3)


def streaming_b1_line(device, H, W, T=24):
    """One sequence of T frames HxW through cdfo_amd.streaming.StreamingSR (B = 1, feature cache, noise drawn per frame):
    frames/s = T / summed per-frame forward time, as test_LD_22_FPS.py:192 computes it (here with device syncs)."""
    import numpy as np
    from arch.SIDECVSR_our import CVSR_V8
    from cdfo_amd.streaming import StreamingSR
    rs = np.random.RandomState(0)
    u8 = lambda: rs.randint(0, 256, size=(T, H, W)).astype(np.uint8)  # noqa: E731
    lr, pms, ufs = u8(), u8(), u8()
    rms = np.clip(np.round(rs.randn(T, H, W) * 6), -128, 127).astype(np.float32)
    mv = rs.randint(-64, 64, size=(2, T, (H + 7) // 8, (W + 7) // 8, 3)).astype(np.float32)
    mv[..., 2] = rs.choice([-2.0, -1.0, 1.0], size=mv.shape[:-1])
    mv = np.repeat(np.repeat(mv, 8, axis=2), 8, axis=3)[:, :, :H, :W]
    torch.manual_seed(0)
    model = CVSR_V8().to(device).eval()
    out = {}
    for key, use_graph in (("eager", False), ("hip_graph", True)):
        s = StreamingSR(model, lr, pms, rms, ufs, mv[0], mv[1], use_graph=use_graph)
        s.run()                               # warm-up: weight packing, first-touch allocations, graph capture
        s.run()
        out[key] = {"frames_per_s": round(s.fps, 2), "ms_per_frame": round(1e3 * s.seconds / T, 3)}
        del s
    del model
    torch.cuda.empty_cache()
    return {"workload": f"one sequence of {T} frames {H}x{W} -> {4 * H}x{4 * W}, B = 1, one new frame per forward with the feature "
                        "cache (test_LD_22_FPS.py:183-192), random init, synthetic priors; frames / summed synced forward time",
            **out}


def train_step_line(device, B=20, H=64, W=64, iters=3):
    """The training script's step (train_LD_37.py:376-381: 20 crops of 64x64, Charbonnier loss) through the HIP autograd path."""
    from arch.SIDECVSR_our import CVSR_V8
    from oracle.cvsr_v8_ref import make_inputs
    torch.manual_seed(0)
    m = CVSR_V8().to(device).train()
    inp = make_inputs(B, H, W, 7)
    d = {k: v.to(device) for k, v in inp.items() if k != "gumbel_u"}
    hr = torch.rand(B, 1, 4 * H, 4 * W, device=device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    fw, bw = [], []
    for it in range(iters + 1):
        m.zero_grad(set_to_none=True)
        ev[0].record()
        out, _ = m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"])
        loss = torch.sum(torch.sqrt((out - hr) ** 2 + 1e-4))
        ev[1].record()
        loss.backward()
        ev[2].record()
        torch.cuda.synchronize()
        if it:
            fw.append(ev[0].elapsed_time(ev[1]))
            bw.append(ev[1].elapsed_time(ev[2]))
    from cdfo_amd import autograd as A
    res = {"workload": f"CVSR_V8 training step, {B} crops of 7x1x{H}x{W}, Charbonnier loss, forward + backward through the HIP "
                       "autograd path (no optimizer step), random init",
           "forward_ms": round(min(fw), 2), "backward_ms": round(min(bw), 2), "step_ms": round(min(f + b for f, b in zip(fw, bw)), 2),
           "conv_arithmetic": A.conv_precision_name()}
    del m, d, hr
    torch.cuda.empty_cache()
    return res


def other_configs_line(model, sd, device):
    """BASELINE.json configs 2 (4 clips of 7x1x120x240 -> 480x960) and 5 (7x1x540x960 -> 2160x3840, one clip per GPU) on the timed
    model, outside the timed region: ms / frames/s / whole-forward TFLOP/s each (c2 also from a HIP-graph replay: ~270 launches in
    ~15 ms are gap-bound), and clip 0 of the c2 batch against the CPU oracle with the noise that forward drew."""
    from oracle.cvsr_v8_ref import cvsr_v8_forward, make_inputs
    out = {}

    def run(B, H, W, seed, pad_rows, steps, want_parity, want_graph):
        inp = make_inputs(B, H, W, seed, pad_rows=pad_rows)
        dd = {k: v.to(device) for k, v in inp.items() if k != "gumbel_u"}
        fwd = lambda: model(dd["x"], dd["mvs0"], dd["mvs1"], dd["pms"], dd["rms"], dd["ufs"])      # noqa: E731
        res = {}
        with torch.no_grad():
            cap = []
            model.capture_noise = cap if want_parity else None
            got, _ = fwd()
            model.capture_noise = None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fwd()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            res.update({"ms_per_step": round(ms, 3), "frames_per_s": round(B / ms * 1e3, 2),
                        "whole_forward_tflops": round(flops_per_clip(H, W) * B / ms / 1e9, 1), "steps": steps})
            if want_graph:
                try:
                    g = model.capture(dd["x"], dd["mvs0"], dd["mvs1"], dd["pms"], dd["rms"], dd["ufs"])
                    g.replay()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        g.replay()
                    torch.cuda.synchronize()
                    gms = (time.perf_counter() - t0) / steps * 1e3
                    res["hip_graph"] = {"ms_per_step": round(gms, 3), "frames_per_s": round(B / gms * 1e3, 2)}
                    del g
                except Exception as e:      # a line of the report, never a reason to lose the headline
                    res["hip_graph"] = {"error": repr(e)[:200]}
            if want_parity:
                ref, _ = cvsr_v8_forward(sd, inp["x"][0:1], None, inp["mvs1"][0:1], inp["pms"][0:1], inp["rms"][0:1], inp["ufs"][0:1], None,
                                         [u[0:1].cpu() for u in cap])
                e = (got[0:1].cpu() - ref).abs().max().item()
                res.update({"clip0_max_abs_vs_oracle": e, "within_1e-3": bool(e <= PARITY_BOUND)})
        del dd, got
        torch.cuda.empty_cache()
        return res

    out["c2"] = {"workload": "4 clips x 7x1x120x240 -> 480x960, fresh path, one GPU", **run(4, 120, 240, 1001, 0, 10, True, True)}
    out["c5_one_clip"] = {"workload": "1 clip x 7x1x540x960 (padded to 544) -> 2160x3840, fresh path = one GPU's share of config 5",
                          **run(1, 544, 960, 1005, 4, 3, False, False)}
    return out


def cvsr_v7_line(device, d, H, W, B, steps=2):
    """CVSR_V7 (arch.py:4215-4367) on the timed clips in both arithmetic modes; clip 0 of the fp16x2 batch is checked against the CPU
    oracle (oracle/cvsr_v7_ref.py over the threaded C DCN oracle) at the timed size: the run FAILS above the 1e-3 bound.  `roofline`
    names the forward's dominant kernel family by live HIP-event time (second pass, one event pair per launch)."""
    from arch.SIDECVSR_our import CVSR_V7
    from cdfo_amd import _lib
    from oracle.cvsr_v7_ref import cvsr_v7_forward, make_state_dict_v7
    lib = _lib.lib()
    sd = make_state_dict_v7(0)          # seeded random init; the offset heads are NOT zero (the reference's init leaves the DCN undeformed)
    m = CVSR_V7()
    m.load_state_dict(sd, strict=True)
    m = m.to(device).eval()
    g = torch.Generator(device=device).manual_seed(7)
    noise = [torch.rand(B, 64, H >> lv, W >> lv, device=device, generator=g).clamp_min_(1e-6) for lv in (2, 1, 0) for _ in range(12)]
    out = {}
    got0 = None
    for prec in ("bf16x3", "fp16x2"):
        m.precision = prec
        with torch.no_grad():
            o, _ = m(d["x"], -d["mvs1"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=noise)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                m(d["x"], -d["mvs1"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=noise)
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        out[prec] = {"ms_per_forward": round(ms, 2), "frames_per_s": round(B / ms * 1e3, 2)}
        if prec == "fp16x2":
            got0 = o[0:1].cpu()
        del o
    # the same fp16x2 forward replayed from a HIP graph (CVSR_V7.capture): ~1 000 launches per forward are partly launch-bound
    try:
        with torch.no_grad():
            g = m.capture(d["x"], -d["mvs1"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=noise)
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                g.replay()
            torch.cuda.synchronize()
        gms = (time.perf_counter() - t0) / steps * 1e3
        out["fp16x2_hip_graph"] = {"ms_per_forward": round(gms, 2), "frames_per_s": round(B / gms * 1e3, 2)}
        del g
    except Exception as e:
        out["fp16x2_hip_graph"] = {"error": repr(e)[:200]}
    torch.cuda.empty_cache()
    # dominant kernel family of the fp16x2 forward: one more forward with one HIP-event pair per launch
    nk = lib.cdfo_prof_kid_count()
    _lib.check(lib.cdfo_prof_begin(8000), "cdfo_prof_begin")
    with torch.no_grad():
        m(d["x"], -d["mvs1"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=noise)
    torch.cuda.synchronize()
    launches, kms, fl, by = (C.c_int * nk)(), (C.c_double * nk)(), (C.c_double * nk)(), (C.c_double * nk)()
    nrec = lib.cdfo_prof_end(launches, kms, fl, by, nk)
    roof = None
    if nrec > 0:
        dom = max(range(nk), key=lambda k: kms[k])
        avg = kms[dom] / max(1, launches[dom])
        name = KID_NAMES[dom]
        if name in MFMA16:
            passes = MFMA16[name][1] or MFMA_PASSES["fp16x2"]
            ach = fl[dom] / max(1, launches[dom]) / (avg * 1e-3) / 1e12 if avg > 0 else 0.0
            roof = {"bound": "mfma", "kernel": name, "kernel_symbol": MFMA16[name][0], "achieved": round(ach, 2), "peak": PEAK_TFLOPS["fp16x2"],
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS["fp16x2"], 4), "mfma_passes_per_mac": passes}
        else:
            ach = by[dom] / max(1, launches[dom]) / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
            roof = {"bound": "hbm", "kernel": name, "kernel_symbols": HBM_FAMILIES.get(name, [name]), "achieved": round(ach, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
        roof.update({"launches_per_forward": launches[dom], "avg_launch_ms": round(avg, 4), "share_of_gpu_time": round(kms[dom] / max(1e-9, sum(kms)), 4),
                     "launches_per_forward_all_families": int(sum(launches)), "traffic": None,
                     "measured_in": "one extra fp16x2 forward with one HIP-event pair per launch on the launch stream"})
    del m
    torch.cuda.empty_cache()
    res = {"workload": f"CVSR_V7 forward, {B} clips x 7x1x{H}x{W}, fresh path, seeded random init (non-zero offset heads)", **out, "roofline": roof}
    # ---- deferred parity gate: CPU oracle on clip 0 of the same batch (same weights, same injected noise)
    torch.set_num_threads(_host_cores())
    t0 = time.perf_counter()
    with torch.no_grad():
        ref, _ = cvsr_v7_forward(sd, d["x"][0:1].cpu(), -d["mvs1"][0:1].cpu(), d["mvs1"][0:1].cpu(), d["pms"][0:1].cpu(), d["rms"][0:1].cpu(),
                                 d["ufs"][0:1].cpu(), None, [u[0:1].cpu() for u in noise])
    e = (got0 - ref).abs().max().item()
    res["parity"] = {"config": f"clip 0 of the fp16x2 batch ({H}x{W}) vs oracle/cvsr_v7_ref.py (C DCN oracle)", "max_abs": e, "bound": PARITY_BOUND,
                     "verified": bool(e <= PARITY_BOUND), "oracle_seconds": round(time.perf_counter() - t0, 1)}
    res["parity_ok"] = bool(e <= PARITY_BOUND)
    return res


def dcn_forward_line(device, H, W, B, iters=10):
    from cdfo_amd.dcn import modulated_deform_conv
    C, Co, dg = 64, 64, 16
    g = torch.Generator(device=device).manual_seed(0)
    x = torch.randn(B, C, H, W, device=device, generator=g)
    w = torch.randn(Co, C, 3, 3, device=device, generator=g) / 24
    b = torch.randn(Co, device=device, generator=g)
    mv = torch.rand(B, 2, (H + 7) // 8, (W + 7) // 8, device=device, generator=g) * 6 - 3    # block-constant motion field
    mv = mv.repeat_interleave(8, 2).repeat_interleave(8, 3)[:, :, :H, :W]
    off = mv.repeat(1, dg * 9, 1, 1) + 0.5 * torch.randn(B, 2 * dg * 9, H, W, device=device, generator=g)
    msk = torch.rand(B, dg * 9, H, W, device=device, generator=g)
    with torch.no_grad():
        for _ in range(2):
            modulated_deform_conv(x, off, msk, w, b, 1, 1, 1, 1, dg)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            modulated_deform_conv(x, off, msk, w, b, 1, 1, 1, 1, dg)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = (C + 3 * dg * 9 + Co) * H * W * 4 * B + w.numel() * 4
    ach = nbytes / ms / 1e6
    traffic, traffic_commit, dcn_stale = None, None, None
    tpath = next((q for q in (os.path.join(ROOT, "profiles", f"traffic_dcn_r{r:02d}.json") for r in range(9, 0, -1)) if os.path.exists(q)), "")
    if tpath and (B, H, W) == (8, 272, 480):     # PMC passes of tools/bench_dcn.py at exactly this shape
        try:
            tj = json.load(open(tpath))
            # the kernel the file was measured on must be the one this build runs (dcn_win since round 3)
            if tj.get("kernel", "").startswith("dcn_win"):
                traffic = tj["hbm_bytes_per_launch"] + tj["prepass_hbm_bytes_per_launch"]
                traffic_commit = tj.get("commit")
                dcn_stale = traffic_stale("dcn", tj.get("source_sha256"))
        except Exception:
            traffic = None
    # the operator's backward at the same shape (all five gradients, as tools/bench_dcn.py --backward)
    from cdfo_amd import deform_conv_cuda as ext
    go = torch.randn(B, Co, H, W, device=device, generator=g)
    gi, gw, gb, goff, gm = (torch.zeros_like(t) for t in (x, w, b, off, msk))
    e = torch.empty(0, device=device)
    bwd = lambda: ext.modulated_deform_conv_cuda_backward(x, w, b, e, off, msk, e, gi, gw, gb, goff, gm, go, 3, 3, 1, 1, 1, 1,  # noqa: E731
                                                          1, 1, 1, dg, True)
    for _ in range(2):
        bwd()
    e0.record()
    for _ in range(iters):
        bwd()
    e1.record()
    torch.cuda.synchronize()
    ms_b = e0.elapsed_time(e1) / iters
    return {"workload": f"DCNv2 forward C=Co=64 dg=16 3x3, {B}x{H}x{W}, MV-like offsets", "ms_per_launch": round(ms, 4),
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
                         "algorithmic_bytes_per_launch": nbytes, "traffic": traffic,
                         "traffic_ratio": (round(traffic / nbytes, 3) if traffic else None), "profiles_commit": traffic_commit,
                         "traffic_stale": dcn_stale},
            "backward_ms_per_launch": round(ms_b, 4)}


def _host_cores():
    """Threads for the CPU baseline: the PHYSICAL cores this process may run on (unique (package, core) pairs of the CPUs in
    its affinity mask; SMT siblings counted once), capped by a cgroup CPU quota if the box sets one."""
    try:
        allowed = set(os.sched_getaffinity(0))
    except Exception:
        allowed = set(range(os.cpu_count() or 1))
    cores = set()
    try:
        cpu, phys = None, 0
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                cpu = int(line.split(":")[1])
            elif line.startswith("physical id"):
                phys = int(line.split(":")[1])
            elif line.startswith("core id") and cpu in allowed:
                cores.add((phys, int(line.split(":")[1])))
    except Exception:
        cores = set()
    n = len(cores) or len(allowed)
    try:                                             # cgroup v2 quota "max 100000" / "1600000 100000"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(math.ceil(int(q) / int(per)))))
    except Exception:
        pass
    return max(1, n)


def device_identity(index):
    """Eight numbers that identify the physical GPU behind cuda:index: PCI domain / bus / device and the 16-byte UUID as four
    32-bit words (zeros where the runtime does not expose them) -- carried by the single metric all_gather."""
    pr = torch.cuda.get_device_properties(index)
    pci = [float(getattr(pr, a, -1)) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
    words = [0.0] * 4
    try:
        raw = getattr(pr, "uuid", None)
        b = raw.bytes if hasattr(raw, "bytes") else bytes.fromhex(str(raw).replace("-", "").replace("GPU", ""))
        words = [float(int.from_bytes(b[4 * i:4 * i + 4], "big")) for i in range(4)]
    except Exception:
        pass
    return [*pci, *words, float(pr.multi_processor_count)]


def format_identity(v):
    dom, bus, devn = (int(x) for x in v[0:3])
    return {"pci": (f"{dom:04x}:{bus:02x}:{devn:02x}.0" if bus >= 0 else None),
            "uuid": "".join(f"{int(x):08x}" for x in v[3:7]), "compute_units": int(v[7])}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def oracle_clip(sd, clip, noise, Hp, Wp, time_it=True):
    """The CPU oracle (torch-cpu port of the reference forward) on ONE clip of the timed workload at its full size: the
    parity reference and, timed after a small warm-up forward, the cpu_baseline (frames/s = 1 / seconds per clip)."""
    from oracle.cvsr_v8_ref import cvsr_v8_forward, make_inputs
    cores = _host_cores()
    torch.set_num_threads(cores)
    with torch.no_grad():
        w = make_inputs(1, 32, 32, 5)
        cvsr_v8_forward(sd, w["x"], None, w["mvs1"], w["pms"], w["rms"], w["ufs"], None, w["gumbel_u"])      # warm-up
        t0 = time.perf_counter()
        ref, _ = cvsr_v8_forward(sd, clip["x"], None, clip["mvs1"], clip["pms"], clip["rms"], clip["ufs"], None, noise)
        dt = time.perf_counter() - t0
    cpu = None
    if time_it:
        cpu = {"value": round(1.0 / dt, 5), "unit": "frames/s", "cores": cores, "kind": "port",
               "sample": f"1 clip 7x1x{Hp}x{Wp} of the timed batch, one full forward = {dt:.2f} s on {cores} torch threads after a "
                         "32x32 warm-up forward (the clip is also the parity reference)",
               "cpu_model": _cpu_model(), "torch": torch.__version__}
    return ref, cpu


def cpu_baseline_scaled(sd, Hp, Wp):
    """Fallback when no full-size oracle run happened (--no-parity / --streaming): ONE clip at half the workload's height
    and width, scaled to the workload by the algorithmic-FLOP ratio."""
    from oracle.cvsr_v8_ref import cvsr_v8_forward, make_inputs
    cores = _host_cores()
    torch.set_num_threads(cores)
    h, w = max(8, Hp // 2 // 8 * 8), max(8, Wp // 2 // 8 * 8)
    s = make_inputs(1, h, w, 77)
    with torch.no_grad():
        t0 = time.perf_counter()
        cvsr_v8_forward(sd, s["x"], None, s["mvs1"], s["pms"], s["rms"], s["ufs"], None, s["gumbel_u"])
        dt = time.perf_counter() - t0
    scale = flops_per_clip(Hp, Wp) / flops_per_clip(h, w)
    return {"value": round(1.0 / (dt * scale), 5), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"1 clip 7x1x{h}x{w} forward = {dt:.2f} s on {cores} torch threads, scaled x{scale:.2f} by "
                      f"algorithmic FLOPs to {Hp}x{Wp}", "cpu_model": _cpu_model(), "torch": torch.__version__}


if __name__ == "__main__":
    main()
