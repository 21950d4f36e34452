#!/usr/bin/env python3
"""bench.py -- decode Mpixel/s (+ bpp, PSNR) of the two_layer_syn codec on Kodak-shaped inputs.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One STEP = one decode pass of the hot path over the rank's batch: (z_hat, symbols) resident in HBM ->
hyper-synthesis -> y_hat = symbols + mu -> two-layer synthesis -> uint8 pixels, for a Kodak-24-shaped
synthetic set (18 x 512x768 + 6 x 768x512; mshyper/configs/two_layer_syn.py, random-init weights --
there is no network for Kodak or checkpoints).  Weak scaling: every rank decodes its own set.
The JSON line also carries the encode+decode rate, the (bpp, PSNR) of the set, the live-measured
roofline of the dominant kernel and a CPU baseline (torch-CPU port of the same decode, rank 0, N=1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import __graft_entry__ as graft  # noqa: E402  (stdlib only: the launcher parent must not touch the GPU)

np = None      # numpy / torch are imported by main() in the worker processes only: the parent of a self-launched
torch = None   # N-rank run (launch_ranks) never initialises HIP and never imports them

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
KODAK_SHAPES = [(512, 768)] * 18 + [(768, 512)] * 6


def synthetic_batch(n, h, w, seed, device):
    """Seeded smooth images (SURVEY.md 8d recipe) generated on the device: 16 low-frequency cosines per
    channel + N(0, 4^2) noise, rounded to uint8 values, normalised to [-0.5, 0.5]."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    yy = (torch.arange(h, device=device, dtype=torch.float32) / h).view(1, h, 1, 1, 1)
    xx = (torch.arange(w, device=device, dtype=torch.float32) / w).view(1, 1, w, 1, 1)
    f = torch.rand((n, 1, 1, 3, 16, 2), device=device, generator=g) * 6.0
    ph = torch.rand((n, 1, 1, 3, 16), device=device, generator=g) * (2 * np.pi)
    amp = torch.rand((n, 1, 1, 3, 16), device=device, generator=g) * 20.0 + 4.0
    img = 128.0 + (amp * torch.cos(2 * np.pi * (f[..., 0] * yy + f[..., 1] * xx) + ph)).sum(-1)
    img = img + 4.0 * torch.randn(img.shape, device=device, generator=g)
    img = torch.clamp(torch.round(img), 0, 255)
    return (img / 255.0 - 0.5).contiguous()


def host_cpu():
    """(model string, physical core count) of the host from /proc/cpuinfo (printed next to the CPU baseline)."""
    model, cores = None, set()
    try:
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model is None:
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    return model, (len(cores) or os.cpu_count())


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """``python bench.py --gpus N`` outside torchrun: start N child processes of this script, one rank per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, exactly what torch.distributed.run would set),
    relay rank 0's JSON line, and return non-zero if any rank does.  The parent makes no HIP call and replaces no
    process image: children are ordinary subprocesses, stopped by their exact PIDs if a sibling fails."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this pool
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    rc = 0
    pending = set(range(n))
    out0 = None
    while pending:
        for r in sorted(pending):
            p = procs[r]
            if r == 0 and out0 is None:
                try:
                    out0, _ = p.communicate(timeout=0.2)
                except subprocess.TimeoutExpired:
                    continue
            elif p.poll() is None:
                continue
            pending.discard(r)
            if p.returncode != 0 and rc == 0:
                rc = p.returncode
                print(f"bench.py: rank {r} exited with {p.returncode}; stopping the other ranks", file=sys.stderr)
                for q in pending:
                    procs[q].terminate()
        time.sleep(0.05)
    for ln in (out0 or "").splitlines():          # rank 0's JSON line to stdout; library chatter (gloo prints there) to stderr
        print(ln, file=sys.stdout if ln.startswith("{") else sys.stderr, flush=True)
    return rc if rc >= 0 else 1


def launch_check(args, json_fd=1):
    """--launch-check: rendezvous, barrier, one all-gather, one max-reduce -- the collectives of the timed path
    without any kernel (works on a CPU-only host with gloo).  Prints a line that is visibly NOT a measurement."""
    from shallow_ntc_amd import distributed as D
    rank, local_rank, world = D.init()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("SNTC_LAUNCH_CHECK_FAIL_RANK") == str(rank):     # test hook: a failing rank must fail the job
        sys.exit(3)
    D.barrier()
    info = D.describe_world(None)
    table = D.gather_rows([[float(rank), 2.0 * rank]], [rank], world)
    t = D.max_over_ranks(float(rank))
    # the schedule broadcast of the timed path (rank 0 measures, everyone runs its choices): one all_gather_object
    choices = D.share_from_rank0([(0, "conv", 192, 192, 18, 256, 384, 3, 0)] if rank == 0 else None)
    assert choices == [(0, "conv", 192, 192, 18, 256, 384, 3, 0)], choices
    D.barrier()
    if rank == 0:
        assert t == world - 1 and table[:, 0].tolist() == list(range(world))
        os.write(json_fd, (json.dumps(dict(metric="launch-check (no kernels, not a measurement)", value=None, n_gpus=world,
                                           dry_run=True, rccl=info)) + "\n").encode())
    D.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--images", type=int, default=24, help="images per rank (Kodak-24 shaped set)")
    ap.add_argument("--workload", choices=["kodak24", "w1", "w3"], default="kodak24",
                    help="kodak24: the metric's configuration (default); w1: SURVEY 8d synthetic batches of "
                         "--images (default 64) 256x256 crops; w3: 5 x 1200x1200 (Tecnick-shaped, padded to 1216)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decode-only", action="store_true",
                    help="only launch decode kernels (the command profiles/ is recorded with): skips the "
                         "encode+decode loop, the (bpp, PSNR) evaluation and the CPU baseline")
    ap.add_argument("--cpu-passes", type=int, default=5, help="timed passes of the CPU baseline after one warm-up (SURVEY.md 8d: >= 5)")
    ap.add_argument("--streams", type=int, default=0,
                    help="decode the batches of the set on this many HIP streams at once (0 = one per batch shape): the tail of "
                         "one batch's kernels is filled by the other's blocks; 1 = serial on the current stream")
    ap.add_argument("--set-decode", action="store_true",
                    help="A/B: Model.decode_set (the hyper-syntheses of the batch shapes side by side, then ONE synthesis launch for all "
                         "of them) instead of one Model.decode per batch shape on its own stream (measured: the join in front of the "
                         "shared launch costs more overlap than the launch saves: 3.39 against 3.16 ms per step)")
    ap.add_argument("--reverse-batches", action="store_true", help="A/B: feed the decode streams largest batch first (default: smallest first)")
    ap.add_argument("--stream-map", type=str, default="",
                    help="A/B: comma-separated stream index per batch of the set (default: batch i on stream i %% streams)")
    ap.add_argument("--chunk", type=int, default=0, help="split every batch shape into sub-batches of at most this many images (0 = no split)")
    ap.add_argument("--graph", action="store_true",
                    help="replay a captured HIP graph instead of eager launches: one graph for the set with a branch per batch "
                         "(graphs.DecodeSetGraph), or one per batch shape with --streams 1 (measured, round 5: 3.14 ms either way -- "
                         "the host already runs ahead of the GPU)")
    ap.add_argument("--no-fuse", action="store_true",
                    help="A/B: run the ResidualBlock tails as two launches instead of the fused one (same bits)")
    ap.add_argument("--no-autotune", action="store_true",
                    help="A/B: let the cost model pick every convolution's tile and schedule instead of measuring the candidates "
                         "once per layer shape before the timed regions (sntc_conv_plan_tune; same bits either way)")
    ap.add_argument("--tuning-file", type=str, default=str(ROOT / "profiles" / "tuning_gfx950.json"),
                    help="the measured launch schedules of an earlier run of this command (every region's per-layer choices and the "
                         "decode step's): applied WITHOUT measuring where present and matching (arch, workload, flags); what it does "
                         "not cover is measured as before.  Same bits either way.  '' = measure everything")
    ap.add_argument("--retune", action="store_true", help="ignore --tuning-file and measure every schedule again")
    ap.add_argument("--write-tuning", type=str, default="",
                    help="write this run's schedules (measured or applied) to this path at the end (copy it to profiles/tuning_gfx950.json "
                         "to persist them; default: gpurun_out/tuning_gfx950.json when anything was measured)")
    ap.add_argument("--no-preheat", action="store_true",
                    help="skip the second measurement of the headline region behind 40 more warm-up steps (`sustained` in the line; "
                         "`value` itself never has them)")
    ap.add_argument("--no-decode-tune", action="store_true",
                    help="A/B: skip the step-level choice of the decode launches' schedules (ops.tune_step on the two-stream step)")
    ap.add_argument("--launch-check", action="store_true",
                    help="only rendezvous the ranks and run the path's collectives (no kernels); prints a dry-run line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # not under torchrun: be the launcher (no GPU call before this)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # stdout carries ONE JSON line and nothing else: libraries write banners to the C-level stdout (RCCL prints its version
    # block there, flushed at exit, i.e. AFTER the line), so file descriptor 1 is pointed at stderr for the life of the process
    # and the line goes to a private duplicate of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    global np, torch
    import numpy as np
    import torch
    graft.load_package()
    if args.launch_check:
        return launch_check(args, json_fd)
    from shallow_ntc_amd import distributed as D
    from shallow_ntc_amd import ops
    if args.no_fuse:
        ops.FUSE_RESIDUAL_TAIL = False
    from shallow_ntc_amd.mshyper import configs
    from shallow_ntc_amd.mshyper.models import Model

    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback for the hot path"
    rccl_note = None
    if int(os.environ.get("WORLD_SIZE", 1)) == 1 and not os.environ.get("SNTC_DIST_BACKEND"):
        # N = 1: still form a ONE-rank RCCL group, so that the barrier / max-reduce / all-gather of the timed path run through
        # RCCL on the GPU (the only RCCL evidence a one-GPU box can give).  If RCCL cannot initialise here the bench carries
        # on without a group and says so in `rccl`.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        try:
            D.init(backend="nccl")
        except Exception as e:                     # noqa: BLE001 -- any backend failure degrades to "no group", never fails the bench
            rccl_note = f"one-rank nccl group not formed: {type(e).__name__}: {e}"
    rank, local_rank, world = D.init()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("SNTC_SHARE_GPU"):          # 2-rank dry run of the distributed flow on a 1-GPU box (gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ops.side_streams(3, dev)          # the library's stream pool FIRST: the first three streams of a process own a hardware queue each
    world_info = D.describe_world(dev)                # one all_gather_object: backend, world, every rank's device
    if rccl_note:
        world_info["note"] = rccl_note

    cfg = configs.two_layer_syn(rd_lambda=0.005)
    model = Model(device=dev, **cfg)
    # spread the predicted scales like a trained model would (untrained nets give one degenerate sigma)
    w = model.get_weights()
    rng = np.random.default_rng(4321)
    b = w["hyper_synthesis/layer_2/bias"].copy()
    b[320:] = rng.uniform(-2.0, 2.5, size=320)
    w["hyper_synthesis/layer_2/bias"] = b.astype(np.float32)
    model.set_weights(w)

    if args.workload == "w1":
        shapes = [(256, 256)] * (64 if args.images == 24 else args.images)
    elif args.workload == "w3":
        shapes = [(1200, 1200)] * (5 if args.images == 24 else args.images)
    else:
        shapes = KODAK_SHAPES[:args.images] if args.images <= 24 else [KODAK_SHAPES[i % 24] for i in range(args.images)]
    groups = {}
    for i, s in enumerate(shapes):
        key = s if args.chunk <= 0 else (s, sum(1 for j in range(i) if shapes[j] == s) // args.chunk)
        groups.setdefault(key, []).append(i)
    batches = []   # (image ids, x)
    for key, ids in groups.items():
        h, wd = key if args.chunk <= 0 else key[0]
        batches.append((ids, synthetic_batch(len(ids), h, wd, 1234 + 97 * rank + h + 7 * ids[0], dev), (h, wd)))
    pixels_per_step = sum(h * wd for h, wd in shapes)

    # ---- codes resident in HBM: synthetic latents of the encoder's output shapes (SURVEY.md 8d:
    # z ~ round(N(0, 3^2)), y - mu ~ round(Laplace(0, 2))), so the timed region launches decode kernels only
    codes = []
    g = torch.Generator(device=dev)
    g.manual_seed(99 + rank)
    for ids, x, (h, wd) in batches:
        n = len(ids)
        hp, wp = -(-h // 64) * 64, -(-wd // 64) * 64               # latents live on the padded grid (pad_images to a multiple of 64)
        z_hat = torch.round(3.0 * torch.randn((n, hp // 64, wp // 64, 320), device=dev, generator=g)).contiguous()
        u = torch.rand((n, hp // 16, wp // 16, 320), device=dev, generator=g) - 0.5
        sym = torch.round(-2.0 * torch.sign(u) * torch.log1p(-2.0 * u.abs())).to(torch.int32).contiguous()
        codes.append((z_hat, sym, (h, wd), x))
    # the decode streams are fed smallest batch first (measured: 3.12 against 3.14 ms per step the other way round -- the small
    # batch's kernels are in flight when the large batch's long launches start, instead of queueing behind them)
    codes.sort(key=lambda c: c[0].shape[0] * c[2][0] * c[2][1], reverse=bool(args.reverse_batches))
    batches.sort(key=lambda b: len(b[0]) * b[2][0] * b[2][1], reverse=bool(args.reverse_batches))     # the encode-side regions alike (-0.3 %)
    torch.cuda.synchronize()

    def decode_eager():
        return [model.decode(z_hat, sym, hw, check=False) for z_hat, sym, hw, _x in codes]

    nstreams = len(codes) if args.streams == 0 else args.streams
    side = ops.side_streams(nstreams, dev) if nstreams > 1 else []      # the library's pool: its first three streams own a hardware queue each

    smap = [int(v) for v in args.stream_map.split(",")] if args.stream_map else None

    def decode_streams():
        """Independent batches on independent streams (joined back into the current stream before returning)."""
        cur = torch.cuda.current_stream()
        outs = []
        for i, (z_hat, sym, hw, _x) in enumerate(codes):
            st = side[smap[i] if smap else i % nstreams]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(model.decode(z_hat, sym, hw, check=False))
        for st in side:
            cur.wait_stream(st)
        return outs

    def decode_set_step():
        """All batches of the set through Model.decode_set: hyper-syntheses side by side on one stream per batch, the synthesis of
        every batch in ONE launch (per-image geometry in the kernel), output layers per batch."""
        return model.decode_set([(z_hat, sym, hw) for z_hat, sym, hw, _x in codes], check=False)

    set_decode = args.set_decode and args.streams == 0 and 2 <= len(codes) <= 4
    if set_decode:
        decode_step = decode_set_step
    elif not args.graph:
        decode_step = decode_streams if nstreams > 1 and len(codes) > 1 else decode_eager
    else:                                  # one captured HIP graph per batch shape, replayed every step
        from shallow_ntc_amd.graphs import DecodeGraph, DecodeSetGraph
        if nstreams > 1 and len(codes) > 1:      # one graph for the set, one branch per batch
            set_graph = DecodeSetGraph(model, codes)
            for _ in range(3):
                for got, ref in zip(set_graph(), decode_eager()):
                    assert torch.equal(got, ref), "HIP-graph replay differs from eager decode"
            decode_step = set_graph
        else:
            graphs = [DecodeGraph(model, z_hat, sym, hw) for z_hat, sym, hw, _x in codes]
            # the graph output must equal the eager output bit for bit
            for g, ref in zip(graphs, decode_eager()):
                assert torch.equal(g(), ref), "HIP-graph replay differs from eager decode"

            def decode_step():
                return [g() for g in graphs]

    def on_streams(fn_per_batch):
        """Run fn(batch) for every batch of the set, independent batches on independent streams."""
        if not side or len(batches) < 2 or ops.AUTOTUNE:
            return [fn_per_batch(b) for b in batches]
        cur = torch.cuda.current_stream()
        outs = []
        for i, b in enumerate(batches):
            st = side[i % nstreams]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(fn_per_batch(b))
        for st in side:
            cur.wait_stream(st)
        return outs

    def on_codes(fn_per_code):
        """fn(code) for every (z_hat, symbols, hw, x) of the set, independent batches on independent streams."""
        if not side or len(codes) < 2 or ops.AUTOTUNE:
            return [fn_per_code(c) for c in codes]
        cur = torch.cuda.current_stream()
        outs = []
        for i, c in enumerate(codes):
            st = side[i % nstreams]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(fn_per_code(c))
        for st in side:
            cur.wait_stream(st)
        return outs

    def e2e_one(batch):
        ids, x, hw = batch
        z_hat, sym, _, _ = model.encode(x, check=False)
        return model.decode(z_hat, sym, hw, reference=x, check=False)

    def e2e_step():
        return on_streams(e2e_one)

    rank_walls = []

    def timed(fn, steps, warmup, stats=None, headline=False):
        """The contract's timing: W untimed warm-up steps, then EXACTLY K steps bracketed by a barrier +
        torch.cuda.synchronize() on both sides, MAX over ranks of the wall time.  ``stats`` (a dict) additionally receives
        the per-step durations from HIP events recorded on the launch stream after every step (hipEvents,
        median + min) -- every step joins its side streams back into the current stream before it returns."""
        for _ in range(warmup):
            fn()
        D.barrier()
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if stats is not None else None
        t0 = time.perf_counter()
        if evs:
            evs[0].record()
        for i in range(steps):
            fn()
            if evs:
                evs[i + 1].record()
        torch.cuda.synchronize()
        D.barrier()
        mine = time.perf_counter() - t0
        wall = D.max_over_ranks(mine, device=dev)
        if headline:                           # the headline region: every rank's own time travels with the maximum
            rank_walls[:] = D.all_ranks(mine, device=dev)
        ops.check_conv_status()         # the timed steps deferred their stream-K health check (check=False): raise here, not a wrong number
        if evs:
            per = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(steps)])
            stats.update(steps=steps, warmup=warmup, mean_ms=round(1e3 * wall / steps, 4), median_ms=round(float(np.median(per)), 4),
                         min_ms=round(float(per.min()), 4), timer="HIP events on the launch stream per step; mean = wall clock / steps")
            if headline:
                stats["per_step_ms"] = [round(float(v), 3) for v in per]      # the clock ramp, if any, is visible here
        return wall

    # >= 10 warm-up and >= 50 timed iterations per region (the 40 - 55 ms encode-side regions: >= 20)
    R_WARM, R_STEPS, R_STEPS_ENC = 10, max(50, args.steps), max(20, min(args.steps, 50))

    def region_frac(fn, seconds_per_step):
        """Algorithmic conv FLOPs of one pass of ``fn`` (sntc_conv_flops of every launch, recorded by ops.PROFILE in an
        untimed extra pass) / the region's measured time / the fp32-MFMA peak -- the whole region, stream kernels included."""
        ops.PROFILE = []
        fn()
        torch.cuda.synchronize()
        flops = sum(e["flops"] for e in ops.PROFILE)
        ops.PROFILE = None
        tf = flops / seconds_per_step / 1e12
        return dict(gflop_per_step=round(flops / 1e9, 2), tflops=round(tf, 2), frac_of_fp32_mfma_peak=round(tf / FP32_MFMA_PEAK_TFLOPS, 4))

    tune_seconds = [0.0]
    # Schedules persisted by an earlier run of the same command on the same architecture (21 s of every run
    # went into re-measuring them): keyed by a fingerprint of what decides which plans exist and which shapes they see.
    tune_fp = dict(arch=torch.cuda.get_device_properties(dev).gcnArchName.split(":")[0], workload=args.workload, images=len(shapes),
                   decode_only=bool(args.decode_only))
    saved_tuning, saved_applied = None, [0]
    tpath = Path(args.tuning_file) if args.tuning_file and not args.retune and not args.no_autotune else None
    if tpath is not None and tpath.exists():
        try:
            cand = json.loads(tpath.read_text())
            if cand.get("fingerprint") == tune_fp:
                cand["entries"] = [tuple(e) for e in cand["entries"]]
                saved_tuning = cand
        except (OSError, ValueError, KeyError):
            saved_tuning = None

    def tune(fn):
        """One untimed, serial, single-stream pass of a region's step in which every convolution plan measures its (tile,
        schedule) candidates on the shapes it meets and keeps the fastest (ops.autotune / sntc_conv_plan_tune).  All
        candidates compute the same chains: the timed steps below run other launches of the same arithmetic, same bits.
        With N > 1 ranks only rank 0 measures; its choices reach the others by one all_gather_object, so every rank runs
        identical launches (a rank with another schedule would be a skewed scaling point) and the job spends the tuning
        time once.  The other ranks run the step once, untimed, so that their allocations are warm as well."""
        if args.no_autotune:
            return
        t0 = time.perf_counter()
        if saved_tuning is not None:       # an earlier run's choices: the step once as it is (its plans exist after that), then the entries
            fn()
            torch.cuda.synchronize()
            saved_applied[0] = max(saved_applied[0], ops.import_tuning(saved_tuning["entries"], strict=False))
        if rank == 0:
            with ops.autotune():           # measures only what the entries above did not cover
                fn()
        else:
            fn()
        torch.cuda.synchronize()
        if world > 1:
            choices = D.share_from_rank0(ops.export_tuning() if rank == 0 else None)
            if rank != 0:
                ops.import_tuning(choices)
        tune_seconds[0] += time.perf_counter() - t0

    # (The hyper-synthesis plans are shared by encode and decode: what the encode pass below measures for them with the device
    # to itself is then re-judged by the decode step's own clock -- see `decode_tuning` further down.)
    enc_fn = w1_x = w1_step = None
    if not args.decode_only:
        enc_fn = lambda: on_streams(lambda b: model.encode(b[1], check=False))
        w1_x = synthetic_batch(64, 256, 256, 4321 + rank, dev)   # encode+decode on the synthetic 256x256 batches the north star asks for

        def w1_step():
            z_hat, sym, _, _ = model.encode(w1_x, check=False)
            model.decode(z_hat, sym, (256, 256), reference=w1_x, check=False)

        tune(enc_fn)                                             # untimed set-up of the regions below, done before anything is timed
        tune(w1_step)
    # The headline's own launches, judged by the step's own clock (ops.tune_step): the two batch shapes run on two streams at
    # once and the steps follow each other without a gap, so a candidate is timed as bursts of four steps of the whole set --
    # not with the device to itself as above (round 3: such choices LOSE here, 3.40 against 3.29 ms).  Untimed set-up like the
    # passes above; every candidate computes the same chains (the pixels are compared below).  Rank 0 measures for all ranks.
    decode_tuning = None
    if not args.no_autotune and not args.graph and not args.no_decode_tune:
        t0 = time.perf_counter()
        want_px = [o.clone() for o in decode_step()]
        tlog, tune_error = [], None
        if rank == 0 and saved_tuning is not None and saved_tuning.get("decode_tuning") is not None:
            # the step's choices as an earlier run of this command measured them (no tuning launches in this run)
            saved_applied[0] = max(saved_applied[0], ops.import_tuning(saved_tuning["entries"], strict=False))
            decode_tuning = dict(saved_tuning["decode_tuning"], source=str(tpath))
        elif rank == 0:
            try:
                before, after = ops.tune_step(decode_step, reps=9, burst=4, passes=2, log=tlog)
            except Exception as e:                   # noqa: BLE001 -- e.g. the step's stream-K hand-offs do not hold on a shared device: the
                before = after = None                # library is on its static schedules from here on (same bits) and the line says so
                tlog = [dict(rolled_back=True)]
                tune_error = f"{type(e).__name__}: {e}"
            decode_tuning = dict(step_ms_before=None if before is None else round(before, 4), step_ms_after=None if after is None else round(after, 4),
                                 chosen=[dict(layer=r["layer"], shape=r["shape"], start=r["start"], chosen=r["chosen"]) for r in tlog if r.get("chosen")],
                                 rolled_back=any(r.get("rolled_back") for r in tlog), **({"error": tune_error} if tune_error else {}))
        if world > 1:
            choices = D.share_from_rank0(ops.export_tuning() if rank == 0 else None)
            if rank != 0:
                ops.import_tuning(choices)
        for got, ref in zip(decode_step(), want_px):
            assert torch.equal(got, ref), "a tuned schedule changed the decoded pixels"
        del want_px
        tune_seconds[0] += time.perf_counter() - t0
    # The headline follows the contract to the letter: W warm-up steps, then K timed steps -- nothing else in front of them.  (Round 5
    # ran 40 more untimed steps first and still reported `warmup: W`: ADVICE r5.  The device reaches its sustained clock only after
    # ~40 ms of uninterrupted load -- behind a tuning pass whose samples end in host synchronisations the first eight timed steps ran
    # 3.54, 3.33, 3.27, 3.23, 3.19, 3.14, 3.12, 3.10 ms before settling at 3.09 -- so that ramp is now INSIDE `value`.)
    hst = {}
    t_dec = timed(decode_step, args.steps, args.warmup, hst, headline=True)   # the headline: exactly --steps steps after --warmup
    ms_per_step = 1e3 * t_dec / args.steps
    ms_by_rank = [round(1e3 * w / args.steps, 4) for w in rank_walls]   # each rank's own clock over the same K steps (value uses the max)
    value = world * pixels_per_step * args.steps / t_dec / 1e6
    # ... and the sustained rate is reported NEXT to it, under its own name: PREHEAT + W untimed steps, then K timed ones
    PREHEAT = 0 if args.no_preheat else 40
    sustained = None
    if PREHEAT:
        sst = {}
        t_sus = timed(decode_step, args.steps, PREHEAT + args.warmup, sst)
        sustained = dict(value=round(world * pixels_per_step * args.steps / t_sus / 1e6, 2), unit="Mpixel/s",
                         ms_per_step=round(1e3 * t_sus / args.steps, 4), warmup=PREHEAT + args.warmup, steps=args.steps,
                         median_ms=sst.get("median_ms"), min_ms=sst.get("min_ms"),
                         note="the same timed region after a longer warm-up (the device's sustained clock); `value` is the contract's W-warm-up figure")
    e2e_value, table = None, None

    def region(fn, steps, px, frac_fn=None, **extra):
        """One measured region: R_WARM warm-up steps, ``steps`` timed ones; rate from the median step (HIP events)."""
        st = {}
        t = timed(fn, steps, R_WARM, st)
        med = st["median_ms"] * 1e-3
        out = dict(extra, ms_per_step=st["median_ms"], mpixels_per_s=round(world * px / med / 1e6, 2), timing=st,
                   roofline=region_frac(frac_fn or fn, med))
        return out, t / steps

    regions = {}
    regions["decode"], _ = region(decode_step, R_STEPS, pixels_per_step, decode_eager)
    if not args.decode_only:
        regions["encode"], _ = region(enc_fn, R_STEPS_ENC, pixels_per_step)
        regions["encode_decode_score"], t_e2e_step = region(e2e_step, R_STEPS_ENC, pixels_per_step)
        e2e_value = regions["encode_decode_score"]["mpixels_per_s"]
        regions["w1_encode_decode_score"], _ = region(w1_step, R_STEPS_ENC, 64 * 256 * 256, workload="64 x 256x256 per GPU")
        # training step at the reference's training shape (two_layer_syn.py:13-15: batch 8 x 256 x 256 per
        # replica); data-parallel: with N > 1 the bucketed gradient all-reduce over RCCL is inside the timed region
        from shallow_ntc_amd.train import Trainer
        train_cfg = configs.two_layer_syn(rd_lambda=0.08)
        train_cfg["optimizer_config"] = dict(learning_rate=1e-4, global_clipnorm=1.0)
        train_model = Model(device=dev, **train_cfg)
        trainer = Trainer(train_model, seed=rank)
        train_x = synthetic_batch(8, 256, 256, 777 + rank, dev)
        st = {}
        tune(lambda: trainer.train_step(train_x))
        timed(lambda: trainer.train_step(train_x), R_STEPS_ENC, R_WARM, st)
        del trainer, train_model
        regions["train_step"] = dict(workload="8 x 256x256 per GPU, unoise, Adam + global_clipnorm, gradient all-reduce when N > 1",
                                     ms_per_step=st["median_ms"], images_per_s=round(world * 8 / (st["median_ms"] * 1e-3), 1),
                                     mpixels_per_s=round(world * 8 * 256 * 256 / (st["median_ms"] * 1e-3) / 1e6, 2), timing=st)
        # the reference's own call pattern (mshyper/models.py:425-433, eval.py): evaluate() one image at a time -- encode,
        # rate, decode, PSNR (and MS-SSIM) per image.  "serial" synchronises after every image like the reference's eager
        # loop; the default launches same-shaped images among the next 16 four at a time and keeps 4 launches in flight on
        # round-robin streams (each image's own numbers, in input order).
        singles = [x[i:i + 1].contiguous() for _ids, x, _hw in batches for i in range(x.shape[0])]
        if len(singles) == 24:             # Kodak's own order: kodim04, 09, 10, 17, 18, 19 are the portrait images
            land, port = singles[:18], singles[18:]
            singles = [port.pop(0) if i in (3, 8, 9, 16, 17, 18) else land.pop(0) for i in range(24)]
        b1 = {}
        keep_q = model._quality_metrics
        model._quality_metrics = False
        tune(lambda: [list(model.evaluate(singles, lookahead=look)) for look in (1, 3)])      # the one- ... eight-image launch shapes
        for label, quality, look in (("psnr_only_serial", False, 1), ("psnr_only", False, 3), ("with_msssim_serial", True, 1),
                                     ("with_msssim", True, 3)):
            model._quality_metrics = quality
            st = {}
            timed(lambda: list(model.evaluate(singles, lookahead=look)), 3, 1, st)         # median of three passes: one pass alone
            t = st["median_ms"] * 1e-3                                                      # now and then carries a host hiccup
            b1[label] = dict(ms_per_image=round(1e3 * t / len(singles), 3), mpixels_per_s=round(world * pixels_per_step / t / 1e6, 2))
        model._quality_metrics = keep_q
        regions["evaluate_b1"] = dict(workload=f"{len(singles)} images per GPU, one at a time through Model.evaluate()", **b1)
        # Split precision, opt-in (Model(precision="bf16x3")): the same decode / encode with the convolutions that
        # qualify on the pre-split bf16 x 3 kernel.  Reported next to -- never instead of -- the fp32 numbers (`value` and `dtype`
        # stay exact fp32), with its own parity figures against the fp32 model on the same codes.
        model3 = Model(device=dev, precision="bf16x3", **cfg)
        model3.set_weights(model.get_weights())
        dec3 = lambda: on_codes(lambda c: model3.decode(c[0], c[1], c[2], check=False))
        r3, _ = region(dec3, R_STEPS, pixels_per_step, lambda: [model3.decode(z_hat, sym, hw, check=False) for z_hat, sym, hw, _x in codes])
        diff = tot = 0
        sse32 = sse3 = 0
        for z_hat, sym, hw, x in codes:
            p32, s32 = model.decode(z_hat, sym, hw, reference=x, check=False)
            p3, s3 = model3.decode(z_hat, sym, hw, reference=x, check=False)
            d = (p32.to(torch.int16) - p3.to(torch.int16)).abs()
            diff += int((d != 0).sum()); tot += d.numel()
            assert int(d.max()) <= 1, "bf16 x 3 decode differs from fp32 by more than one code value"
            sse32 += int(s32.sum()); sse3 += int(s3.sum())
        psnr = lambda sse: 10.0 * np.log10(255.0 ** 2 / (sse / tot))
        regions["decode_bf16x3"] = dict(
            r3, mode="opt-in split precision: 3 bf16 terms per fp32 operand (activations and weights pre-split), 6 cross products on "
                     "v_mfma_f32_32x32x16_bf16, fp32 accumulate; csrc/bf3_gemm.hip",
            speedup_over_fp32=round(regions["decode"]["ms_per_step"] / r3["ms_per_step"], 3), pixel_values=tot,
            pixels_differing_from_fp32=diff, max_code_difference=1 if diff else 0,
            d_psnr_vs_fp32_db=round(float(psnr(sse3) - psnr(sse32)), 7))
        enc3 = lambda: on_streams(lambda b: model3.encode(b[1], check=False))
        tune(enc3)
        r3e, _ = region(enc3, R_STEPS_ENC, pixels_per_step)
        sym_diff = sym_tot = 0
        for ids, xb, hw in batches:
            _, s_a, _, _ = model.encode(xb, check=False)
            _, s_b, _, _ = model3.encode(xb, check=False)
            sym_diff += int((s_a != s_b).sum()); sym_tot += s_a.numel()
        regions["encode_bf16x3"] = dict(r3e, speedup_over_fp32=round(regions["encode"]["ms_per_step"] / r3e["ms_per_step"], 3),
                                        symbols=sym_tot, symbols_differing_from_fp32=sym_diff)
        del model3
        # ---- SGA iterative inference (row a21, BASELINE configs[4]: two_layer_syn2 + itinf, common/itinf_lib.py:26-93): one step
        # = Gumbel-softmax rounding of (z, y), hyper-synthesis + synthesis forward, analytic input gradients, Adam on the latents.
        # No host synchronisation inside the step (the loop fetches metrics only where it logs them).
        sga_cfg = {**configs.two_layer_syn2(rd_lambda=0.02, hidden_channels=24), **configs.itinf()}
        sga_model = Model(device=dev, quality_metrics=False, **sga_cfg)
        sga = {}
        for label, (bn, bh, bw) in (("tecnick_5x1200x1200", (5, 1200, 1200)), ("kodak_1x512x768", (1, 512, 768))):
            xb = synthetic_batch(bn, bh, bw, 555 + rank, dev)
            sga_model.initialize_itinf(xb)
            step_fn = lambda: sga_model.itinf_train_step(xb, fetch=False)
            tune(step_fn)
            st = {}
            timed(step_fn, R_STEPS, R_WARM, st)
            sga_model.itinf_last_metrics()                      # one fetch after the loop: raises if anything was flagged / non-finite
            med = st["median_ms"] * 1e-3
            sga[label] = dict(ms_per_step=st["median_ms"], seconds_per_3000_steps=round(3000 * med, 2),
                              mpixels_per_s=round(world * bn * bh * bw / med / 1e6, 2), timing=st, roofline=region_frac(step_fn, med))
        regions["sga_step"] = dict(workload="mshyper/configs/two_layer_syn2.py (24 hidden channels) + itinf.py: one SGA step on the latents of a batch",
                                   **sga)
        del sga_model
        # ---- the real bitstream (row f2): images -> bytes (analysis, hyper path, rANS encode of z and y, host copy) and bytes ->
        # uint8 pixels (rANS decode of z, hyper-synthesis, rANS decode of y, synthesis).  The random-init model codes almost
        # nothing, so the hyper-synthesis bias is set as in tests/test_hip_bitstream.py to put the scales in a coding range.
        bs_model = Model(device=dev, **cfg)
        wts = dict(model.get_weights())
        bias = wts["hyper_synthesis/layer_2/bias"].copy()
        bias[320:] = np.random.default_rng(0).uniform(-1.0, 2.5, size=320)
        wts["hyper_synthesis/layer_2/bias"] = bias.astype(np.float32)
        bs_model.set_weights(wts)
        t0 = time.perf_counter()
        bs_model._get_codec()
        torch.cuda.synchronize()
        codec_setup_s = time.perf_counter() - t0
        blobs = []
        comp_fn = lambda: blobs.__setitem__(slice(None), bs_model.compress_many([x for _ids, x, _hw in batches]))
        tune(comp_fn)
        st_c, st_d = {}, {}
        timed(comp_fn, 5, 2, st_c)
        decomp_fn = lambda: bs_model.decompress_many(blobs)
        timed(decomp_fn, 5, 2, st_d)
        file_bpp = 8.0 * sum(len(b) for b in blobs) / pixels_per_step
        for label, st, fn in (("compress", st_c, comp_fn), ("decompress", st_d, decomp_fn)):
            med = st["median_ms"] * 1e-3
            regions[label] = dict(workload="the Kodak-shaped set, one bitstream per batch shape; " +
                                           ("images -> bytes on the host" if label == "compress" else "bytes on the host -> uint8 pixels"),
                                  ms_per_step=st["median_ms"], mpixels_per_s=round(world * pixels_per_step / med / 1e6, 2), timing=st,
                                  file_bpp=round(file_bpp, 4), table_build_s=round(codec_setup_s, 3),
                                  fraction_of_decode_value=round(world * pixels_per_step / med / 1e6 / value, 3) if label == "decompress" else None,
                                  roofline=region_frac(fn, med))
        del bs_model
        rows = []                          # per-image (bpp, psnr, mse) of the rank's set, then ONE all-gather
        for ids, x, hw in batches:
            for d, i in zip(model.evaluate_batched(x), ids):
                rows.append((i, d["bpp"], d["psnr"], d["mse"]))
        rows.sort()
        table = D.gather_rows([r[1:] for r in rows], [r[0] + rank * len(shapes) for r in rows], world * len(shapes),
                              device=dev)

    # ---- roofline of the dominant kernel: HIP events on the launch stream, per launch ---------------
    roofline = None

    def kernel_table(fn, reps):
        per_kernel = {}
        for _rep in range(reps):
            ops.PROFILE = []
            fn()
            torch.cuda.synchronize()
            for e in ops.PROFILE:
                vec = "true, false" if e["vec"] else "false, true"      # <TM,TN,WM,WN,VEC,PRO> as rocprof prints it
                shape = {8: "1, 1, 2, 2", 9: "2, 2, 2, 2", 10: "2, 4, 4, 1"}.get(e["variant"], f"1, {e['variant']}, 4, 1")
                # <..., BF3, DMA, DEEP, FUSE2>: the default fp32 register-staged instance, or the fused ResidualBlock tail
                name = f"gg_kernel<{shape}, {vec}, false, false, 0, {'true' if '+1x1' in e['kind'] else 'false'}>"
                if e.get("colm"):                                           # the column-major stream-K twin (template argument COLM)
                    name = name[:-1] + ", true>"
                if e["variant"] >= 11:                                      # the pre-split bf16 x 3 kernel (csrc/bf3_gemm.hip)
                    name = f"bf3_kernel<4, 2, 2, {4 if e['variant'] == 11 else 2}>"
                if e["kind"] == "tail":                                     # the two-layer syntheses' output layer (csrc/pixel.hip; VALU, not MFMA)
                    name = f"two_layer_tail_kernel<{e['cin']}>"
                if e["kind"] == "synthesis":                                # the fused two-layer synthesis (csrc/syn_fused.hip)
                    name = f"syn_kernel<{e['cout']}, {'true' if e['cout'] > 12 and e['cout'] % 24 == 0 and model._synthesis._has_res else 'false'}>"
                if e["kind"] == "resblock":                                 # the whole ResidualBlock in one launch (csrc/rb_fused.hip)
                    name = f"rb_kernel<{e['cin']}>"
                if e["kind"] == "resblock3":                                # ... in bf16 x 3 (csrc/rb_fused_bf3.hip)
                    name = f"rb3_kernel<{e['cin']}>"
                k = per_kernel.setdefault(name, dict(ms=0.0, flops=0, launches=0, what=[]))
                what = f"{e['kind']} k{e['k']} s{e['s']} {e['cin']}->{e['cout']} @ {e.get('n', '?')}x{e['h']}x{e['w']}"
                if what not in k["what"]:
                    k["what"].append(what)
                k["ms"] += e["e0"].elapsed_time(e["e1"])
                k["flops"] += e["flops"]
                k["launches"] += 1
            ops.PROFILE = None
        return per_kernel

    if rank == 0:
        per_kernel = kernel_table(decode_eager, 3)
        name, k = max(per_kernel.items(), key=lambda kv: kv[1]["ms"])
        achieved = k["flops"] / (k["ms"] * 1e-3) / 1e12
        # HBM-side bytes per launch come from committed rocprofv3 --pmc passes (separate runs, never this one); the summary
        # records the hash of the kernel source it was measured on, so a number from an older kernel is marked stale
        import hashlib
        traffic, traffic_src, traffic_stale = None, None, None
        cur_sha = hashlib.sha256((ROOT / "shallow-ntc_amd/csrc/gather_gemm_kernel.h").read_bytes()).hexdigest()[:16]
        import re as _re
        for f in sorted(f for f in (ROOT / "profiles").glob("*_pmc_summary.json") if _re.fullmatch(r"r\d+_pmc_summary\.json", f.name))[::-1]:
            summ = json.loads(f.read_text())
            for kn, e in summ.items():
                if kn != "_meta" and name in kn and "hbm_side_bytes_per_launch" in e:
                    traffic, traffic_src = e["hbm_side_bytes_per_launch"], f"profiles/{f.name}"
                    traffic_stale = summ.get("_meta", {}).get("gather_gemm_sha16") != cur_sha
            if traffic is not None:
                break
        roofline = dict(bound="mfma", kernel=name, achieved=round(achieved, 2), peak=FP32_MFMA_PEAK_TFLOPS,
                        unit="TFLOP/s", frac=round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), traffic=traffic,
                        traffic_source=traffic_src, traffic_stale=traffic_stale,
                        frac_step=round(regions["decode"]["roofline"]["gflop_per_step"] / ms_per_step / FP32_MFMA_PEAK_TFLOPS, 4),
                        frac_step_region_median=regions["decode"]["roofline"]["frac_of_fp32_mfma_peak"],
                        frac_step_note="the whole decode step against the same peak: all convolution FLOPs of a step / the HEADLINE's own "
                                       "ms_per_step (frac_step_region_median: the same from the 50-step region's median step) -- `frac` "
                                       "describes the dominant kernel's launches only",
                        avg_launch_ms=round(k["ms"] / k["launches"], 4), launches_per_step=k["launches"] // 3,
                        precision="fp32 MFMA (v_mfma_f32_32x32x2_f32)",
                        launches=k["what"],
                        note=("dominant kernel by GPU time of the decode step, timed launch by launch on ONE stream (a kernel's duration is its "
                              "own); `launches` lists the layers that ran on it (the hyper-synthesis layers run on the column-major twin of the "
                              "128x128 instance, template argument COLM, a kernel of its own in rocprof's tables); the other kernels are in "
                              "all_kernels, the whole step's fraction is frac_step"),
                        all_kernels={n: dict(tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                             ms_per_step=round(v["ms"] / 3, 4), launches=v["what"]) for n, v in per_kernel.items()})

        if not args.decode_only:             # the same table for the analysis side (ELIC encoder: MFMA utilisation)
            enc = kernel_table(lambda: [model.encode(x, check=False) for _ids, x, _hw in batches], 1)
            tot_ms, tot_fl = sum(v["ms"] for v in enc.values()), sum(v["flops"] for v in enc.values())
            # counters of the same kernels from the committed encode-side passes (profiles/r*_encode_pmc_summary.json: separate
            # rocprofv3 --pmc runs of tools/profile_layers.py), with the same staleness flag as `traffic` above
            enc_pmc, enc_src, enc_stale = {}, None, None
            for f in sorted((ROOT / "profiles").glob("*_encode_pmc_summary.json"), reverse=True):
                enc_pmc, enc_src = json.loads(f.read_text()), f"profiles/{f.name}"
                rb_sha = hashlib.sha256((ROOT / "shallow-ntc_amd/csrc/rb_fused.hip").read_bytes()).hexdigest()[:16]
                meta = enc_pmc.get("_meta", {})
                enc_stale = meta.get("gather_gemm_sha16") != cur_sha or meta.get("rb_fused_sha16") != rb_sha
                break

            def counters(n):
                for kn, e in enc_pmc.items():
                    if kn != "_meta" and n in kn:
                        return dict(traffic=e.get("hbm_side_bytes_per_launch"), mfma_busy_over_simd_cycles=e.get("mfma_busy_over_simd_cycles"),
                                    lds_bank_conflict_cycles=e.get("counters_mean_per_launch", {}).get("SQ_LDS_BANK_CONFLICT"))
                return {}

            roofline["encode_kernels"] = dict(
                conv_tflops=round(tot_fl / (tot_ms * 1e-3) / 1e12, 2), frac=round(tot_fl / (tot_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                conv_ms_per_step=round(tot_ms, 3), counters_source=enc_src, counters_stale=enc_stale,
                by_kernel={n: dict(tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2), ms_per_step=round(v["ms"], 3),
                                   launches=v["launches"], **counters(n))
                           for n, v in sorted(enc.items(), key=lambda kv: -kv[1]["ms"])})

    # ---- CPU baseline: the torch-CPU port of the same decode on this box's host cores ---------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.decode_only:
        from oracle import model_np, torch_ref
        ref_model = model_np.Model(cfg["transform_config"], rd_lambda=cfg["rd_lambda"])
        weights = model.get_weights()
        host_codes = [(z.cpu().numpy(), s.cpu().numpy().astype(np.float32), hw) for z, s, hw, _ in codes]
        n_img = sum(z.shape[0] for z, _, _ in host_codes)
        times = []
        for p in range(args.cpu_passes + 1):
            t0 = time.perf_counter()
            for z, s, hw in host_codes:
                torch_ref.decode(ref_model, weights, z, s, hw)
            if p > 0:
                times.append(time.perf_counter() - t0)
        t_cpu = float(np.median(times))
        # the encoder side on a 2-image sample (the ELIC analysis is 12x the decoder's FLOPs: the full set would take minutes)
        from oracle import transforms_np as T
        enc_x = batches[0][1][:2].cpu().numpy()
        enc_times = []
        for p in range(args.cpu_passes + 1):
            t0 = time.perf_counter()
            yc = torch_ref.analysis_only(ref_model, weights, enc_x)
            with torch.no_grad():
                ref_model.hyper_analysis(T.sub_params(weights, "hyper_analysis/"), yc, be=torch_ref)
            if p > 0:
                enc_times.append(time.perf_counter() - t0)
        enc_px = enc_x.shape[0] * enc_x.shape[1] * enc_x.shape[2]
        cpu_model, phys = host_cpu()
        cpu_baseline = dict(value=round(pixels_per_step / t_cpu / 1e6, 3), unit="Mpixel/s", cores=torch.get_num_threads(),
                            cpu_model=cpu_model, physical_cores=phys, min_value=round(pixels_per_step / float(np.max(times)) / 1e6, 3),
                            max_value=round(pixels_per_step / float(np.min(times)) / 1e6, 3),
                            encode_transforms_value=round(enc_px / float(np.median(enc_times)) / 1e6, 3),
                            encode_sample=f"analysis + hyper-analysis of {enc_x.shape[0]} images of the set, median of "
                                          f"{args.cpu_passes} passes after 1 warm-up",
                            kind="port",
                            sample=f"{n_img} Kodak-shaped images decoded by oracle/torch_ref.py (float32, oneDNN, "
                                   f"{torch.get_num_threads()} threads), median of {args.cpu_passes} passes after 1 warm-up; "
                                   f"the reference's TF-CPU path cannot be installed here")

    if rank == 0:
        rd = None
        if table is not None:
            tbl = table[~np.isnan(table[:, 0])]
            rd = dict(bpp=round(float(tbl[:, 0].mean()), 5), psnr=round(float(tbl[:, 1].mean()), 4), images=int(tbl.shape[0]),
                      note="untrained random-init weights: parity-checked numbers, not a trained R-D point")
        line = dict(
            metric="decode Mpixels/s + (bpp, PSNR) on Kodak, two_layer_syn",
            value=round(value, 2), unit="Mpixel/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
            ms_per_step=round(ms_per_step, 4), ms_per_step_by_rank=ms_by_rank, higher_is_better=True, scaling="weak", vs_baseline=None,
            dtype="f32", data="synthetic",
            config=dict(workload=f"mshyper/configs/two_layer_syn.py (ElicAnalysis 192,192,192,320 + TwoLayerResSynthesis 12,3), "
                                 + (f"W1 synthetic batch per GPU ({len(shapes)} x 256x256), " if args.workload == "w1" else
                                    f"W3 synthetic batch per GPU ({len(shapes)} x 1200x1200), " if args.workload == "w3" else
                                    f"Kodak-24-shaped synthetic set per GPU ({len(shapes)} images: 512x768 / 768x512), ") +
                                 "random-init weights", images_per_gpu=len(shapes), parallelism=f"dp{world}",
                        timed_region="decode: (z_hat, symbols) in HBM -> uint8 pixels",
                        untimed_before=f"the {args.warmup} warm-up steps only; the timed steps' own HIP-event statistics: median "
                                       f"{hst.get('median_ms')} ms, min {hst.get('min_ms')} ms",
                        timed_steps_ms=hst.get("per_step_ms"),
                        launch="hipGraph replay (one graph per batch shape)" if args.graph else
                        ("eager, Model.decode_set: hyper-syntheses of the batch shapes on concurrent streams, one synthesis launch for all" if set_decode else
                         f"eager, {min(nstreams, len(codes))} concurrent streams (one per batch)" if nstreams > 1 and len(codes) > 1 else "eager"),
                        codes="synthetic latents: z_hat ~ round(N(0,9)), symbols ~ round(Laplace(0,2))",
                        launch_schedule="cost model" if args.no_autotune else
                        (f"{saved_applied[0]} schedules of an earlier run of this command applied from {tpath.name} (fingerprint arch / workload / "
                         f"flags matched), the rest " if saved_tuning is not None else "") +
                        f"measured before the timed regions ({tune_seconds[0]:.1f} s untimed; every candidate computes the same chains, same "
                        "bits): once per layer shape with the device idle (sntc_conv_plan_tune) for the single-stream regions; the "
                        "two-stream Kodak decode of `value` by bursts of its own step (ops.tune_step, `decode_tuning`)"
                        + ("" if decode_tuning or rank != 0 or args.no_autotune else " -- switched off for this run: cost model / the encode pass's choices"),
                        decode_tuning=decode_tuning),
            encode_decode_mpixels_per_s=None if e2e_value is None else round(e2e_value, 2), sustained=sustained,
            regions=regions, rd=rd, roofline=roofline, cpu_baseline=cpu_baseline, rccl=world_info,
        )
        # this run's schedules as a file for later runs (the box's copy of profiles/ does not travel back: gpurun_out/ does)
        wpath = args.write_tuning or ("" if saved_tuning is not None or args.no_autotune else str(ROOT / "gpurun_out" / "tuning_gfx950.json"))
        if wpath:
            try:
                Path(wpath).parent.mkdir(parents=True, exist_ok=True)
                Path(wpath).write_text(json.dumps(dict(fingerprint=tune_fp, entries=ops.export_tuning(), decode_tuning=decode_tuning)))
            except OSError:
                pass
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    D.shutdown()


if __name__ == "__main__":
    main()
