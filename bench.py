#!/usr/bin/env python3
"""bench.py -- headline benchmark of the obia hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on; it fits one GPU):
  16384 x 16384 x 8-band float32 synthetic raster, create_tiled_segments(tile_size=2048, buffer=64,
  crown_radius=5, pixel size 0.5 m, all-ones mask, compactness=10)  +  zonal mean/var/min/max on all 8 bands.
One "step" = that whole pipeline once, input already resident in HBM.  value = H*W / step time (Mpixel/s).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c4] [--size S] [--no-cpu] [--no-side]
For N > 1 the driver launches one rank per GPU with torch.distributed.run (RCCL); --gpus must equal WORLD_SIZE.
  --config c3 (default): every rank owns one 16384-row slab (the N = 1 workload) of a (N*16384) x 16384 x 8 raster, and the
      slabs are segmented as ONE raster: halo rows and seam label rows travel by send/recv between neighbouring ranks
      (obia_amd/distributed.py), no collective on the data path.  The work per GPU is the same at every N, the N = 1 case
      included => "scaling": "weak".  (White tile rows run in two parity classes there -- the order that lets neighbouring
      slabs work at the same time -- so a rank segments four tile rows per batch instead of one.)
  --config c4: BASELINE configs[3], the raster is ALWAYS 32768 x 32768 x 8 (tile 2048, overlap 64): N = 1 segments it whole
      (34 GB of raster on one 288-GB GPU), N ranks take 32768/N rows each (8 GPUs: slabs of 4096 rows = 2 tile rows)
      => "scaling": "strong"; this is the case north_star's ">= 6x at 8 GPUs" speaks of.
Side legs at N = 1 (never `value`; --no-side skips them): the same workload at compactness 0.25 (the author's regime,
notebooks/deepfor.ipynb:402 -- at compactness 10 on [0,1] features the result is nearly a grid), and BASELINE configs[4]
(8192 x 8192 x 3 quickshift, kernel_size 5, max_dist 10).

The JSON line carries `roofline` (dominant kernel = the SLIC colour sweep slic_assign_kernel<8,true,false,false,false,8>:
algorithmic bytes (4*C + 4 = 36 B/pixel) x pixels per launch / launch time from HIP events on
the library's stream) and `cpu_baseline` (the C oracle -- a port, 1 thread -- on one 2048^2 tile of the same
workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy reaches


def synth_raster(H, W, C, seed, device, row0=0, host_rng=False):
    """BASELINE.md 3 / SURVEY 8d generator: band_c = 400 sin(x/(11+3c)) cos(y/(13+2c)) + 1000 + 50c + N(0,20^2), float32.
    The noise is drawn on the DEVICE by default (torch's Philox stream): SURVEY 8d words it as NumPy RandomState(seed) on the
    host, which takes ~1 minute for the 2.1e9 normals of the headline raster -- same formula, same distribution, another stream
    (DESIGN.md 4).  host_rng=True (--host-rng) draws with NumPy RandomState(seed), row block by row block, band by band, and
    uploads once, outside every timed region."""
    if host_rng:
        rs = np.random.RandomState(seed)
        out = torch.empty((H, W, C), device=device, dtype=torch.float32)
        xx = np.arange(W, dtype=np.float32)[None, :]
        for y0 in range(0, H, 2048):
            h = min(2048, H - y0)
            yy = np.arange(row0 + y0, row0 + y0 + h, dtype=np.float32)[:, None]
            blk = np.empty((h, W, C), np.float32)
            for c in range(C):
                blk[:, :, c] = 400.0 * np.sin(xx / (11 + 3 * c)) * np.cos(yy / (13 + 2 * c)) + 1000 + 50 * c + rs.normal(0, 20, (h, W))
            out[y0:y0 + h] = torch.from_numpy(blk).to(device)
        return out
    g = torch.Generator(device=device).manual_seed(seed)
    out = torch.empty((H, W, C), device=device, dtype=torch.float32)
    xx = torch.arange(W, device=device, dtype=torch.float32)[None, :]
    rows = 2048
    for y0 in range(0, H, rows):
        h = min(rows, H - y0)
        yy = torch.arange(row0 + y0, row0 + y0 + h, device=device, dtype=torch.float32)[:, None]
        for c in range(C):
            out[y0:y0 + h, :, c] = 400.0 * torch.sin(xx / (11 + 3 * c)) * torch.cos(yy / (13 + 2 * c)) + 1000 + 50 * c \
                + 20.0 * torch.randn((h, W), device=device, generator=g)
    return out


KERNEL_SOURCES = ("obia_amd/csrc/slic_sweep.hip", "obia_amd/csrc/slic.hpp", "obia_amd/csrc/slic.hip")


def kernel_source_sha256():
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


TRAFFIC_FILE = "r04_traffic.json"   # profiles/: the newest PMC traffic measurement (tools/traffic_json.py)


def traffic_fields(args, C, world):
    """roofline.traffic from profiles/<TRAFFIC_FILE> (written by tools/traffic_json.py from two rocprofv3 --pmc passes of
    this workload).  null unless the file was measured on exactly these kernel sources and this workload."""
    path = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
    none = {"traffic": None, "traffic_source": None}
    try:
        with open(path) as fh:
            t = json.load(fh)
    except Exception:
        return none
    same_src = t.get("kernel_source_sha256") == kernel_source_sha256()
    wl = t.get("workload", {})
    same_wl = (world == 1 and args.config == "c3" and wl.get("size") == args.size and wl.get("tile") == args.tile
               and wl.get("buffer") == args.buffer and wl.get("bands") == C and abs(wl.get("compactness", -1) - args.compactness) < 1e-12)
    if not (same_src and same_wl):
        return dict(none, traffic_source=f"profiles/{TRAFFIC_FILE} is for other {'sources' if not same_src else 'workload'} "
                                         f"(measured at commit {t.get('commit')}): not reported")
    k = t["kernels"]["slic_assign_colour"]
    return {"traffic": int(round(k["bytes_per_launch"])),
            "traffic_source": f"profiles/{TRAFFIC_FILE}: PMC FETCH_SIZE x2 + WRITE_SIZE per launch, measured at commit {t.get('commit')} "
                              f"on the same kernel sources ({k['bytes_per_pixel']:.1f} B/pixel vs 32.4 algorithmic)"}


def cpu_baseline(C, tile, buffer_, crown_radius, pixel, compactness):
    """The oracle (C port of the reference's arithmetic, single thread) on ONE tile of the same workload:
    normalise -> maskSLIC structure (pre-pass + 10 sweeps) -> connectivity -> zonal statistics."""
    from oracle import oracle as orc
    import math
    orc.build()
    rs = np.random.RandomState(0)
    H = W = tile
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    raw = np.empty((H, W, C), np.float32)
    for c in range(C):
        raw[:, :, c] = 400.0 * np.sin(xx / (11 + 3 * c)) * np.cos(yy / (13 + 2 * c)) + 1000 + 50 * c + rs.normal(0, 20, (H, W))
    mask = np.ones((H, W), np.uint8)
    n = round(H * W * pixel * pixel / (math.pi * crown_radius ** 2))
    t0 = time.time()
    lab = orc.slic(orc.normalize(raw), n_segments=n, compactness=compactness, mask=mask)
    orc.zonal_stats_c(raw, lab, n_labels=int(lab.max()))
    dt = time.time() - t0
    out = {"value": H * W / dt / 1e6, "unit": "Mpixel/s", "cores": 1, "kind": "port",
           "sample": f"one {tile}x{tile}x{C} tile of the workload (all-ones mask, n_segments={n}): normalise + "
                     f"spatial pre-pass + 10 SLIC sweeps + connectivity + zonal stats, {dt:.1f} s on 1 thread; the port seeds with the "
                     "build's masked-grid rule (DESIGN.md 5) -- scikit-image's own _get_mask_centroids (kmeans2 + a K x K pdist at "
                     f"K = {n}) would add minutes and gigabytes on top"}
    # the fair "all host cores" figure: the same tile in T independent processes (no GPU in the children)
    try:
        import subprocess
        T = max(1, min(os.cpu_count() or 1, 16))
        code = ("import sys, time, math, numpy as np; sys.path.insert(0, %r); from oracle import oracle as orc; "
                "H = W = %d; C = %d; rs = np.random.RandomState(int(sys.argv[1])); "
                "yy, xx = np.mgrid[0:H, 0:W].astype(np.float32); raw = np.empty((H, W, C), np.float32)\n"
                "for c in range(C): raw[:, :, c] = 400.0 * np.sin(xx / (11 + 3 * c)) * np.cos(yy / (13 + 2 * c)) + 1000 + 50 * c + rs.normal(0, 20, (H, W))\n"
                "mask = np.ones((H, W), np.uint8); norm = orc.normalize(raw); sys.stdin.readline(); t0 = time.time(); "
                "lab = orc.slic(norm, n_segments=%d, compactness=%r, mask=mask); orc.zonal_stats_c(raw, lab, n_labels=int(lab.max())); "
                "print(time.time() - t0)") % (ROOT, tile, C, n, compactness)
        procs = [subprocess.Popen([sys.executable, "-c", code, str(i)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
                 for i in range(T)]
        time.sleep(0.1)
        for pr in procs:      # every child has built its tile and waits on stdin: release them together
            pr.stdin.write("go\n"); pr.stdin.flush()
        dts = [float(pr.communicate(timeout=600)[0].strip().splitlines()[-1]) for pr in procs]
        out["all_cores"] = {"value": T * H * W / max(dts) / 1e6, "unit": "Mpixel/s", "cores": T,
                            "sample": f"{T} such tiles in {T} processes at once, slowest {max(dts):.1f} s"}
    except Exception as e:   # the single-thread figure stands on its own
        out["all_cores"] = {"error": str(e)}
    out["library"] = library_baseline(C, tile, n, compactness)
    return out


def library_baseline(C, tile, n, compactness):
    """scikit-image itself -- the library whose `slic` the reference's create_segments calls (segment_boundaries.py:48-51) -- on the
    same tile, when an interpreter that has it is on the box (the build image's /opt/conda/bin/python3.9: scikit-image 0.18.3; the
    bench's own interpreter has none).  Segmentation only and WITHOUT the mask: with one, scikit-image first seeds by k-means
    (_get_mask_centroids) over a K x K distance matrix, minutes and gigabytes at this K.  A reported figure beside the port's, nothing
    is derived from it."""
    import subprocess
    py = os.environ.get("OBIA_SKIMAGE_PYTHON", "/opt/conda/bin/python3.9")
    if not os.path.exists(py):
        return None
    code = ("import time, warnings, numpy as np; warnings.filterwarnings('ignore'); import skimage; from skimage.segmentation import slic; "
            "H = W = %d; C = %d; rs = np.random.RandomState(0); yy, xx = np.mgrid[0:H, 0:W].astype(np.float32); img = np.empty((H, W, C), np.float32)\n"
            "for c in range(C):\n"
            "    b = 400.0 * np.sin(xx / (11 + 3 * c)) * np.cos(yy / (13 + 2 * c)) + 1000 + 50 * c + rs.normal(0, 20, (H, W)); img[:, :, c] = (b - b.min()) / (b.max() - b.min())\n"
            "t0 = time.time(); lab = slic(img, n_segments=%d, compactness=%r, max_iter=10, sigma=0, multichannel=True, convert2lab=False, start_label=1); "
            "print(skimage.__version__, time.time() - t0, len(np.unique(lab)))") % (tile, C, n, compactness)
    try:
        r = subprocess.run([py, "-c", code], capture_output=True, text=True, timeout=180)
        ver, dt, nseg = r.stdout.strip().splitlines()[-1].split()
        return {"value": tile * tile / float(dt) / 1e6, "unit": "Mpixel/s", "cores": 1, "what": f"skimage.segmentation.slic {ver}",
                "sample": f"one {tile}x{tile}x{C} tile, n_segments={n}, no mask, segmentation only (normalised bands in, labels out): {float(dt):.1f} s, {nseg} segments"}
    except Exception as e:
        return {"error": str(e)[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=16384, help="raster side at N=1 (default: BASELINE configs[2])")
    ap.add_argument("--tile", type=int, default=2048)
    ap.add_argument("--buffer", type=int, default=64)
    ap.add_argument("--bands", type=int, default=8)
    ap.add_argument("--compactness", type=float, default=10.0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-side", action="store_true", help="skip the side legs (exit_on_fixed_point, compactness 0.25, quickshift, configs[3] whole)")
    ap.add_argument("--host-rng", action="store_true", help="draw the raster's noise with NumPy RandomState on the host (SURVEY 8d's wording; ~1 min)")
    ap.add_argument("--config", choices=("c3", "c4"), default="c3",
                    help="c3: BASELINE configs[2] per GPU (weak scaling); c4: BASELINE configs[3], 32768^2 x 8 split over the ranks (strong)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # checked before anything touches the GPU: a bench keyed on --gpus must never report a run that did not shard
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}.  For N > 1 launch one rank per GPU:\n"
                         f"  python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
                         f"--master-port 29500 bench.py --gpus {args.gpus} --steps {args.steps} --warmup {args.warmup}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: obia_amd has no CPU path")
    # OBIA_BENCH_BACKEND=gloo rehearses the N > 1 path with several ranks on ONE card (seam rows staged through
    # the host); the real runs use RCCL ("nccl") with one rank per GPU.
    backend = os.environ.get("OBIA_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if world > 1 and backend == "nccl" and world > n_dev:
        # RCCL needs one device per rank; ranks stacked on a card are a rehearsal and must say so (OBIA_BENCH_BACKEND=gloo)
        raise SystemExit(f"bench.py: {world} ranks over RCCL need {world} GPUs, this node shows {n_dev}.  To rehearse the sharded path "
                         "with several ranks on one card set OBIA_BENCH_BACKEND=gloo (seam rows staged through the host; not a scaling number)")
    gpu = local_rank % n_dev
    wire = "RCCL send/recv over xGMI" if backend == "nccl" else f"{backend} (REHEARSAL: seam rows staged through the host, {min(world, n_dev)} device(s) for {world} ranks)"
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from obia_amd import _lib
    from obia_amd.statistics import zonal_stats
    from obia_amd.tiling import create_tiled_segments
    from obia_amd.distributed import ShardedTiler

    C = args.bands
    if args.config == "c4":
        W = Hg = 32768 if args.size == 16384 else args.size     # --size scales the case down for rehearsals
        if Hg % (world * args.tile) != 0:
            raise SystemExit(f"--config c4: {Hg} rows do not split into {world} slabs of whole {args.tile}-row tile rows")
        H = Hg // world
        row0 = rank * H
        workload = (f"{Hg}x{W}x{C} create_tiled_segments(tile={args.tile}, overlap={args.buffer}) + zonal stats (BASELINE configs[3])"
                    + (f", {world} slabs of {H} rows, seam exchange over {wire}" if world > 1 else ", whole raster on one GPU"))
    elif world == 1:
        H = W = args.size
        workload = f"{H}x{W}x{C} create_tiled_segments(tile={args.tile}, overlap={args.buffer}) + zonal stats (BASELINE configs[2])"
        row0 = 0
    else:
        W = args.size
        H = args.size           # every GPU gets a slab of the size of the N = 1 raster: fixed work per GPU
        row0 = rank * H
        workload = (f"{world * H}x{W}x{C} raster sharded over {world} GPUs ({H}-row slab per GPU = the N = 1 workload), "
                    f"tile={args.tile}, overlap={args.buffer}, seam exchange over {wire}")
    if world == 1:
        img = synth_raster(H, W, C, seed=rank, device=dev, row0=row0, host_rng=args.host_rng)
        mask = torch.ones((H, W), dtype=torch.uint8, device=dev)
        ext_img = ext_mask = None
    else:
        # the slab lives in the middle of rasters that have room for the neighbours' halo rows: the sharded driver then
        # copies nothing per call
        top, bot = ShardedTiler.halo_rows(rank, world, args.buffer)
        ext_img = torch.empty((top + H + bot, W, C), dtype=torch.float32, device=dev)
        ext_img[top:top + H] = synth_raster(H, W, C, seed=rank, device=dev, row0=row0, host_rng=args.host_rng)   # per-GPU slab seed = slab index
        ext_mask = torch.ones((top + H + bot, W), dtype=torch.uint8, device=dev)
        img, mask = ext_img[top:top + H], ext_mask[top:top + H]
    ctx = _lib.Context(gpu)
    ctx.set_profiling(1)
    kw = dict(tile_size=args.tile, buffer=args.buffer, crown_radius=5, pixel_size=(0.5, 0.5), compactness=args.compactness, ctx=ctx)

    def step(profile=False):
        if world == 1:
            lab, n = create_tiled_segments(img, input_mask=mask, **kw)
            t_seg = ctx.timing()
            st = zonal_stats(img, lab, n_labels=n, ctx=ctx)
        else:
            t = ShardedTiler(img, mask, world * H, H // args.tile, args.tile, args.buffer, 5, (0.5, 0.5), ctx=ctx,
                             ext_image=ext_img, ext_mask=ext_mask, compactness=args.compactness, profile=profile)
            lab, n = t.run()
            halo_img, dense, n_owned = t.owned_labels()
            t.clock.lap("owned_labels_ms")
            t.close()
            t_seg = ctx.timing()
            st = zonal_stats(halo_img, dense, n_labels=n_owned, ctx=ctx)   # every segment counted once, by its owner
            if profile:
                torch.cuda.synchronize()
                t.clock.lap("zonal_ms")
                phase_ms.clear()
                phase_ms.update(t.clock.ms)
        t_z = ctx.timing()
        return lab, n, st, t_seg, t_z

    phase_ms = {}   # N > 1: host clock per phase of the sharded call, from the breakdown step after the timed region

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_fixed_point_leg():
        """Same workload with exit_on_fixed_point=True (bit-identical labels; sweeps of converged tiles skipped).
        Reported beside the headline, never as `value`: the headline runs every one of the 10 + 10 sweeps."""
        if world != 1 or args.no_side:
            return None
        kw2 = dict(kw, exit_on_fixed_point=True)
        lab_a, n_a = create_tiled_segments(img, input_mask=mask, **kw)
        lab_b, n_b = create_tiled_segments(img, input_mask=mask, **kw2)
        same = bool(n_a == n_b and torch.equal(lab_a, lab_b))
        del lab_a
        torch.cuda.synchronize()
        t1 = time.time()
        px_eval = px_all = pre_ms = asg_ms = 0.0
        for _ in range(args.steps):
            lab_b, n_b = create_tiled_segments(img, input_mask=mask, **kw2)
            tt = ctx.timing()
            px_eval += tt["assign_px"] + tt["prepass_px"]
            pre_ms += tt["prepass_ms"]
            asg_ms += tt["assign_ms"]
            zonal_stats(img, lab_b, n_labels=n_b, ctx=ctx)
        torch.cuda.synchronize()
        d = (time.time() - t1) / args.steps
        return {"value": round(float(H) * W / d / 1e6, 2), "unit": "Mpixel/s", "ms_per_step": round(d * 1e3, 3),
                "labels_identical_to_full_sweeps": same, "prepass_ms": round(pre_ms / args.steps, 3),
                "assign_ms": round(asg_ms / args.steps, 3),
                "pixel_sweeps_evaluated_per_step": round(px_eval / args.steps)}

    # An event pair costs ~2.5 us of stream time; with every kernel class bracketed a step records ~420 pairs = 1.1 ms (2 %).
    # The timed steps therefore bracket only the launches of the dominant kernel, the colour sweep -- what the roofline needs,
    # measured over the whole timed region; the per-stage breakdown comes from one more step AFTER the timed region (all
    # classes on, like the side legs: never part of `value`).
    parts = {"features_ms": 0.0, "prepass_ms": 0.0, "assign_ms": 0.0, "connectivity_ms": 0.0, "zonal_ms": 0.0}
    parts_steps = 0
    for _ in range(args.warmup):
        step()
    ctx.set_profiling(2)
    barrier()
    t0 = time.time()
    assign_ms = assign_px = assign_store_px = sweeps = assign_busy_ms = 0.0
    n_seg = 0
    step_wall = []          # host clock after each step's calls returned (no extra synchronisation: the zonal kernels may still run)
    for _ in range(args.steps):
        lab, n_seg, st, t_seg, t_z = step()
        step_wall.append(time.time())
        assign_ms += t_seg["assign_ms"]
        assign_busy_ms += t_seg["assign_busy_ms"]
        assign_px += t_seg["assign_px"]
        assign_store_px += t_seg["assign_store_px"]
        sweeps += t_seg["sweeps"]
    barrier()
    dt = time.time() - t0
    ctx.set_profiling(1)
    _, _, _, t_seg, t_z = step(profile=True)   # the breakdown step (untimed)
    for k in ("features_ms", "prepass_ms", "assign_ms", "connectivity_ms"):
        parts[k] += t_seg[k]
    parts["zonal_ms"] += t_z["zonal_ms"]
    parts_steps += 1
    barrier()
    if dist is not None:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    phases_all = None
    if dist is not None:
        # every rank's phase times of the breakdown step (a device synchronisation at every phase boundary: they add up to the call,
        # which is therefore a little slower than a timed step): the line of the first real multi-GPU run has to explain itself
        keys = ["halo_ms", "black_ms", "import_ms", "class0_ms", "class1_ms", "export_ms", "relabel_ms", "owned_labels_ms", "zonal_ms"]
        mine = torch.tensor([phase_ms.get(k, 0.0) for k in keys], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        allp = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        tot = [float(v.sum().item()) for v in allp]
        slow = max(range(world), key=lambda r: tot[r])
        phases_all = {"source": "one sharded call after the timed region with the device synchronised at every phase boundary (host clock)",
                      "slowest_rank": slow, "slowest_rank_ms": {k: round(float(allp[slow][i].item()), 3) for i, k in enumerate(keys)},
                      "max_over_ranks_ms": {k: round(max(float(v[i].item()) for v in allp), 3) for i, k in enumerate(keys)},
                      "total_ms_per_rank": [round(x, 3) for x in tot]}
    fp_leg = timed_fixed_point_leg()

    def compactness_leg(c):
        """The headline workload at another compactness (SURVEY 8d asks for 0.25, the author's regime, beside 10)."""
        if world != 1 or args.no_side:
            return None
        kw2 = dict(kw, compactness=c)
        create_tiled_segments(img, input_mask=mask, **kw2)
        torch.cuda.synchronize()
        t1 = time.time()
        a_ms = a_px = a_spx = sw = 0.0
        for _ in range(args.steps):
            lab_c, n_c = create_tiled_segments(img, input_mask=mask, **kw2)
            tt = ctx.timing()
            a_ms += tt["assign_ms"]; a_px += tt["assign_px"]; a_spx += tt["assign_store_px"]; sw += tt["sweeps"]
            zonal_stats(img, lab_c, n_labels=n_c, ctx=ctx)
        torch.cuda.synchronize()
        d = (time.time() - t1) / args.steps
        ach = (a_px * 4 * C + a_spx * 4) / (a_ms * 1e-3) / 1e9 if a_ms > 0 else 0.0
        return {"compactness": c, "value": round(float(H) * W / d / 1e6, 2), "unit": "Mpixel/s", "ms_per_step": round(d * 1e3, 3),
                "segments": int(n_c), "sweep_avg_launch_ms": round(a_ms / max(1.0, sw), 4),
                "sweep_roofline_frac": round(ach / HBM_PEAK_GBS, 4)}

    def quickshift_leg():
        """BASELINE configs[4]: 8192 x 8192 x 3, quickshift(kernel_size=5, max_dist=10), tie noise drawn on the device."""
        if world != 1 or args.no_side:
            return None
        from obia_amd.segmentation import quickshift
        S = 8192 if args.size >= 8192 else args.size
        rgb = synth_raster(S, S, 3, seed=5, device=dev)
        mn, mx = rgb.amin(dim=(0, 1)), rgb.amax(dim=(0, 1))
        rgb = ((rgb - mn) / (mx - mn)).contiguous()      # values in [0, 1], treated as sRGB
        quickshift(rgb, kernel_size=5, max_dist=10, ratio=1.0, rng="device", ctx=ctx)
        torch.cuda.synchronize()
        reps = max(1, min(args.steps, 3))
        t1 = time.time()
        for _ in range(reps):
            ql = quickshift(rgb, kernel_size=5, max_dist=10, ratio=1.0, rng="device", ctx=ctx)
        torch.cuda.synchronize()
        d = (time.time() - t1) / reps
        return {"workload": f"{S}x{S}x3 quickshift(kernel_size=5, max_dist=10) (BASELINE configs[4])", "value": round(S * S / d / 1e6, 2),
                "unit": "Mpixel/s", "ms_per_call": round(d * 1e3, 2), "segments": int(ql.max().item()) + 1, "dtype": "f64"}

    def c4_leg():
        """BASELINE configs[3] WHOLE on this one GPU (32768 x 32768 x 8, tile 2048, overlap 64: 34 GB of raster), the same pipeline
        as the headline.  The 8-GPU form of this configuration is `--config c4 --gpus 8` (strong scaling); this leg is what one
        GPU does with the whole of it, so that the driver's default line carries a measured figure for the configuration."""
        if world != 1 or args.no_side or args.config != "c3" or args.size != 16384:
            return None
        S = 32768
        big = synth_raster(S, S, C, seed=0, device=dev)
        bmask = torch.ones((S, S), dtype=torch.uint8, device=dev)
        ctx4 = _lib.Context(gpu)
        ctx4.set_profiling(2)
        kw4 = dict(kw, ctx=ctx4)
        for _ in range(2):   # warm-up: the first call sizes the workspace, the second merges its blocks into one (context.cpp: Arena)
            lab4, n4 = create_tiled_segments(big, input_mask=bmask, **kw4)
            zonal_stats(big, lab4, n_labels=n4, ctx=ctx4)
        torch.cuda.synchronize()
        reps = max(1, min(args.steps, 3))
        a_ms = a_px = a_spx = sw = 0.0
        t1 = time.time()
        for _ in range(reps):
            lab4, n4 = create_tiled_segments(big, input_mask=bmask, **kw4)
            tt = ctx4.timing()
            a_ms += tt["assign_ms"]; a_px += tt["assign_px"]; a_spx += tt["assign_store_px"]; sw += tt["sweeps"]
            zonal_stats(big, lab4, n_labels=n4, ctx=ctx4)
        torch.cuda.synchronize()
        d = (time.time() - t1) / reps
        ach = (a_px * 4 * C + a_spx * 4) / (a_ms * 1e-3) / 1e9 if a_ms > 0 else 0.0
        out4 = {"workload": f"{S}x{S}x{C} create_tiled_segments(tile={args.tile}, overlap={args.buffer}) + zonal stats (BASELINE configs[3]), "
                            "whole raster on one GPU", "value": round(float(S) * S / d / 1e6, 2), "unit": "Mpixel/s",
                "ms_per_step": round(d * 1e3, 3), "steps": reps, "segments": int(n4),
                "sweep_avg_launch_ms": round(a_ms / max(1.0, sw), 4), "sweep_roofline_frac": round(ach / HBM_PEAK_GBS, 4)}
        del big, bmask, lab4
        ctx4.close()
        torch.cuda.empty_cache()
        return out4

    def bands_leg(nb):
        """The headline raster and tiling with another band count: 9 is what the author's rasters hold (notebooks/deepfor.ipynb);
        planes, records and accumulators come in groups of four channels, the kernels skip the padded ones."""
        if world != 1 or args.no_side or args.config != "c3" or C == nb:
            return None
        img9 = synth_raster(H, W, nb, seed=0, device=dev)
        ctx9 = _lib.Context(gpu)
        ctx9.set_profiling(2)
        kw9 = dict(kw, ctx=ctx9)
        for _ in range(2):   # (warm-up, as in c4_leg)
            lab9, n9 = create_tiled_segments(img9, input_mask=mask, **kw9)
            zonal_stats(img9, lab9, n_labels=n9, ctx=ctx9)
        torch.cuda.synchronize()
        reps = max(1, min(args.steps, 5))
        a_ms = a_px = a_spx = sw = 0.0
        t1 = time.time()
        for _ in range(reps):
            lab9, n9 = create_tiled_segments(img9, input_mask=mask, **kw9)
            tt = ctx9.timing()
            a_ms += tt["assign_ms"]; a_px += tt["assign_px"]; a_spx += tt["assign_store_px"]; sw += tt["sweeps"]
            zonal_stats(img9, lab9, n_labels=n9, ctx=ctx9)
        torch.cuda.synchronize()
        d = (time.time() - t1) / reps
        ach = (a_px * 4 * nb + a_spx * 4) / (a_ms * 1e-3) / 1e9 if a_ms > 0 else 0.0
        out9 = {"bands": nb, "value": round(float(H) * W / d / 1e6, 2), "unit": "Mpixel/s", "ms_per_step": round(d * 1e3, 3), "steps": reps,
                "segments": int(n9), "sweep_avg_launch_ms": round(a_ms / max(1.0, sw), 4), "sweep_roofline_frac": round(ach / HBM_PEAK_GBS, 4)}
        del img9, lab9
        ctx9.close()
        torch.cuda.empty_cache()
        return out9

    def next_rows_leg():
        """SURVEY 8f's rows that share the headline's inputs, on the headline raster and its label map: skewness / kurtosis (f2, the
        second pass over (labels, raster) of zonal_stats(moments=True): segment_statistics.py:173-175) and the GLCM texture statistics
        (f3, segment_statistics.py:179-298).  Times are whole calls (host clock around a synchronised call, median of three);
        bytes = the raster and the labels read once by the pass: 4C + 4 per pixel."""
        if world != 1 or args.no_side:
            return None
        from obia_amd.statistics import texture_stats
        lab_n, n_n = create_tiled_segments(img, input_mask=mask, **kw)

        def med(fn):
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t1 = time.time(); fn(); torch.cuda.synchronize(); ts.append(time.time() - t1)
            return sorted(ts)[1]
        t_stats = med(lambda: zonal_stats(img, lab_n, n_labels=n_n, ctx=ctx))
        t_both = med(lambda: zonal_stats(img, lab_n, n_labels=n_n, ctx=ctx, moments=True))
        t_tex1 = med(lambda: texture_stats(img, lab_n, bands=[0], n_labels=n_n, ctx=ctx))
        gb = float(H) * W * (4 * C + 4) / 1e9
        mom = max(t_both - t_stats, 1e-9)
        return {"segments": int(n_n), "statistics_ms": round(t_stats * 1e3, 3), "statistics_GBps": round(gb / t_stats, 1),
                "moments_ms": round(mom * 1e3, 3), "moments_GBps": round(gb / mom, 1), "moments_frac_of_hbm_peak": round(gb / mom / HBM_PEAK_GBS, 3),
                "texture_one_band_ms": round(t_tex1 * 1e3, 3), "texture_one_band_GBps": round(float(H) * W * 8 / 1e9 / t_tex1, 1),
                "note": "moments = zonal_stats(moments=True) minus zonal_stats(): the skewness / kurtosis pass alone; texture reads one band and the labels (8 B per pixel)"}

    c025_leg = compactness_leg(0.25) if abs(args.compactness - 0.25) > 1e-9 else None
    nr_leg = next_rows_leg()
    b9_leg = bands_leg(9)
    qs_leg = quickshift_leg()
    c4_whole = c4_leg()
    ms_per_step = dt / args.steps * 1e3
    total_px = float(H) * W * world
    value = total_px / (dt / args.steps) / 1e6

    if rank == 0:
        # algorithmic bytes of a colour sweep: 4C (features read) per pixel, + 4 (label written) on the sweeps that store their
        # labels -- only the last sweep of a batch does, the other sweeps' labels are dead stores the kernel does not make
        # (DESIGN.md 3.2).  SURVEY 8d's figure, 4C + 4 on every sweep, is kept beside it for comparison with round 1.
        bytes_per_px = 4 * C + 4
        avg_launch_ms = assign_ms / max(1.0, sweeps)
        alg_bytes = assign_px * 4 * C + assign_store_px * 4
        achieved = alg_bytes / (assign_ms * 1e-3) / 1e9 if assign_ms > 0 else 0.0
        achieved_survey = (assign_px * bytes_per_px) / (assign_ms * 1e-3) / 1e9 if assign_ms > 0 else 0.0
        out = {
            "metric": "Mpixel/s (SLIC+zonal feats) on 16384²×8-band; achieved HBM GB/s fraction",
            "value": round(value, 2), "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "step_wall_ms": [round((b - a) * 1e3, 2) for a, b in zip([t0] + step_wall[:-1], step_wall)],
            "scaling": "strong" if args.config == "c4" else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" + (" (NumPy RandomState on the host)" if args.host_rng else " (SURVEY 8d formula, noise drawn on the device)"),
            "config": {"workload": workload, "tile_size": args.tile, "buffer": args.buffer, "crown_radius": 5,
                       "pixel_size_m": 0.5, "compactness": args.compactness, "max_num_iter": 10, "mask": "all-ones",
                       "segments": int(n_seg), "parallelism": f"slab{world}" if world > 1 else "1gpu",
                       "backend": (backend if world > 1 else None), "devices": (min(world, n_dev) if world > 1 else 1)},
            # bound: the roofline the fraction is taken against (BASELINE's metric: achieved HBM GB/s fraction).  limited_by: what the
            # round-4 counters say stands in the way -- SQ_ACTIVE_INST_VALU 12.1 k cycles per wave x 6 waves per SIMD against
            # SQ_WAVE_CYCLES 73.6 k: the vector ALUs are 98 % busy in steady state (profiles/r04_notes.md, r04_pmc_sweep_kernels.txt);
            # with no feature / mask traffic at all the launch is 8 % shorter (OBIA_ABL_NOLOAD).  Dated figures, not live ones.
            "roofline": {"bound": "hbm", "limited_by": "valu", "valu_busy_steady_state": 0.91,
                         "limited_by_source": "profiles/r04_pmc_sweep_kernels.txt (PMC passes at 8192^2 at the round's HEAD: 10.97 k vector-ALU cycles per wave x 6 waves per SIMD of 72.2 k cycles of wave life; 0.98 before the round's last two cuts of vector work, profiles/r04_notes.md sections 2, 11, 12)",
                         "kernel": f"slic_assign_kernel<{(C + 3) // 4 * 4},true,false,false,false,{C}>",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         # HBM bytes per launch from the PMC counters (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes,
                         # gfx950 correction; tools/traffic_json.py).  Dated, not live: only reported when the kernel sources
                         # are byte-identical to the ones the counters were collected on and the workload is the same.
                         **traffic_fields(args, C, world),
                         "bytes_per_pixel": round(alg_bytes / max(1.0, assign_px), 2), "launches": int(sweeps), "avg_launch_ms": round(avg_launch_ms, 4),
                         "bytes_note": "4C per pixel read + 4 per pixel written by the sweeps that store labels (1 in 10)",
                         "frac_at_survey_36B_per_px": round(achieved_survey / HBM_PEAK_GBS, 4),
                         "pixels_per_launch_avg": round(assign_px / max(1.0, sweeps), 1),
                         "sweep_sum_ms_per_step": round(assign_ms / args.steps, 3), "sweep_busy_ms_per_step": round(assign_busy_ms / args.steps, 3)},
            "stage_ms_per_step": dict({k: round(v / max(1, parts_steps), 3) for k, v in parts.items()},
                                      source="one step after the timed region, every kernel class bracketed by events"),
        }
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(C, args.tile, args.buffer, 5, 0.5, args.compactness)
        else:
            out["cpu_baseline"] = None
        if phases_all is not None:
            out["sharded_phases"] = phases_all
        out["with_exit_on_fixed_point"] = fp_leg
        out["compactness_0.25"] = c025_leg
        out["quickshift"] = qs_leg
        out["c4_whole_on_one_gpu"] = c4_whole
        out["bands_9"] = b9_leg
        out["next_rows"] = nr_leg
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
