#!/usr/bin/env python3
"""Benchmark of the JPEG XL decode hot path on MI355X (contract: README of the build driver).

Workloads (--workload):
  4k        (default; BASELINE.json configs[1]) one *step* = one pass of the hot path over one batch: B 3840x2160 RGBA8 lossy VarDCT
            (distance 1.0) images - eight DISTINCT synthetic images, tests/golden/bench_index.json, repeated - decoded from
            HBM-resident .jxl bytes to HBM-resident RGBA8 through the C-ABI batch entry point jxlhip_decode_batch (host header
            parsing included in the timed region).  With --gpus N every rank decodes its own batch: weak scaling, independent images
            shard with no data-path collective (DESIGN.md "Multi-GPU").
  16k-bands (configs[2]) one step = ONE 16384x16384 RGBA8 lossy frame decoded by all ranks together: every rank decodes its band of
            256x256-group rows and the bands are gathered with one RCCL all_gather over xGMI.  Strong scaling.
  4k-lossless (configs[4]) one step = one 3840x2160 Modular lossless frame (Squeeze + weighted predictor: the stream kind the config
            names; two more kinds and the product encoder's own stream are timed beside it) decoded from HBM-resident bytes, checked
            bit-exact against the source picture.
  4k-encode (configs[3]) one step = one SaveImage of the 3840x2160 picture (lossy VarDCT, distance 1.0, effort 7) through the C-ABI.
  512-abi   (configs[0]) one step = one LoadImage of a 512x512 RGBA8 lossy file through the C-ABI (host buffers, callbacks): the call
            the reference's plugin makes, on the shape where a GPU has the least to offer.
  With --gpus N these three run as N independent replicas ("replicas only": one frame does not shard).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself (torch.distributed.run, one
process per GPU, before this process touches a GPU) and exits with their code.

value = megapixels decoded per second, whole job (all ranks).  Extra objects on the JSON line:
  roofline      dominant kernel = the stage with the most time per step; bound = HBM; achieved = algorithmic bytes of ONE launch
                (frames in that launch * (jxl bytes + W*H*4)) / the average HIP-event duration of one launch
  cpu_baseline  the CPU oracle (kind "port": libjxl is looked for with find_library and timed if present) built -O3 -march=native
                on this box and run on its host cores, rank 0 at N = 1 only
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def bench_files():
    """The eight distinct 4K inputs, in the order of their seeds."""
    index = json.load(open(os.path.join(GOLD, "bench_index.json")))
    names = sorted(index, key=lambda n: index[n]["seed"])
    return [open(os.path.join(GOLD, n + ".jxl"), "rb").read() for n in names]


# ------------------------------------------------------------------------------------------------ CPU baseline (rank 0, N = 1)
def _libjxl_probe(data, width, height, threads):
    """if a libjxl is discoverable at run time, time it on the same file; otherwise say that it is not there."""
    import ctypes as C
    import ctypes.util
    path = ctypes.util.find_library("jxl")
    if not path:
        return "not found by ctypes.util.find_library('jxl') on this box (offline image)"
    try:
        J = C.CDLL(path)
        tpath = ctypes.util.find_library("jxl_threads")
        T = C.CDLL(tpath) if tpath else None

        class PixelFormat(C.Structure):
            _fields_ = [("num_channels", C.c_uint32), ("data_type", C.c_int), ("endianness", C.c_int), ("align", C.c_size_t)]
        J.JxlDecoderCreate.restype = C.c_void_p
        J.JxlDecoderCreate.argtypes = [C.c_void_p]
        for fn in ("JxlDecoderSubscribeEvents", "JxlDecoderSetInput", "JxlDecoderProcessInput", "JxlDecoderSetImageOutBuffer",
                   "JxlDecoderSetParallelRunner"):
            getattr(J, fn).restype = C.c_int
        J.JxlDecoderSubscribeEvents.argtypes = [C.c_void_p, C.c_int]
        J.JxlDecoderSetInput.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        J.JxlDecoderProcessInput.argtypes = [C.c_void_p]
        J.JxlDecoderSetImageOutBuffer.argtypes = [C.c_void_p, C.POINTER(PixelFormat), C.c_void_p, C.c_size_t]
        J.JxlDecoderCloseInput.argtypes = [C.c_void_p]
        J.JxlDecoderDestroy.argtypes = [C.c_void_p]
        runner = None
        if T is not None:
            T.JxlThreadParallelRunnerCreate.restype = C.c_void_p
            T.JxlThreadParallelRunnerCreate.argtypes = [C.c_void_p, C.c_size_t]
            runner = T.JxlThreadParallelRunnerCreate(None, threads)
        fmt = PixelFormat(4, 2, 0, 0)   # RGBA, JXL_TYPE_UINT8, native endianness, tight rows (DecoderContext.cpp:22)
        out = (C.c_uint8 * (width * height * 4))()
        best = None
        for _ in range(3):
            dec = J.JxlDecoderCreate(None)
            if runner:
                J.JxlDecoderSetParallelRunner.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
                J.JxlDecoderSetParallelRunner(dec, C.cast(T.JxlThreadParallelRunner, C.c_void_p), runner)
            J.JxlDecoderSubscribeEvents(dec, 0x1000)   # JXL_DEC_FULL_IMAGE
            t0 = time.perf_counter()
            J.JxlDecoderSetInput(dec, data, len(data))
            J.JxlDecoderCloseInput(dec)
            while True:
                st = J.JxlDecoderProcessInput(dec)
                if st == 5:     # JXL_DEC_NEED_IMAGE_OUT_BUFFER
                    J.JxlDecoderSetImageOutBuffer(dec, C.byref(fmt), out, len(out))
                elif st in (0, 0x1000):
                    break
                elif st in (1, 2):
                    raise RuntimeError("libjxl status %d" % st)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            J.JxlDecoderDestroy(dec)
        return {"path": path, "mp_per_s": round(width * height / best / 1e6, 2), "threads": threads}
    except Exception as e:   # a library that is there but unusable is reported, not hidden
        return "found %s but could not time it: %s" % (path, e)


def cpu_baseline(files, width, height, seconds_budget=14.0):
    """Times the CPU oracle on the same files (bounded sample).  Only this leg touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    build = "-O3 -march=native, built on this box"
    try:
        O.use(O.build_native())
    except Exception as e:   # no compiler on the box: the portable build that travelled with the repo
        build = "portable -O2 build (native build failed: %s)" % str(e)[:80]
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    data = files[0]
    O.decode(data, num_threads=min(usable, 16))   # warm-up (page-in, table init)
    # the oracle threads over the 135 pass groups of a frame: find the thread count that is fastest on this box, then sample with it
    cands = sorted({t for t in (16, 32, 64, 128, usable) if t <= usable} or {usable})
    probe = {}
    for t in cands:
        t0 = time.perf_counter()
        O.decode(data, num_threads=t)
        probe[t] = time.perf_counter() - t0
    threads = min(probe, key=probe.get)
    times = []
    t_end = time.perf_counter() + seconds_budget
    k = 0
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 16):
        t0 = time.perf_counter()
        O.decode(files[k % len(files)], num_threads=threads)
        times.append(time.perf_counter() - t0)
        k += 1
    med = sorted(times)[len(times) // 2]
    return {"value": round(width * height / med / 1e6, 3), "unit": "MP/s", "cores": threads, "kind": "port",
            "sample": "%d decodes over the %d distinct 3840x2160 RGBA8 d=1.0 files with the CPU oracle (%s; %d threads over groups, the fastest of %s), median"
                      % (len(times), len(files), build, threads, cands),
            "host_cores_usable": usable, "host_cores_total": os.cpu_count(), "best_mp_per_s": round(width * height / min(times) / 1e6, 3),
            "libjxl": _libjxl_probe(data, width, height, threads)}


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(args):
    """--gpus N > 1 without a torchrun environment: start the N ranks as children.  Nothing here touches a GPU."""
    import torch
    rehearsal = os.environ.get("JXLHIP_BENCH_REHEARSAL") == "1"
    visible = torch.cuda.device_count()   # counting devices does not initialise the GPU on this image
    if visible < args.gpus and not rehearsal:
        print("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, visible), file=sys.stderr)
        return 2
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def stage_kernels(B, chunk):
    nchunks = (B + chunk - 1) // chunk
    return {"lf_ans": ("lf_ans_kernel", 1, B), "hf_decode": ("hf_decode_kernel", 1, B), "alpha_ans": ("alpha_ans_kernel", 1, B),
            "alpha_finish": ("alpha_finish_kernel", 1, B), "reconstruct": ("recon_tile_kernel", nchunks, min(B, chunk)),
            "filters+output": ("filter_stream_kernel", nchunks, min(B, chunk))}


def roofline_object(stage_ms, kernels, alg_bytes_per_image):
    """The dominant kernel of the run (most time per step) against the HBM roof."""
    dom = max(kernels, key=lambda k: stage_ms.get(k, 0.0))
    kname, launches, imgs_per_launch = kernels[dom]
    dom_ms = stage_ms.get(dom, 0.0) / launches          # average duration of ONE launch of that kernel (HIP events on its stream)
    alg_bytes = imgs_per_launch * alg_bytes_per_image
    # HBM traffic of that kernel from the newest committed PMC passes (profiles/r*_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE and
    # --pmc WRITE_SIZE in separate runs, gfx950 correction 2*FETCH + WRITE per the micro-architecture guide), scaled to one launch
    traffic = None
    try:
        pmc = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]))
        key = [k for k in pmc["kernels"] if k.startswith(kname)][0]
        traffic = int(pmc["kernels"][key]["hbm_bytes_per_image_corrected"] * imgs_per_launch)
    except Exception:
        pass
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    return {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "algorithmic_bytes_per_launch": int(alg_bytes),
            "launch_ms": round(dom_ms, 4), "launches_per_step": launches, "images_per_launch": imgs_per_launch}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["4k", "16k-bands", "4k-lossless", "4k-encode", "512-abi"], default=os.environ.get("JXLHIP_BENCH_WORKLOAD", "4k"))
    ap.add_argument("--batch", type=int, default=int(os.environ.get("JXLHIP_BENCH_BATCH", "384")))
    ap.add_argument("--frame-size", type=int, default=16384, help="side of the 16k-bands frame (smaller for rehearsals)")
    ap.add_argument("--lane-stride", type=int, default=0)
    ap.add_argument("--distinct", type=int, default=8, help="how many of the eight distinct 4K images the batch cycles through (experiments)")
    ap.add_argument("--distance", type=float, default=1.0, help="4k workload at another distance (2.0: two EPF iterations, 4.5: three): the eight "
                    "pictures are re-encoded by the product's own encoder before the timed region; a side measurement, not the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-steps", action="store_true", help="synchronise after every step (no cross-batch overlap)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {"4k": 20, "16k-bands": 10, "4k-lossless": 5, "4k-encode": 10, "512-abi": 30}[args.workload]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import torch
    import torch.distributed as dist
    from pdn_jpegxl_amd import api

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
    # Rehearsal hook for a one-GPU box (the real multi-GPU run uses one device per rank and RCCL): JXLHIP_BENCH_REHEARSAL=1 puts
    # every rank on device 0 and runs the collectives over gloo (RCCL refuses two ranks on one device).
    rehearsal = os.environ.get("JXLHIP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)   # "nccl" is RCCL on ROCm
    coll_dev = "cpu" if rehearsal else "cuda"

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # what every rank saw: the driver's scaling record can check that N ranks really ran on N devices over RCCL
    me = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(), "device_name": torch.cuda.get_device_name(),
          "world_size_seen": dist.get_world_size() if world > 1 else 1, "backend": (dist.get_backend() if world > 1 else "none"),
          "visible_devices": torch.cuda.device_count()}
    ranks = [me]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    dec = api.Decoder(local_rank)
    dec.set_option("lane_stride", args.lane_stride)
    if args.workload == "16k-bands":
        line = run_bands(args, dec, rank, world, rehearsal, coll_dev, fence, max_over_ranks)
    elif args.workload == "4k":
        line = run_4k(args, dec, rank, world, coll_dev, fence, max_over_ranks)
    else:
        line = run_side(args, dec, rank, world, fence, max_over_ranks)
    if rank == 0 and line is not None:
        line["ranks"] = ranks
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_4k(args, dec, rank, world, coll_dev, fence, max_over_ranks):
    import torch
    import torch.distributed as dist
    from pdn_jpegxl_amd import api
    distinct = bench_files()[: max(1, args.distinct)]
    if args.distance != 1.0:
        import numpy as np
        from pdn_jpegxl_amd.synth import synth
        index = json.load(open(os.path.join(GOLD, "bench_index.json")))
        seeds = sorted(v["seed"] for v in index.values())[: len(distinct)]
        distinct = [api.save_image(np.ascontiguousarray(synth(3840, 2160, sd)[..., [2, 1, 0, 3]]), distance=args.distance, effort=7) for sd in seeds]
    info = api.peek(distinct[0])
    W, H, C = info.width, info.height, info.num_channels
    # Batch per GPU: as many frames as the entropy stages can keep in flight (their time per batch is nearly constant), bounded by
    # HBM: three pipeline slots of entropy-stage state per frame (worst-case entry lists, alpha, LF), two output sets, and the shared
    # pixel-stage planes.
    free_b = torch.cuda.mem_get_info()[0]
    fit = int((free_b * 0.88 - 40e9) / (3 * 165e6 + 2 * W * H * C))
    B = max(1, min(args.batch, fit))
    if world > 1:
        t = torch.tensor([B], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        B = int(t.item())
    # inputs resident in HBM before the timed region (padded: the bit readers fetch whole 32-bit words)
    srcs = []
    for d in distinct:
        s = torch.zeros(len(d) + 64, dtype=torch.uint8, device="cuda")
        s[: len(d)] = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
        srcs.append(s)
    files = [distinct[(i + rank) % len(distinct)] for i in range(B)]
    dev_in = [srcs[(i + rank) % len(distinct)].data_ptr() for i in range(B)]
    outs = [torch.empty(W * H * C, dtype=torch.uint8, device="cuda") for _ in range(B)]
    dev_out = [o.data_ptr() for o in outs]
    # two output sets: consecutive asynchronous batches must not write the same buffers
    outs2 = [torch.empty(W * H * C, dtype=torch.uint8, device="cuda") for _ in range(B)]
    dev_out2 = [o.data_ptr() for o in outs2]
    counter = [0]

    def step():
        # asynchronous submit: the LF stage of this batch overlaps the HF/pixel stages of the previous one
        # (three workspace slots inside the decoder); everything is complete at dec.finish() below.
        counter[0] += 1
        st = dec.decode_batch(files, dev_out if counter[0] & 1 else dev_out2, dev_in, synchronize=args.sync_steps)
        assert all(s == 0 for s in st), st

    for _ in range(3):      # setup: populate the decoder's three workspace slots (allocation is not decode work)
        step()
    dec.finish()
    for _ in range(args.warmup):
        step()
    dec.finish()
    dec.stage_totals(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dec.finish()          # every one of the K batches is complete (statuses checked) before the clock stops
    fence()
    elapsed = time.perf_counter() - t0
    stage_sum, nb = dec.stage_totals(reset=True)
    assert nb == args.steps, (nb, args.steps)
    elapsed = max_over_ranks(elapsed)

    # host cost of submitting one batch into an idle pipeline (header parsing of B files, table blob, enqueue), outside the timed region:
    # it has to stay below the GPU time per step for the step to be GPU-bound
    host_ms = []
    for _ in range(2):
        th = time.perf_counter()
        dec.decode_batch(files, dev_out, dev_in, synchronize=False)
        host_ms.append((time.perf_counter() - th) * 1e3)
        dec.finish()
    # single-image latency (B = 1), outside the timed region
    lat = []
    for _ in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        dec.decode_batch(files[:1], dev_out[:1], dev_in[:1], synchronize=True)
        lat.append((time.perf_counter() - t1) * 1e3)
    lat_stages = dec.stage_times()
    # ... and what the reference's plugin calls: LoadImage through the C-ABI, host buffers in and out, callbacks (one image per
    # synchronous call, JpegXLLoad.cs:30-69).  The first call of a thread pays for the decoder's workspaces: reported as "cold".
    t1 = time.perf_counter()
    api.load_image(distinct[0])
    abi_cold = (time.perf_counter() - t1) * 1e3
    abi = []
    for _ in range(4):
        t1 = time.perf_counter()
        api.load_image(distinct[0])
        abi.append((time.perf_counter() - t1) * 1e3)
    if rank != 0:
        return None
    mp = W * H / 1e6
    stage_ms = {k: v / args.steps for k, v in stage_sum.items()}
    chunk = dec.set_option("query_pixel_chunk", 0) or B
    mean_jxl = sum(len(f) for f in files) / len(files)
    line = {
        "metric": "megapixels/sec decode (4K lossy VarDCT)",
        "value": round(world * B * args.steps * mp / elapsed, 2),
        "unit": "MP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "3840x2160 RGBA8 lossy VarDCT (distance=%.1f%s) decode, HBM-resident .jxl -> HBM RGBA8"
                               % (args.distance, "" if args.distance == 1.0 else ", files written by the product encoder"),
                   "batch_per_gpu": B, "distinct_images": len(distinct), "jxl_bytes_mean": int(mean_jxl), "groups_per_image": info.num_groups,
                   "lane_stride": args.lane_stride or "auto", "async_steps": not args.sync_steps,
                   "parallelism": "images sharded across ranks, no data-path collective"},
        "roofline": roofline_object(stage_ms, stage_kernels(B, chunk), int(mean_jxl) + W * H * C),
        "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
        "stage_ms_note": "HIP-event time per stage on its own stream; three streams overlap, so the stages add up to more than ms_per_step",
        "host_submit_ms_per_batch": round(min(host_ms), 3),
        "single_image": {"latency_ms": round(min(lat), 3), "mp_per_s": round(mp / (min(lat) * 1e-3), 2),
                         "stage_ms": {k: round(v, 4) for k, v in lat_stages.items()},
                         "through_abi_ms": {"warm": round(min(abi), 3), "cold_first_call_of_the_thread": round(abi_cold, 3),
                                            "mp_per_s_warm": round(mp / (min(abi) * 1e-3), 2),
                                            "note": "LoadImage with host buffers and callbacks (PCIe both ways included); cold = first call of a thread "
                                                    "in a process whose HIP context exists: workspace allocation, not decode work"}},
    }
    if line["roofline"]["traffic"] is not None:
        line["roofline"]["traffic_note"] = "PMC bytes (2*FETCH_SIZE + WRITE_SIZE) from the newest committed profiles/r*_pmc_traffic.json, profiled at batch 16 and scaled to this launch's images; not measured in this run"
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(distinct, W, H)
    return line


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    try:
        O.use(O.build_native())
        return O, "-O3 -march=native, built on this box"
    except Exception as e:
        return O, "portable -O2 build (native build failed: %s)" % str(e)[:80]


def _cpu_side(fn, unit_mp, what, seconds_budget=12.0):
    """cpu_baseline of a side workload: fn(threads) runs the oracle once; thread count probed, bounded sample, median."""
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cands = sorted({t for t in (1, 8, 16, 32, 64) if t <= usable})
    probe = {}
    for t in cands:
        t0 = time.perf_counter()
        fn(t)
        probe[t] = time.perf_counter() - t0
    threads = min(probe, key=probe.get)
    times = []
    t_end = time.perf_counter() + seconds_budget
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 12):
        t0 = time.perf_counter()
        fn(threads)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(unit_mp / med, 3), "unit": "MP/s", "cores": threads, "kind": "port",
            "sample": "%d runs of %s, %d threads (the fastest of %s), median" % (len(times), what, threads, cands),
            "host_cores_usable": usable, "ms": round(med * 1e3, 2)}


def _roof(stage_ms, alg_bytes, kernel_of):
    dom = max(stage_ms, key=stage_ms.get)
    ms = stage_ms[dom]
    ach = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {"bound": "hbm", "kernel": kernel_of.get(dom, dom), "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6),
            "traffic": None, "algorithmic_bytes_per_launch": int(alg_bytes), "launch_ms": round(ms, 4), "stage": dom,
            "note": "dominant stage of one call (HIP events on the stage's stream); one call = one launch of each of its kernels"}


def run_side(args, dec, rank, world, fence, max_over_ranks):
    """configs[0], [3], [4]: one frame per step, N replicas when --gpus N."""
    import numpy as np
    import torch
    from pdn_jpegxl_amd import api
    from pdn_jpegxl_amd.synth import synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench_inputs
    want_cpu = world == 1 and not args.no_cpu_baseline
    extra = {}
    if args.workload == "4k-lossless":
        streams, rgb = bench_inputs.lossless_4k_streams()
        streams = dict(streams)
        streams["product-encoder (YCoCg-R + gradient, row-static)"] = api.save_image(np.ascontiguousarray(np.dstack([rgb[..., 2], rgb[..., 1], rgb[..., 0], np.full(rgb.shape[:2], 255, np.uint8)])), lossless=True)
        main_key = "squeeze+weighted"
        H, W = rgb.shape[:2]
        results = {}
        for key, data in streams.items():
            info = api.peek(data)
            out = torch.empty(info.width * info.height * info.num_channels, dtype=torch.uint8, device="cuda")
            src = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
            src[: len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
            run = lambda: dec.decode_batch([data], [out.data_ptr()], [src.data_ptr()], synchronize=True)
            run()
            got = out.cpu().numpy().reshape(info.height, info.width, info.num_channels)
            assert np.array_equal(got[..., :3], rgb), "lossless decode of %s is not bit-exact" % key
            k = args.steps if key == main_key else 2
            for _ in range(args.warmup if key == main_key else 0):
                run()
            fence()
            t0 = time.perf_counter()
            for _ in range(k):
                run()
            fence()
            ms = max_over_ranks((time.perf_counter() - t0) / k * 1e3)
            stages = dec.stage_times()
            api.load_image(data)
            t1 = time.perf_counter()
            api.load_image(data)
            results[key] = {"ms": round(ms, 2), "mp_per_s": round(W * H / 1e3 / ms, 2), "jxl_bytes": len(data), "stage_ms": {n: round(v, 3) for n, v in stages.items()},
                            "through_abi_ms": round((time.perf_counter() - t1) * 1e3, 2), "bit_exact_vs_source": True}
        ms = results[main_key]["ms"]
        metric, workload = "megapixels/sec decode (4K Modular lossless)", "3840x2160 RGB8 Modular lossless (Squeeze + weighted predictor + MA tree), HBM-resident .jxl -> HBM RGB8, bit-exact vs source"
        roof = _roof({k: v for k, v in results[main_key]["stage_ms"].items()}, len(streams[main_key]) + W * H * 3, {"modular": "modular_ans_kernel"})
        extra["variants"] = results
        mp = W * H / 1e6
        if want_cpu:
            O, build = _oracle()
            extra["cpu_baseline"] = _cpu_side(lambda t: O.decode(streams[main_key], num_threads=t), mp, "the CPU oracle decoding the same stream (%s)" % build)
    elif args.workload == "4k-encode":
        img = synth(3840, 2160, 2)
        bgra = np.ascontiguousarray(img[..., [2, 1, 0, 3]])
        H, W = img.shape[:2]
        run = lambda: api.save_image(bgra, distance=1.0, effort=7)
        data = run()
        for _ in range(args.warmup):
            run()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            data = run()
        fence()
        ms = max_over_ranks((time.perf_counter() - t0) / args.steps * 1e3)
        stages = api.last_save_stage_times()
        back = api.load_image(data).pixels
        err = np.abs(back.astype(np.int32) - img.astype(np.int32))
        metric, workload = "megapixels/sec encode (4K lossy VarDCT)", "3840x2160 BGRA8 -> lossy VarDCT .jxl (distance 1.0, effort 7) through SaveImage (host buffer in, Write callbacks out: PCIe included, the ABI has no device-resident form)"
        roof = _roof(stages, W * H * 4 + len(data), {})
        extra.update({"jxl_bytes": len(data), "bits_per_pixel": round(len(data) * 8 / (W * H), 3), "stage_ms": {n: round(v, 3) for n, v in stages.items()},
                      "roundtrip_mean_abs_error_u8": round(float(err[..., :3].mean()), 3), "alpha_exact": bool((back[..., 3] == img[..., 3]).all())})
        mp = W * H / 1e6
        if want_cpu:
            O, build = _oracle()
            extra["cpu_baseline"] = _cpu_side(lambda t: O.encode(img, distance=1.0, num_threads=t), mp, "the CPU oracle encoding the same picture, same settings (%s)" % build)
    else:   # 512-abi
        data, img = bench_inputs.lossy_512()
        H, W = img.shape[:2]
        run = lambda: api.load_image(data)
        t1 = time.perf_counter()
        got = run()
        cold = (time.perf_counter() - t1) * 1e3
        for _ in range(args.warmup):
            run()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        fence()
        ms = max_over_ranks((time.perf_counter() - t0) / args.steps * 1e3)
        stages = api.last_load_stage_times()
        metric, workload = "megapixels/sec decode (512x512 lossy, LoadImage)", "512x512 RGBA8 lossy VarDCT (distance 1.0) through LoadImage: host buffers, callbacks, PCIe both ways"
        roof = _roof(stages, len(data) + W * H * 4, {"hf_decode": "hf_decode_kernel", "lf_ans": "lf_ans_kernel", "alpha_ans": "alpha_ans_kernel"})
        extra.update({"jxl_bytes": len(data), "stage_ms": {n: round(v, 3) for n, v in stages.items()}, "cold_first_call_ms": round(cold, 2)})
        mp = W * H / 1e6
        if want_cpu:
            O, build = _oracle()
            ref = O.decode(data).pixels
            d = np.abs(got.pixels.astype(np.int32) - ref.astype(np.int32))
            extra["max_abs_diff_vs_oracle_u8"] = int(d.max())
            extra["cpu_baseline"] = _cpu_side(lambda t: O.decode(data, num_threads=t), mp, "the CPU oracle decoding the same file (%s)" % build, seconds_budget=6.0)
    if rank != 0:
        return None
    line = {"metric": metric, "value": round(world * mp / (ms * 1e-3), 2), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32" if args.workload == "4k-lossless" else "f32", "data": "synthetic",
            "config": {"workload": workload, "parallelism": "replicas only (one frame does not shard)" if world > 1 else "one GPU"},
            "roofline": roof}
    line.update(extra)
    return line


def make_big_frame(n):
    """An n x n RGBA8 lossy frame (distance 1.0) written by the product's own encoder from tiled synthetic content."""
    import numpy as np
    from pdn_jpegxl_amd import api
    from pdn_jpegxl_amd.synth import synth
    base = synth(3840, 2160, 3)
    img = np.tile(base, (n // 2160 + 1, n // 3840 + 1, 1))[:n, :n]
    bgra = np.ascontiguousarray(img[..., [2, 1, 0, 3]])
    return api.save_image(bgra, distance=1.0)


def run_bands(args, dec, rank, world, rehearsal, coll_dev, fence, max_over_ranks):
    import torch
    import torch.distributed as dist
    from pdn_jpegxl_amd import api
    from pdn_jpegxl_amd.distributed import BandDecoder
    n = args.frame_size
    # rank 0 writes the frame, everybody gets the same bytes
    if world > 1:
        if rank == 0:
            data = make_big_frame(n)
            size = torch.tensor([len(data)], dtype=torch.int64, device=coll_dev)
        else:
            size = torch.zeros(1, dtype=torch.int64, device=coll_dev)
        dist.broadcast(size, 0)
        buf = torch.empty(int(size.item()), dtype=torch.uint8, device=coll_dev)
        if rank == 0:
            buf.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
        dist.broadcast(buf, 0)
        data = bytes(buf.cpu().numpy().tobytes())
    else:
        data = make_big_frame(n)
    info = api.peek(data)
    W, H, C = info.width, info.height, info.num_channels
    bd = BandDecoder(dec, data, rank, world, gather_device=coll_dev)   # the codestream goes to HBM once, before the timed region
    for _ in range(2 + args.warmup):
        bd.step()
    dec.stage_totals(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        bd.step()
    fence()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    stage_sum, nb = dec.stage_totals(reset=True)
    # where one band's time goes (rank 0, stage times of the last step), and the gather alone
    band_stages = dec.stage_times()
    tg = []
    for _ in range(3):
        fence()
        t1 = time.perf_counter()
        bd.gather()
        torch.cuda.synchronize()
        tg.append((time.perf_counter() - t1) * 1e3)
    if rank != 0:
        return None
    mp = W * H / 1e6
    stage_ms = {k: v / max(1, nb) for k, v in stage_sum.items()}
    rows = bd.rows
    frac = rows / H if H else 1.0
    kernels = {"lf_ans": ("lf_ans_kernel", 1, frac), "hf_decode": ("hf_decode_kernel", 1, frac), "alpha_ans": ("alpha_ans_kernel", 1, frac),
               "alpha_finish": ("alpha_finish_kernel", 1, frac), "reconstruct": ("recon_tile_kernel", 1, frac),
               "filters+output": ("filter_stream_kernel", 1, frac)}
    return {
        "metric": "megapixels/sec decode (16384x16384 lossy VarDCT frame, band-sharded)",
        "value": round(args.steps * mp / elapsed, 2),
        "unit": "MP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%dx%d RGBA8 lossy VarDCT (distance=1.0) frame, 256x256 groups sharded by group rows across ranks, RCCL all_gather of the bands"
                               % (W, H), "jxl_bytes": len(data), "groups": info.num_groups, "band_rows_rank0": rows,
                   "collective": "gloo (one-GPU rehearsal)" if rehearsal else ("all_gather over RCCL" if world > 1 else "none (one rank)"),
                   "parallelism": "bands%d" % world},
        "roofline": roofline_object(stage_ms, kernels, len(data) + W * H * C),
        "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
        "band_stage_ms_rank0": {k: round(v, 4) for k, v in band_stages.items()},
        "gather_ms": round(min(tg), 3),
    }


if __name__ == "__main__":
    main()
