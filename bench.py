#!/usr/bin/env python3
"""Benchmark of the JPEG XL decode hot path on MI355X (contract: README of the build driver).

One *step* = one pass of the hot path over one batch: B copies of the 3840x2160 RGBA8 lossy VarDCT (distance 1.0)
synthetic image (BASELINE.json configs[1]) decoded from HBM-resident .jxl bytes to HBM-resident RGBA8, through the
C-ABI batch entry point jxlhip_decode_batch (host header parsing included in the timed region).
value = megapixels decoded per second, whole job (all ranks).  With --gpus N each rank decodes its own batch
(weak scaling; independent images shard with no data-path collective — DESIGN.md "Multi-GPU").

Extra objects on the JSON line:
  roofline      dominant kernel = the one with the most time per batch among lf_ans / hf_decode / alpha_ans / alpha_finish /
                recon_tile / filter_gab_epf1, bound = HBM; achieved = algorithmic bytes of ONE launch (frames in that launch *
                (jxl bytes + W*H*4)) / the average HIP-event duration of one launch
  cpu_baseline  the CPU oracle (kind "port"; libjxl is not available offline) on this box's host cores, rank 0 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FIXTURE = os.path.join(ROOT, "tests", "golden", "synth_3840x2160_seed2_d1.jxl")
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def cpu_baseline(data, width, height, seconds_budget=12.0):
    """Times the CPU oracle on the same file (bounded sample).  Only this leg touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = os.cpu_count() or 1
    threads = min(cores, 32)
    O.decode(data, num_threads=threads)  # warm-up (page-in, table init)
    times = []
    t_end = time.perf_counter() + seconds_budget
    while len(times) < 2 or (time.perf_counter() < t_end and len(times) < 8):
        t0 = time.perf_counter()
        O.decode(data, num_threads=threads)
        times.append(time.perf_counter() - t0)
    best = min(times)
    return {"value": round(width * height / best / 1e6, 3), "unit": "MP/s", "cores": threads, "kind": "port",
            "sample": "%d decodes of the same 3840x2160 RGBA8 d=1.0 file with the CPU oracle (%d threads over groups), best of %d"
                      % (len(times), threads, len(times)), "host_cores": cores, "libjxl": "unavailable (offline)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("JXLHIP_BENCH_BATCH", "384")))
    ap.add_argument("--lane-stride", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-steps", action="store_true", help="synchronise after every step (no cross-batch overlap)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pdn_jpegxl_amd import api

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
    # Rehearsal hook for a one-GPU box (the real multi-GPU run uses one device per rank and RCCL): JXLHIP_BENCH_REHEARSAL=1 puts
    # every rank on device 0 and runs the (tiny, non-data-path) collectives over gloo.
    rehearsal = os.environ.get("JXLHIP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)   # "nccl" is RCCL on ROCm
    coll_dev = "cpu" if rehearsal else "cuda"

    data = open(FIXTURE, "rb").read()
    info = api.peek(data)
    W, H, C = info.width, info.height, info.num_channels
    # Batch per GPU: as many frames as the entropy stages can keep in flight (their time per batch is nearly constant), bounded by
    # HBM: three pipeline slots of ~165 MB of entropy-stage state per frame, two output sets, and the shared pixel-stage planes.
    free_b = torch.cuda.mem_get_info()[0]
    fit = int((free_b * 0.88 - 26e9) / (3 * 165e6 + 2 * W * H * C))
    B = max(1, min(args.batch, fit))
    if world > 1:
        t = torch.tensor([B], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        B = int(t.item())
    dec = api.Decoder(local_rank)
    dec.set_option("lane_stride", args.lane_stride)
    # inputs resident in HBM before the timed region (padded: the bit readers fetch whole 32-bit words)
    src = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
    src[: len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    outs = [torch.empty(W * H * C, dtype=torch.uint8, device="cuda") for _ in range(B)]
    files = [data] * B
    dev_in = [src.data_ptr()] * B
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

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

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
        dec.decode_batch([data], dev_out[:1], dev_in[:1], synchronize=True)
        lat.append((time.perf_counter() - t1) * 1e3)
    lat_stages = dec.stage_times()

    if rank == 0:
        mp = W * H / 1e6
        stage_ms = {k: v / args.steps for k, v in stage_sum.items()}
        # stage -> (kernel, launches per batch, images per launch).  The pixel stages run in chunks of `chunk` frames that share one set
        # of float planes, so their kernels are launched ceil(B / chunk) times per batch; the entropy kernels cover the whole batch.
        chunk = dec.set_option("query_pixel_chunk", 0) or B
        nchunks = (B + chunk - 1) // chunk
        kernels = {"lf_ans": ("lf_ans_kernel", 1, B), "hf_decode": ("hf_decode_kernel", 1, B), "alpha_ans": ("alpha_ans_kernel", 1, B),
                   "alpha_finish": ("alpha_finish_kernel", 1, B), "reconstruct": ("recon_tile_kernel", nchunks, min(B, chunk)),
                   "filters+output": ("filter_gab_epf1_kernel", nchunks, min(B, chunk))}
        # the dominant kernel of this run: most time per batch
        dom = max(kernels, key=lambda k: stage_ms.get(k, 0.0))
        kname, launches, imgs_per_launch = kernels[dom]
        dom_ms = stage_ms.get(dom, 0.0) / launches          # average duration of ONE launch of that kernel (HIP events on its stream)
        alg_bytes = imgs_per_launch * (len(data) + W * H * C)
        # HBM traffic of that kernel from the newest committed PMC passes (profiles/r*_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE and
        # --pmc WRITE_SIZE in separate runs, gfx950 correction 2*FETCH + WRITE per the micro-architecture guide), scaled to one launch
        traffic = None
        try:
            import glob
            pmc = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]))
            key = [k for k in pmc["kernels"] if k.startswith(kname)][0]
            traffic = int(pmc["kernels"][key]["hbm_bytes_per_image_corrected"] * imgs_per_launch)
        except Exception:
            pass
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
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
            "config": {"workload": "3840x2160 RGBA8 lossy VarDCT (distance=1.0) decode, HBM-resident .jxl -> HBM RGBA8",
                       "batch_per_gpu": B, "jxl_bytes": len(data), "groups_per_image": info.num_groups,
                       "lane_stride": args.lane_stride or "auto", "async_steps": not args.sync_steps, "parallelism": "images sharded across ranks, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "launch_ms": round(dom_ms, 4), "launches_per_step": launches,
                         "images_per_launch": imgs_per_launch},
            "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
            "host_submit_ms_per_batch": round(min(host_ms), 3),
            "single_image": {"latency_ms": round(min(lat), 3), "mp_per_s": round(mp / (min(lat) * 1e-3), 2),
                             "stage_ms": {k: round(v, 4) for k, v in lat_stages.items()}},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(data, W, H)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
