"""One 4K lossless stream kind decoded a few times (device-resident): the workload of tools/pmc_lossless.sh.
usage: python tools/lossless_one.py {gradient-context-tree|weighted+prop15|squeeze+weighted} [repeats]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_inputs
from pdn_jpegxl_amd import api
kind = sys.argv[1] if len(sys.argv) > 1 else "gradient-context-tree"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
streams, rgb = bench_inputs.lossless_4k_streams()
data = streams[kind]
dec = api.Decoder(0)
info = api.peek(data)
out = torch.empty(info.width * info.height * info.num_channels, dtype=torch.uint8, device="cuda")
for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dec.decode_batch([data], [out.data_ptr()]); torch.cuda.synchronize(); t = time.perf_counter() - t0
print("%s: %.1f ms, %d sections, samples %d" % (kind, t * 1e3, info.num_groups, 3 * info.width * info.height), flush=True)
