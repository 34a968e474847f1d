"""hf_decode time of every bench fixture decoded alone (one section per wavefront): the kernel's time is its slowest section, so the
slowest fixture bounds the batch launch from below."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pdn_jpegxl_amd import api
idx = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_index.json")))
dec = api.Decoder(0)
for name in sorted(idx):
    data = open(os.path.join(ROOT, "tests", "golden", name + ".jxl"), "rb").read()
    info = api.peek(data)
    out = torch.empty(info.width * info.height * info.num_channels, dtype=torch.uint8, device="cuda")
    best = None
    for _ in range(3):
        dec.decode_batch([data], [out.data_ptr()])
        st = dec.stage_times()
        best = st if best is None or st["hf_decode"] < best["hf_decode"] else best
    sizes = api.section_sizes(data) if hasattr(api, "section_sizes") else None
    print(name, "hf %.1f ms  lf_ans %.1f  alpha_ans %.1f" % (best["hf_decode"], best["lf_ans"], best["alpha_ans"]), "largest pass-group section %s B" % (max(sizes[-info.num_groups:]) if sizes else "?"))
