import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from pdn_jpegxl_amd import api
data = open(os.path.join(ROOT, "tests", "golden", "synth_3840x2160_seed2_d1.jxl"), "rb").read()
info = api.peek(data)
dec = api.Decoder(0)
out = torch.empty(info.width * info.height * info.num_channels, dtype=torch.uint8, device="cuda")
for _ in range(2):
    dec.decode_batch([data], [out.data_ptr()]); torch.cuda.synchronize()
print(dec.stage_times())
