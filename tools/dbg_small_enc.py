import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth
bgra = np.ascontiguousarray(synth(200, 150, 3)[..., [2, 1, 0, 3]])
api.save_image(bgra, distance=1.0)
print("---- second call", file=sys.stderr, flush=True)
api.save_image(bgra, distance=1.0)
print("---- lossless", file=sys.stderr, flush=True)
api.save_image(bgra, lossless=True)
