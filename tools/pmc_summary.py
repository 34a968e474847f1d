#!/usr/bin/env python3
"""Folds two rocprofv3 counter-collection CSVs (one --pmc FETCH_SIZE pass, one --pmc WRITE_SIZE pass, same command) into
per-kernel HBM traffic per image.  Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
counter values are KB; FETCH_SIZE under-reports wide streaming reads by 2x, so hbm_bytes = (2*FETCH + WRITE) * 1024."""
import csv
import json
import sys


def fold(path):
    out = {}
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("jxlhip::", "")
        tot, cnt = out.get(name, (0.0, 0))
        out[name] = (tot + float(row["Counter_Value"]), cnt + 1)
    return out


def main():
    fetch, write, batch, dst = fold(sys.argv[1]), fold(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 1))
        w, nw = write.get(k, (0.0, 1))
        kernels[k] = {"launches": max(nf, nw), "fetch_size_kb_per_launch": f / max(nf, 1), "write_size_kb_per_launch": w / max(nw, 1),
                      "hbm_bytes_per_image_corrected": (2 * f / max(nf, 1) + w / max(nw, 1)) * 1024 / batch}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace), bench.py --batch %d --sync-steps, MI355X" % batch,
               "units": "counter values are KB; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per the micro-architecture guide (gfx950 correction); "
                        "narrow / scattered accesses (the entropy kernels) are uncalibrated",
               "batch": batch, "kernels": kernels}, open(dst, "w"), indent=1)


if __name__ == "__main__":
    main()
