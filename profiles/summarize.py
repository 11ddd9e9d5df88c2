"""Turns the rocprofv3 CSVs of one profiling round (gpurun_out/<dir>) into the committed summaries:
  profiles/<tag>_kernel_stats.csv      copy of `rocprofv3 --kernel-trace --stats` for `python bench.py`
  profiles/<tag>_pmc.json              FETCH_SIZE / WRITE_SIZE per launch of the dominant kernels
Usage: python profiles/summarize.py gpurun_out/prof2 r01
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-reports a wide coalesced stream by 2x
(MI355X_MICROARCH.md, HBM section), so the read side is doubled; WRITE_SIZE is taken as reported (uncalibrated)."""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main(src, tag):
    here = os.path.dirname(os.path.abspath(__file__))
    shutil.copy(os.path.join(src, "stats_kernel_stats.csv"), os.path.join(here, tag + "_kernel_stats.csv"))
    if os.path.exists(os.path.join(src, "serial_kernel_stats.csv")):      # VNET_PARAM_GRAD_STREAM=0: no kernels overlap
        shutil.copy(os.path.join(src, "serial_kernel_stats.csv"), os.path.join(here, tag + "_kernel_stats_serial.csv"))
    pmc = {}
    for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        path = os.path.join(src, which + "_counter_collection.csv")
        if not os.path.exists(path):
            continue
        agg = defaultdict(lambda: [0, 0.0, 0.0])
        # several rows per dispatch (one per XCD/instance) -> sum per dispatch first
        per_dispatch = defaultdict(float)
        meta = {}
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            key = r["Dispatch_Id"]
            per_dispatch[key] += float(r["Counter_Value"])
            meta[key] = (short(r["Kernel_Name"]), r["Grid_Size"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for key, val in per_dispatch.items():
            name, grid, dur = meta[key]
            a = agg[(name, grid)]
            a[0] += 1
            a[1] += val
            a[2] += dur
        pmc[counter] = agg
    out = {"note": "per launch; KB from rocprofv3 --pmc (separate passes); fetch doubled per the gfx950 correction", "kernels": {}}
    keys = set()
    for agg in pmc.values():
        keys |= set(agg)
    for key in sorted(keys):
        name, grid = key
        if not (name.startswith("conv_kernel<5") or name.startswith("wgrad_kernel<5") or name.startswith("conv5_bf16") or name.startswith("wgrad5_bf16")):
            continue
        e = {"grid_threads": int(grid)}
        if "FETCH_SIZE" in pmc and key in pmc["FETCH_SIZE"]:
            n, v, d = pmc["FETCH_SIZE"][key]
            e["launches"] = n
            e["fetch_KB_reported"] = v / n
            e["hbm_read_bytes"] = 2.0 * 1024.0 * v / n
            e["avg_us_profiled"] = d / n / 1e3
        if "WRITE_SIZE" in pmc and key in pmc["WRITE_SIZE"]:
            n, v, d = pmc["WRITE_SIZE"][key]
            e["hbm_write_bytes"] = 1024.0 * v / n
        e["hbm_bytes_per_launch"] = e.get("hbm_read_bytes", 0.0) + e.get("hbm_write_bytes", 0.0)
        out["kernels"]["%s grid=%s" % (name, grid)] = e
    # the kernel family bench.py reports (decoder level 1 conv_1 at 128^3: forward 32->16, backward-data 16->32, filter
    # gradient): launches on 8192 bricks x 256 threads (conv) / the one-slab filter-gradient kernels
    fams = {"fp32": [k for k in out["kernels"] if (k.startswith("conv_kernel<5, 1, 4, 8, 8, 4, 4, 1, false, 5> grid=2097152")
                                                     or k.startswith("conv_kernel<5, 1, 4, 8, 8, 4, 4, 2, false, 5> grid=2097152")
                                                     or k.startswith("wgrad_kernel<5, 1, 4, 4, 16, 1, 16, 5>"))],
            # bf16: the 128^3 launches cannot be told apart by grid -- all five conv (4->16, 16->16, 32->16 forward; 16->16,
            # 16->32 backward-data) and three filter-gradient launches per step of the C5 network are averaged
            "bf16": [k for k in out["kernels"] if (k.startswith("conv5_bf16_kernel<4, 8, 16, 1, 8> grid=2097152")
                                                     or k.startswith("conv5_bf16_c16_kernel")
                                                     or k.startswith("wgrad5_bf16_kernel<4, 4, 16, 1, 16>"))]}
    out["families"] = {}
    for name, keys in fams.items():
        sel = [out["kernels"][k] for k in keys if out["kernels"][k].get("launches")]
        if sel:
            n = sum(e["launches"] for e in sel)
            out["families"][name] = {"kernels": keys, "launches": n,
                                     "hbm_bytes_per_launch": round(sum(e["launches"] * e["hbm_bytes_per_launch"] for e in sel) / n)}
    json.dump(out, open(os.path.join(here, tag + "_pmc.json"), "w"), indent=1)
    for k, e in out["kernels"].items():
        print("%-70s n=%3d read %8.1f MB write %8.1f MB  %8.1f us" % (k, e.get("launches", 0), e.get("hbm_read_bytes", 0) / 1e6,
                                                                      e.get("hbm_write_bytes", 0) / 1e6, e.get("avg_us_profiled", 0)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
