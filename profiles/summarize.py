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

# launches per step of the C5 network's 32-cout row-pair kernel without statistics / of the row-reuse filter gradient with TZ = 4
# (checked against the per-step launch counts the script prints)
FAM_BWD_PER_STEP = None
FAM_WGRAD_PER_STEP = None


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main(src, tag):
    here = os.path.dirname(os.path.abspath(__file__))
    shutil.copy(os.path.join(src, "stats_kernel_stats.csv"), os.path.join(here, tag + "_kernel_stats.csv"))
    if os.path.exists(os.path.join(src, "serial_kernel_stats.csv")):      # VNET_PARAM_GRAD_STREAM=0: no kernels overlap
        shutil.copy(os.path.join(src, "serial_kernel_stats.csv"), os.path.join(here, tag + "_kernel_stats_serial.csv"))
    pmc = {}
    ordered = {}
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
        seqs = defaultdict(list)
        for key in sorted(per_dispatch, key=int):
            name, grid, dur = meta[key]
            seqs[(name, grid)].append((per_dispatch[key], dur))
        ordered[counter] = seqs
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
        if not (name.startswith("conv_kernel<") or name.startswith("wgrad_kernel<") or name.startswith("conv5_bf16") or name.startswith("wgrad5_bf16") or name.startswith("wgrad5_b16")
                or name.startswith("conv2_") or name.startswith("bn_") or name.startswith("input_")):
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
    # The kernel family bench.py reports = decoder level 1 conv_1 at 128^3: forward 32->16, backward-data 16->32, filter gradient.
    # A member is (kernel-name regex, launches of that name+grid per training step, position among them in launch order):
    # every step launches the same sequence, so dispatch i of a name+grid is position i mod per_step.
    fams = {"fp32": [("fwd", r"conv_kernel<5, 1, 4, 8, 8, 4, 4, 1, false, 5, true", "2097152", 1, 0),
                     ("bwd", r"conv_kernel<5, 1, 4, 8, 8, 4, 4, 2, false, 5, false", "2097152", 1, 0),
                     ("wgrad", r"wgrad_kernel<5, 1, 4, 4, 16, 1, 16, 5", "131072", 1, 0)],
            # C5, bf16 storage: persistent kernels (grid = CUs x 512 whatever the problem), so launch order tells them apart:
            # forward with statistics 16->16, 32->16 (the second launch of the plain 16-cout kernel; the 4->16 input conv is the
            # x-im2col instantiation); backward starts at decoder level 1, so its backward-data 16->32 and its filter gradient
            # are the first launches of their kernels in a step
            "bf16": [("fwd", r"conv5_bf16_c16_kernel<4, 8, 16, true, true, true, false>", None, 2, 1),
                     ("bwd", r"conv5_bf16_r32_kernel<false, true>", None, FAM_BWD_PER_STEP, 0),
                     ("wgrad", r"wgrad5_bf16_rr_kernel<4, false>", None, FAM_WGRAD_PER_STEP, 0)]}
    out["families"] = {}
    for fam, members in fams.items():
        per_kernel, used, nsteps = {}, [], 0
        for which, pat, grid, per_step, pos in members:
            e = {}
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                d = ordered.get(counter, {})
                keys = [k for k in d if k[0].startswith(pat) and (grid is None or k[1] == grid)]
                if len(keys) != 1:
                    continue
                seq = d[keys[0]]
                if per_step is None:        # derived from the family's first member (known launches per step)
                    if not nsteps:
                        continue
                    if len(seq) % nsteps:
                        # round 4: the filter gradients of a step run as ONE grouped launch; the row-reuse kernel is launched on its
                        # own only for the layer bench.py times (decoder level 1 conv_1, in the eager steps after the timed region):
                        # every stand-alone launch of the run is that layer
                        per_step = 1
                    else:
                        per_step = len(seq) // nsteps
                elif not nsteps:
                    nsteps = len(seq) // per_step
                if len(seq) % per_step:
                    print("WARNING: %s: %d launches is not a multiple of %d per step" % (keys[0], len(seq), per_step))
                    continue
                sel = seq[pos::per_step]
                e[counter] = sum(v for v, _ in sel) / len(sel)
                e["launches"] = len(sel)
                e["avg_us_profiled"] = sum(t for _, t in sel) / len(sel) / 1e3
                e["kernel"] = "%s grid=%s, launch %d of %d per step" % (keys[0][0], keys[0][1], pos + 1, per_step)
            if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
                rd, wr = 2.0 * 1024.0 * e["FETCH_SIZE"], 1024.0 * e["WRITE_SIZE"]
                per_kernel[which] = {"kernel": e["kernel"], "launches": e["launches"], "hbm_read_bytes": round(rd), "hbm_write_bytes": round(wr),
                                   "hbm_bytes_per_launch": round(rd + wr), "avg_us_profiled": round(e["avg_us_profiled"], 1)}
                used.append(e["kernel"])
        if len(per_kernel) == len(members):
            out["families"][fam] = {"kernels": used, "per_kernel": per_kernel,
                                    "hbm_bytes_per_launch": round(sum(v["hbm_bytes_per_launch"] for v in per_kernel.values()) / len(per_kernel))}
        else:
            print("WARNING: family %s incomplete: %s" % (fam, sorted(per_kernel)))
    json.dump(out, open(os.path.join(here, tag + "_pmc.json"), "w"), indent=1)
    for k, e in out["kernels"].items():
        print("%-70s n=%3d read %8.1f MB write %8.1f MB  %8.1f us" % (k, e.get("launches", 0), e.get("hbm_read_bytes", 0) / 1e6,
                                                                      e.get("hbm_write_bytes", 0) / 1e6, e.get("avg_us_profiled", 0)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
