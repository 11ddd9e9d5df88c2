"""Turns the rocprofv3 CSVs of one profiling round (gpurun_out/<dir>, made by profiles/collect.sh) into the committed summaries:
  profiles/<tag>_kernel_stats_<leg>.csv   `rocprofv3 --kernel-trace --stats` of `python bench.py <leg>`, ONE LEG PER RUN (fp32 headline,
                                          x3 = fp32_split3, c5 = bf16 4-modality / 5-class, c2 = 64^3 batch 2): every row is one
                                          problem shape per kernel instantiation or -- for the persistent-grid kernels -- one leg
  profiles/<tag>_pmc.json                 FETCH_SIZE / WRITE_SIZE per launch of the dominant kernels, per leg
Usage: python profiles/summarize.py gpurun_out/prof5 r05
Round 5: rounds 1-4 profiled ONE command that ran all legs; the persistent-grid kernels (fp32 filter gradient:
131072 threads whatever the shape) then averaged the 128^3 and the 64^3 B=2 launches into one row.  Family members are now picked
per leg by (kernel, grid, position in the step's launch order), the step count taken from the once-per-step softmax_dice_bwd kernel.
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-reports a wide coalesced stream by 2x (MI355X_MICROARCH.md, HBM
section), so the read side is doubled; WRITE_SIZE is taken as reported (uncalibrated)."""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

LEGS = ("fp32", "x3", "c5", "c2")
KEEP = ("conv_kernel<", "wgrad_kernel<", "conv5_bf16", "wgrad5_bf16", "wgrad5_b16", "conv5_x3", "wgrad5_x3", "conv2_", "bn_", "input_")

# The kernel family bench.py reports = decoder level 1 conv_1 at 128^3: forward 32->16, backward-data 16->32, filter gradient.
# A member: (role, kernel-name prefix, grid or None, position in the step: an index into that kernel's launches of one step, or
# "longest").  Every step launches the same sequence, so the launches of a (name, grid) split into nsteps equal groups.
FAMILIES = {
    "fp32": ("fp32", [("fwd", "conv_kernel<5, 1, 4, 8, 8, 4, 4, 1, false, 5, true", "2097152", 0),
                      ("bwd", "conv_kernel<5, 1, 4, 8, 8, 4, 4, 2, false, 5, false", "2097152", 0),
                      ("wgrad", "wgrad_kernel<5, 1, 4, 4, 16, 1, 16, 5", "131072", 0)]),
    # fp32_split3: persistent kernels (grid = CUs x 512 whatever the problem), and the plain instantiation serves the K-split forward
    # launches of the 16^3 level as well as every backward-data launch: a member is the LONGEST launch of its kernel in a step -- the
    # 128^3 launches of dec1/conv_1 (268 GF each) take twice as long as any other layer's
    # (<STATS, NB, W8>: dec1/conv_1 has 16 output channels forward: one cout block per item; backward-data its two 16-channel destinations
    #  are a pair of cout blocks since round 6)
    "f32x3": ("x3", [("fwd", "conv5_x3_kernel<true, 1, false>", None, "longest"),
                     ("bwd", "conv5_x3_kernel<false, 2, false>", None, "longest"),
                     ("wgrad", "wgrad5_x3_kernel<false>", None, "longest")]),
    # C5, bf16 storage: forward with statistics 16->16, 32->16 = the second launch of the plain 16-cout kernel (the 4->16 input conv is
    # the x-im2col instantiation); backward-data 16->32 and the stand-alone filter gradient are the first launches of their kernels
    # (round 5: forward 16->16 / 32->16 with statistics take conv5_bf16_c16pp_kernel<true> -- persistent grid, the 4->16 input conv stays on the
    #  x-im2col instantiation of the c16 kernel: the 32->16 launch is the LONGEST of its kernel in a step)
    "bf16": ("c5", [("fwd", "conv5_bf16_c16pp_kernel<true>", None, "longest"),
                    ("bwd", "conv5_bf16_r32_kernel<false", None, 0),       # (round 6: the O16 template argument is gone)
                    ("wgrad", "wgrad5_bf16_rr_kernel<4, false>", None, 0)]),
}


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def read_pass(path, counter):
    """{(kernel, grid): [(value, duration ns), ...] in dispatch order}; several rows per dispatch (one per XCD) are summed."""
    per_dispatch, meta = defaultdict(float), {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        key = int(r["Dispatch_Id"])
        per_dispatch[key] += float(r["Counter_Value"])
        meta[key] = (short(r["Kernel_Name"]), r["Grid_Size"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    seqs = defaultdict(list)
    for key in sorted(per_dispatch):
        name, grid, dur = meta[key]
        seqs[(name, grid)].append((per_dispatch[key], dur))
    return seqs


def family_dispatches(src, here, tag):
    """Round 6 (VERDICT r5 #7b): the family's launches one row per DISPATCH from the counter-free `--kernel-trace --stats` run of each
    leg (<leg>_kernel_trace.csv), selected by (kernel, position in the step / longest launch of the step) like the PMC passes -- so the
    `avg_ms` of the bench line can be re-derived to a few percent from a run WITHOUT counters (the --stats rows of a persistent-grid
    kernel average every layer that kernel serves).  -> profiles/<tag>_family_dispatches.csv, and a summary on stdout."""
    rows = []
    for fam, (leg, members) in FAMILIES.items():
        path = os.path.join(src, leg + "_kernel_trace.csv")
        if not os.path.exists(path):
            continue
        seqs, steps = defaultdict(list), 0
        trace = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
        for r in trace:
            name = short(r["Kernel_Name"])
            if name.startswith("softmax_dice_bwd_kernel"):
                steps += 1
            grid = r.get("Grid_Size") or str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))      # (the trace has X / Y / Z)
            seqs[(name, grid)].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for role, pat, grid, pos in members:
            keys = [k for k in seqs if k[0].startswith(pat) and (grid is None or k[1] == grid)]
            if len(keys) != 1 or not steps:
                print("WARNING: %s/%s: %d kernels match" % (fam, role, len(keys)))
                continue
            seq = seqs[keys[0]]
            per_step = len(seq) // steps if len(seq) % steps == 0 else 1
            if pos == "longest":
                sel = [max(seq[i:i + per_step], key=lambda t: t[1]) for i in range(0, len(seq), per_step)]
            else:
                sel = seq[pos % per_step::per_step]
            for i, (t0, dur) in enumerate(sel):
                rows.append((fam, leg, role, keys[0][0], i, dur))
            d = sorted(t[1] for t in sel)
            print("%-6s %-6s %-40s n=%3d  avg %8.1f us  median %8.1f  min %8.1f  max %8.1f" %
                  (fam, role, keys[0][0][:40], len(d), sum(d) / len(d) / 1e3, d[len(d) // 2] / 1e3, d[0] / 1e3, d[-1] / 1e3))
    if rows:
        with open(os.path.join(here, tag + "_family_dispatches.csv"), "w") as f:
            f.write("family,leg,role,kernel,launch_index,duration_ns\n")
            for r in rows:
                f.write('%s,%s,%s,"%s",%d,%d\n' % r)


def main(src, tag):
    here = os.path.dirname(os.path.abspath(__file__))
    family_dispatches(src, here, tag)
    out = {"note": "per launch; KB from rocprofv3 --pmc (separate passes, one leg per run); fetch doubled per the gfx950 correction",
           "legs": {}, "families": {}}
    for leg in LEGS:
        st = os.path.join(src, leg + "_kernel_stats.csv")
        if os.path.exists(st):
            shutil.copy(st, os.path.join(here, "%s_kernel_stats_%s.csv" % (tag, leg)))
        seqs = {}
        for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            path = os.path.join(src, "%s_%s_counter_collection.csv" % (leg, which))
            if os.path.exists(path):
                seqs[counter] = read_pass(path, counter)
        if not seqs:
            continue
        kern = {}
        for key in sorted(set().union(*[set(s) for s in seqs.values()])):
            name, grid = key
            if not name.startswith(KEEP):
                continue
            e = {"grid_threads": int(grid)}
            if key in seqs.get("FETCH_SIZE", {}):
                sq = seqs["FETCH_SIZE"][key]
                e["launches"] = len(sq)
                e["hbm_read_bytes"] = 2.0 * 1024.0 * sum(v for v, _ in sq) / len(sq)
                e["avg_us_profiled"] = sum(d for _, d in sq) / len(sq) / 1e3
            if key in seqs.get("WRITE_SIZE", {}):
                sq = seqs["WRITE_SIZE"][key]
                e["hbm_write_bytes"] = 1024.0 * sum(v for v, _ in sq) / len(sq)
            e["hbm_bytes_per_launch"] = e.get("hbm_read_bytes", 0.0) + e.get("hbm_write_bytes", 0.0)
            kern["%s grid=%s" % (name, grid)] = e
        out["legs"][leg] = kern
        # training steps of each pass: the once-per-step kernel (round 6: PER PASS -- the eager steps that run while rocm-smi is read
        # make the step count of two passes differ)
        steps_of = {c: sum(len(v) for k, v in sq.items() if k[0].startswith("softmax_dice_bwd_kernel")) for c, sq in seqs.items()}
        nsteps = min(steps_of.values())
        for fam, (fleg, members) in FAMILIES.items():
            if fleg != leg:
                continue
            per_kernel, used = {}, []
            for role, pat, grid, pos in members:
                e = {}
                for counter, sq in seqs.items():
                    keys = [k for k in sq if k[0].startswith(pat) and (grid is None or k[1] == grid)]
                    nsteps = steps_of[counter]
                    if len(keys) != 1 or not nsteps:
                        continue
                    seq = sq[keys[0]]
                    if len(seq) % nsteps:
                        # (C5: the stand-alone filter gradient exists only in the eager steps that time the family -- every launch is that layer)
                        per_step = 1
                    else:
                        per_step = len(seq) // nsteps
                    if pos == "longest":
                        sel = [max(seq[i:i + per_step], key=lambda vt: vt[1]) for i in range(0, len(seq), per_step)]
                        where = "the longest of %d launches per step" % per_step
                    else:
                        sel = seq[pos % per_step::per_step]
                        where = "launch %d of %d per step" % (pos % per_step + 1, per_step)
                    e[counter] = sum(v for v, _ in sel) / len(sel)
                    e["launches"] = len(sel)
                    e["avg_us_profiled"] = sum(t for _, t in sel) / len(sel) / 1e3
                    e["kernel"] = "%s grid=%s, %s" % (keys[0][0], keys[0][1], where)
                if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
                    rd, wr = 2.0 * 1024.0 * e["FETCH_SIZE"], 1024.0 * e["WRITE_SIZE"]
                    per_kernel[role] = {"kernel": e["kernel"], "launches": e["launches"], "hbm_read_bytes": round(rd), "hbm_write_bytes": round(wr),
                                        "hbm_bytes_per_launch": round(rd + wr), "avg_us_profiled": round(e["avg_us_profiled"], 1)}
                    used.append(e["kernel"])
            if len(per_kernel) == len(members):
                out["families"][fam] = {"leg": leg, "steps_profiled": min(steps_of.values()), "kernels": used, "per_kernel": per_kernel,
                                        "hbm_bytes_per_launch": round(sum(v["hbm_bytes_per_launch"] for v in per_kernel.values()) / len(per_kernel))}
            else:
                print("WARNING: family %s incomplete: %s" % (fam, sorted(per_kernel)))
    json.dump(out, open(os.path.join(here, tag + "_pmc.json"), "w"), indent=1)
    for leg, kern in out["legs"].items():
        for k, e in kern.items():
            print("%-5s %-72s n=%3d read %8.1f MB write %8.1f MB  %8.1f us" % (leg, k[:72], e.get("launches", 0), e.get("hbm_read_bytes", 0) / 1e6,
                                                                               e.get("hbm_write_bytes", 0) / 1e6, e.get("avg_us_profiled", 0)))
    for fam, f in out["families"].items():
        print(fam, {r: (v["hbm_bytes_per_launch"] / 1e6, v["avg_us_profiled"], v["launches"]) for r, v in f["per_kernel"].items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
