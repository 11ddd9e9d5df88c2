#!/bin/bash
# Compiles every translation unit of libvnet_hip.so to gfx950 ISA (hipcc cross-compiles without a GPU) and reads the code-object
# metadata of each kernel: fails (exit 1) when a convolution kernel (conv* / wgrad*) executes scratch instructions between its first and last MFMA or
# spills more than 8 VGPRs (a private segment with no scratch instruction is the frame of SGPRs spilled into VGPR lanes: reported, not failed).  Usage: bash profiles/check_isa.sh [outfile]   (default: profiles/r06_check_isa.txt)
cd "$(dirname "$0")/../vnet_tensorflow_amd/csrc"
OUT=${1:-../../profiles/r06_check_isa.txt}
TMP=$(mktemp -d)
for f in conv_mfma conv_x3 conv_b16 conv2_b16 elementwise input_block; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -S --cuda-device-only $f.hip -o $TMP/$f.s 2>/dev/null &
done
wait
python3 - "$TMP" > "$OUT" <<'PY'
import glob, os, re, subprocess, sys
bad = 0
rows = []
for path in sorted(glob.glob(os.path.join(sys.argv[1], "*.s"))):
    txt = open(path).read()
    # scratch instructions per kernel body (label ... .Lfunc_end)
    nscr = {}
    hot = {}
    for m in re.finditer(r"^(_Z\w+):.*?^\.Lfunc_end\d+:", txt, re.S | re.M):
        nscr[m.group(1)] = len(re.findall(r"^\s+scratch_(?:load|store)", m.group(0), re.M))
        # scratch instructions BETWEEN the first and the last MFMA of the body (the loops): the ones that cost time
        lines = m.group(0).split("\n")
        mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
        hot[m.group(1)] = sum(1 for i, l in enumerate(lines) if mf and mf[0] < i < mf[-1] and re.match(r"\s+scratch_(?:load|store)", l))
    meta = txt[txt.rfind("amdhsa.kernels:"):]
    for blk in meta.split("  - .agpr_count:")[1:]:
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
        mangled = g("name")
        name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0] if not name.startswith("_Z") else name
        spill, sspill, priv, vg, sg = (int(g("vgpr_spill_count")), int(g("sgpr_spill_count")), int(g("private_segment_fixed_size")),
                                       int(g("vgpr_count")), int(g("sgpr_count")))
        scr = nscr.get(mangled, 0)
        conv = name.startswith(("conv", "wgrad"))
        # a private segment WITHOUT scratch instructions is the frame hipcc reserves when it spills SGPRs into VGPR lanes
        # (v_writelane / v_readlane outside the loops): no memory traffic
        hscr = hot.get(mangled, 0)
        # (round 5) a few dwords spilled in a kernel's prologue and reloaded in its epilogue, none inside the MFMA loops: reported, not failed
        flag = ("FAIL" if conv and (hscr or spill > 8) else ("outside-loops" if conv and (spill or scr) else
                ("sgpr-lanes" if priv and not scr else ("note" if (spill or scr) else "ok"))))
        bad += flag == "FAIL"
        rows.append((os.path.basename(path)[:-2], name, vg, sg, spill, sspill, priv, scr, flag))
print("%-12s %-78s %5s %5s %7s %7s %8s %8s  %s" % ("unit", "kernel", "vgpr", "sgpr", "vspill", "sspill", "private", "scratch", ""))
for r in rows:
    print("%-12s %-78s %5d %5d %7d %7d %8d %8d  %s" % r)
print("convolution kernels with VGPR spills or scratch instructions: %d" % bad)
sys.exit(1 if bad else 0)
PY
rc=$?
rm -rf "$TMP"
tail -1 "$OUT"
exit $rc
