"""Turns the printed counter dictionaries of profiles/pmc_one.sh (one line per kernel and pass) into the lines of profiles/rNN_pmc_f32x3.txt:
python profiles/pmc_f32x3_report.py "<label>" <pmc_one output file> [...]"""
import ast
import collections
import sys

args = sys.argv[1:]
for label, path in zip(args[0::2], args[1::2]):
    acc = collections.defaultdict(dict)
    for line in open(path):
        if "{" not in line:
            continue
        name, d = line[:line.index("{")].strip(), ast.literal_eval(line[line.index("{"):])
        acc[name].update(d)
    for name, c in acc.items():
        if "MFMA" not in "".join(c) or not c.get("SQ_INSTS_MFMA"):
            continue
        mf, wc = c["SQ_INSTS_MFMA"], c["SQ_WAVE_CYCLES"]
        print("%s | %s" % (label, name))
        print("    MFMA busy %.1f %% of SIMD-resident cycles | MFMA %d, other VALU %.2f per MFMA, LDS instr %.2f per MFMA, SALU %.2f per MFMA | "
              "LDS active %.1f %% (bank conflicts %.1f %% of it) | waves parked (s_waitcnt / barrier) %.1f %%, issue-stalled %.1f %%"
              % (100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (2.0 * wc), mf, (c["SQ_INSTS_VALU"] - mf) / mf, c["SQ_INSTS_LDS"] / mf, c["SQ_INSTS_SALU"] / mf,
                 100.0 * c["SQ_LDS_IDX_ACTIVE"] / (wc / 2.0), 100.0 * c["SQ_LDS_BANK_CONFLICT"] / max(1, c["SQ_LDS_IDX_ACTIVE"]),
                 100.0 * c["SQ_WAIT_ANY"] / wc, 100.0 * c["SQ_WAIT_INST_ANY"] / wc))
