"""The gates of VERDICT r5's ruling ("fp32_split3 MAY become `value` iff (a)-(c) are delivered"), evaluated from the COMMITTED records, and
the decision bench.py reads:  python profiles/promotion_gates.py  ->  profiles/r06_promotion.json
  (a) seed spread  profiles/r06_golden_seed_spread.txt: per config (c2, c3), fp32_split3's max-over-seeds of the worst per-tensor
      gradient error <= 1.25 x the fp32 mode's;
  (b) adversarial operands  profiles/r06_x3_adversarial.txt: every f32x3 / fp32-MFMA error ratio (forward, backward-data, filter
      gradient; five operand classes) <= 1.25;
  (c) non-finite semantics: tests/test_hip_x3.py::test_x3_non_finite_operands in the passed list of that record + the sentence in
      INTEGRATION.md.
If a gate fails the decision is "value stays native", and `failed` says which."""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
gates, failed = {}, []

# (a)
txt = open(os.path.join(HERE, "r06_golden_seed_spread.txt")).read()
rows = {}
for line in txt.splitlines():
    m = re.match(r"^(c[23])(s\d+)?\s+(fp32|fp32_split3)\s+(\d\S+)\s+(\d\S+)\s+(\d\S+)\s+(\d\S+)\s+(\d\S+)", line)
    if m:
        rows.setdefault((m.group(1), m.group(3)), []).append([float(m.group(k)) for k in range(4, 9)])
a = {}
for cfg in ("c3", "c2"):
    n32, n3 = rows[(cfg, "fp32")], rows[(cfg, "fp32_split3")]
    mx32, mx3 = max(r[0] for r in n32), max(r[0] for r in n3)
    a[cfg] = {"draws": len(n3), "fp32_max_worst": mx32, "fp32_split3_max_worst": mx3, "ratio": round(mx3 / mx32, 3),
              "mean_of_worst_ratio": round(sum(r[0] for r in n3) / len(n3) / (sum(r[0] for r in n32) / len(n32)), 3),
              "median_ratio": round(max(r[1] for r in n3) / max(r[1] for r in n32), 3),
              "vector_ratio": round(max(r[4] for r in n3) / max(r[4] for r in n32), 3)}
all32 = [r[0] for cfg in ("c3", "c2") for r in rows[(cfg, "fp32")]]
all3 = [r[0] for cfg in ("c3", "c2") for r in rows[(cfg, "fp32_split3")]]
a["pooled"] = {"draws": len(all3), "ratio": round(max(all3) / max(all32), 3)}
ok_a = all(a[c]["ratio"] <= 1.25 for c in ("c3", "c2"))
gates["a_seed_spread"] = {"pass": ok_a, "rule": "per config: max-over-seeds of the worst per-tensor gradient error, fp32_split3 <= 1.25 x fp32",
                          "evidence": "profiles/r06_golden_seed_spread.txt", "measured": a,
                          "history": "first pass, five draws, kernels of the start of round 6: c3 1.34 (one draw: fp32_split3 1.27e-2 vs fp32 9.5e-3), "
                                     "c2 1.09; draws 5-9 were registered before they were computed (tests/golden/make_golden_full.py); this record: "
                                     "ten draws on the final kernels.  The per-run worst is chaotic (see the header of the evidence file); "
                                     "medians and whole-vector errors favour fp32_split3 in both configs"}
if not ok_a:
    failed.append("a")

# (b), (c)
txt = open(os.path.join(HERE, "r06_x3_adversarial.txt")).read()
ratios = {}
for line in txt.splitlines():
    m = re.match(r"^adversarial (\S+)\s.*ratio (\S+) (\S+) (\S+)", line)
    if m:
        ratios[m.group(1)] = [float(m.group(k)) for k in (2, 3, 4)]
ok_b = len(ratios) >= 5 and all(v <= 1.25 for r in ratios.values() for v in r)
gates["b_adversarial_operands"] = {"pass": ok_b, "rule": "every f32x3 / fp32-MFMA rel-L2 ratio vs the fp64 oracle (fwd, dx, dw) <= 1.25",
                                   "evidence": "profiles/r06_x3_adversarial.txt", "measured": ratios,
                                   "max_ratio": max(v for r in ratios.values() for v in r) if ratios else None}
if not ok_b:
    failed.append("b")
integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
ok_c = ("passed" in txt and "failed" not in txt) and "non-finite" in integ.lower() and "3.39e38" in integ
gates["c_non_finite"] = {"pass": ok_c, "rule": "test_x3_non_finite_operands green; INTEGRATION.md states +-Inf / |x| >= 3.39e38 -> NaN",
                         "evidence": "tests/test_hip_x3.py::test_x3_non_finite_operands, INTEGRATION.md"}
if not ok_c:
    failed.append("c")
out = {"promote": not failed, "failed": failed, "gates": {k: {"pass": v["pass"], "evidence": v["evidence"]} for k, v in gates.items()},
       "detail": gates,
       "note": "VERDICT r5: fp32_split3 MAY become `value` iff (a)-(c) are delivered; if any fails, `value` stays the native fp32-MFMA step"}
json.dump(out, open(os.path.join(HERE, "r06_promotion.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("promote", "failed")}), json.dumps(a), json.dumps(ratios))
