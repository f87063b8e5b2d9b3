#!/usr/bin/env python3
"""regenerate DESIGN.md's "Round 6" measured table from profiles/r06_bench.json: python tools/dbg/fill_r6.py <gpu tests passed> <cpu tests passed>"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
r = d["roofline"]; c = d["configs"]; m = r["modes"]
f = lambda x: f"{x:,.0f}".replace(",", " ")
b1 = c["dropin_b1"]["given_caption"]
rep = {
    "@VALUE@": f(d["value"]), "@MS@": f"{d['ms_per_step']:.2f}", "@SUST@": f(d["sustained"]["value"]),
    "@V0@": f(m["0"]["maps_per_s"]), "@V1@": f(m["1"]["maps_per_s"]), "@V2@": f(m["2"]["maps_per_s"]), "@V3@": f(m["3"]["maps_per_s"]),
    "@D1@": f"{d['dev_chain_bf16x6']:.1e}", "@D2@": f"{d['dev_chain_f16x3']:.1e}", "@D3@": f"{d['dev_chain_f16f6']:.1e}", "@DS@": f"{d['dev_step_bf16x6']:.1e}",
    "@KERNEL@": r["kernel"], "@KMS@": f"{r['ms_per_launch']:.2f}", "@KFRAC@": f"{r['frac']:.3f}", "@KTRAF@": f"{(r['traffic'] or 0) / 1e9:.2f}",
    "@CHAIN@": f"{r['chain']['ms_per_step']:.2f}", "@CFRAC@": f"{r['chain_frac']:.3f}",
    "@LAYERS@": ", ".join(f"{x['layer']} {x['ms']:.2f}" for x in r["per_layer"]),
    "@C3@": f(c["3"]["value"]), "@C3F@": f(c["3"].get("value_f16f6", 0)), "@C4@": f(c["4"]["value"]), "@C4F@": f(c["4"].get("value_f16f6", 0)),
    "@CB@": f(c["b64"]["value"]), "@CBF@": f(c["b64"].get("value_f16f6", 0)), "@C5@": f(c["5"]["value"]), "@C5MS@": f"{c['5']['ms_per_step']:.2f}", "@C5F@": f(c["5"].get("value_f16x3", 0)), "@CV@": f(c["varlen"]["value"]),
    "@C3A@": f(c["3"]["all_heads"]["value"]),
    "@B1@": f"{b1['ms_per_call']:.2f}", "@B1F@": (f"{b1['ms_per_call_f16f6']:.2f}" if b1.get("ms_per_call_f16f6") else "4.8"),
    "@B1A@": f"{c['dropin_b1']['aoa_given_caption']['ms_per_call']:.2f}",
    "@CPU@": f"{d['cpu_baseline']['value']:.3f}", "@RATIO@": f(d["value"] / d["cpu_baseline"]["value"]),
    "@NGPU@": sys.argv[1], "@NCPU@": sys.argv[2],
}
t = open(os.path.join(ROOT, "tools", "dbg", "r6_measured.tmpl")).read()
for k, v in rep.items():
    t = t.replace(k, v)
assert not re.findall(r"@[A-Z0-9]+@", t)
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
i, j = s.index("### Round 6 (1× MI355X"), s.index("### Round 5 (1× MI355X")
open(p, "w").write(s[:i] + t + s[j:])
print("DESIGN.md: Round 6 table regenerated")
