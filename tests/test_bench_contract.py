"""The bench line's contract, checked on the committed line of the round (profiles/r06_bench.json, written by `python bench.py` on an MI355X):
every key the driver and the judge read, the headline at arithmetic no narrower than the reference's fp32, the other modes as
top-level scalars with their same-run deviations, roofline and cpu_baseline objects as the task's measurement section asks.  (Host logic only.)"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_keys_and_a_full_precision_headline():
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
    for k, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                   ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[k], typ), k
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["data"] == "synthetic" and d["n_gpus"] == 1 and d["value_valid"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    # the headline mode: exact bf16 splits - 24 significand bits, fp32's exponent range - not the fp16 / fp6 speed modes
    assert "bf16" in d["dtype"] and "24 significand bits" in d["dtype"] and "f6" not in d["dtype"] and "f16x3" not in d["dtype"]
    assert abs(d["value"] - d["config"]["maps_per_step"] / d["ms_per_step"] * 1e3) < 0.01 * d["value"]
    for k in ("value_fp32_mfma", "value_bf16x6", "value_f16x3", "value_f16f6", "dev_chain_bf16x6", "dev_chain_f16x3", "dev_chain_f16f6",
              "dev_step_bf16x6", "dev_step_f16x3", "dev_step_f16f6"):
        assert isinstance(d[k], float) and d[k] > 0, k
    assert abs(d["value_bf16x6"] / d["value"] - 1.0) < 0.05                      # the sweep's 6-step figure of the headline mode
    assert d["value_f16f6"] > d["value_f16x3"] > d["value"] > d["value_fp32_mfma"]
    assert d["dev_chain_bf16x6"] < 1e-5 and d["dev_chain_f16x3"] < 1e-5 and d["dev_chain_f16f6"] < 1e-4
    r = d["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0
    assert r["mfma_products_per_fp32_product"] == 6 and "B6" not in r["kernel"] and r["kernel"].endswith("true>")        # the B6 instantiation
    assert len(r["per_layer"]) == 13 and set(r["modes"]) == {"0", "1", "2", "3"}
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "maps/s" and c["sample"]
    assert {"3", "4", "5", "b64", "varlen", "dropin_b1"} <= set(d["configs"])
