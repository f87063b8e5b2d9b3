"""Per-model explainers (mirror of the `Explain*` classes of models/gridTDmodel.py / models/aoamodel.py)."""
