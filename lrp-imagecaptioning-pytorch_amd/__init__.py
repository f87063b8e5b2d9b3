"""MI355X-native LRP relevance-propagation hot path (drop-in for the reference's
`LRPtools/` hook API and the `explain_*` methods of gridTDmodel.py / aoamodel.py).

Import as `lrp_amd` (see ../lrp_amd.py).  Sub-modules are imported lazily so that the
CPU-only pieces (weights generator, host logic) work without the HIP library; anything that
computes relevance loads `csrc/liblrpx.so` and fails loudly when it is missing."""
__version__ = "0.1.0"
