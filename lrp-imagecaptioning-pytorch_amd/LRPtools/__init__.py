"""Drop-in mirror of the reference's `LRPtools` package (hook API), backed by liblrpx.so."""
