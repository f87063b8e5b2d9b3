#!/bin/bash
# Phase stamps of the four 4-wave relevance layers (conv1_2, conv2_1, conv2_2, conv3_1) from three profiling builds
# (make STAMP=1 EXTRA=-DLRPX_STAMP_HW=224|112|56, copied to csrc/variants/liblrpx_stamp<HW>.so)
V=lrp-imagecaptioning-pytorch_amd/csrc/variants
mkdir -p gpurun_out
{
echo "== conv1_2 (224 pooled-input)"; LRPX_LIB_PATH=$V/liblrpx_stamp224.so timeout -k 10 200 python tools/stamp_probe.py lrpx_debug_stamps_h8p &&
echo "== conv2_2 (112 pooled-input)"; LRPX_LIB_PATH=$V/liblrpx_stamp112.so timeout -k 10 200 python tools/stamp_probe.py lrpx_debug_stamps_h8p &&
echo "== conv2_1 (112 narrow)"; LRPX_LIB_PATH=$V/liblrpx_stamp112.so timeout -k 10 200 python tools/stamp_probe.py lrpx_debug_stamps_h8b &&
echo "== conv3_1 (56, 4 waves)"; LRPX_LIB_PATH=$V/liblrpx_stamp56.so timeout -k 10 200 python tools/stamp_probe.py lrpx_debug_stamps_h8 &&
echo "== conv3_2/3 (56, 8 waves)"; LRPX_LIB_PATH=$V/liblrpx_stamp56.so timeout -k 10 200 python tools/stamp_probe.py lrpx_debug_stamps_h8w
} > gpurun_out/stamps_4wave.txt 2>&1
