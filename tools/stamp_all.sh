#!/bin/bash
# Phase stamps (s_memtime) of every relevance-chain kernel from the profiling builds csrc/variants/liblrpx_stamp<HW>.so
# (make STAMP=1 EXTRA=-DLRPX_STAMP_HW=<HW>): usage tools/stamp_all.sh > gpurun_out/stamps.txt
V=lrp-imagecaptioning-pytorch_amd/csrc/variants
run() { echo "== $1"; LRPX_LIB_PATH=$V/liblrpx_stamp$2.so timeout -k 10 120 python tools/stamp_probe.py lrpx_debug_stamps_$3 2>&1 | grep -v amdgpu.ids; }
run "conv1_2 (224 pooled, 4 waves)" 224 h8p &&
run "conv2_2 (112 pooled, 4 waves)" 112 h8p &&
run "conv2_1 (112 narrow, 4 waves)" 112 h8b &&
run "conv3_1 (56, 4 waves)" 56 h8 &&
run "conv3_2 (56, 8 waves)" 56 h8w &&
run "conv3_3 (56 pooled, 8 waves)" 56 h8x &&
run "conv4_1 + conv4_2 (28, 8 waves)" 28 h8w &&
run "conv4_3 (28 pooled, 8 waves)" 28 h8x &&
run "conv5_x (14, 8 waves)" 14 h8w
