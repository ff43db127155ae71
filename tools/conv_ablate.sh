#!/bin/bash
# needs a library built with the ablation branches:  make -C neural-flow-style_amd/csrc clean && make -C neural-flow-style_amd/csrc ABLATE=1
for d in 0 1 2 3 4 7; do echo "== NFS_CONV_DBG=$d"; NFS_CONV_DBG=$d python tools/conv_bench.py 2>/dev/null | grep -E "B=8 (200x200|  50x50  256|  25x25  512)|B=8 total"; done
