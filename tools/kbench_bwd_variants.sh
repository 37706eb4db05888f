#!/bin/bash
# development aid: backward micro-benchmark against ablation builds (variants/*.so from tools/build_variant.sh)
for v in variants/*.so; do echo "== $v"; PIR_HIP_LIB=$PWD/$v timeout -k 10 120 python -u tools/kbench_bwd.py 1048576 2>/dev/null || exit 1; done
