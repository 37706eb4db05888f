#!/bin/bash
# development aid: kbench against ablation builds (variants/*.so built by hand with -DPIR_ABLATE_*)
for v in variants/*.so; do echo "== $v"; PIR_HIP_LIB=$PWD/$v timeout -k 10 120 python -u tools/kbench.py 2>/dev/null | grep "M=4194304" || exit 1; done
