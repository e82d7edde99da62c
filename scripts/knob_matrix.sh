#!/bin/bash
# the GPU parity suite under every tuning knob (each run is a separate process: the knobs are read at commit)
run() { echo "== $*"; env "$@" timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py tests/test_zz_no_alloc.py -x -q --timeout 200 --timeout-method thread -m gpu 2>&1 | tail -1; }
run RT_FRAME_SPLIT=2
run RT_FRAME_SPLIT=3
run RT_SHARE_LEAVES=0
run RT_MAX_LEAF=1
run RT_MAX_LEAF=4
run RT_MAX_LEAF=15
run RT_SAH_TRAV=1
run RT_SAH_BOTTOMUP=0
run RT_SAH_SWEEP=100000000
run RT_LBVH_TREELETS=0
run RT_LBVH_BIG_TREELETS=0
run RT_LBVH_TRAV=1
run RT_SORT_RAYS=0
run RT_SORT_RAYS=1 RT_SORT_MIN_LEVEL=2
run RT_SORT_RAYS=1 RT_SORT_MIN_LEVEL=1
run RT_SHADOW_PASS=1
run RT_SHADOW_PASS=1 RT_SORT_RAYS=0
run RT_SHADOW_PASS=1 RT_FRAME_SPLIT=3
