#!/bin/bash
# Register, spill, scratch and LDS figures of the render kernels in the built library
# (the code object's metadata notes): scripts/kernel_regs.sh [pattern]
set -e
cd "$(dirname "$0")/.."
OBJ=${OBJ:-raytracer_amd/build/rt_core.o}
TMP=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin $OBJ
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$TMP/fat.bin --output=$TMP/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/k.co | python3 -c "
import sys, re, subprocess
pat = sys.argv[1] if len(sys.argv) > 1 else 'k_wf|k_render_mega'
txt = sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    g = lambda k: re.search(r'\.' + k + r':\s+(\d+)', blk).group(1)
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
    if re.search(pat, dem):
        print(f\"{dem:55s} vgpr {g('vgpr_count'):>3} spill {g('vgpr_spill_count'):>3} sgpr {g('sgpr_count'):>3} scratch {g('private_segment_fixed_size'):>4} B lds {g('group_segment_fixed_size'):>5} B\")
" "$@"
rm -rf $TMP
