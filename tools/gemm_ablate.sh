#!/bin/bash
# Where does the fp32 GEMM main loop lose its matrix-pipe time?  Builds ablated copies of gemm_f32.hip on the GPU box
# (the product source stays clean: the variants are sed edits; their RESULTS ARE WRONG, only their timing is read):
#   noload    the in-loop global loads of the next tile are dropped (stale registers go to LDS)
#   nobarrier the in-loop __syncthreads() is dropped
#   nostore   the in-loop LDS stores of the next tile are dropped
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/ablate; mkdir -p $O /tmp/abl
C=speech_amd/csrc
objs=$(ls $C/*.o | grep -v gemm_f32.o)
mk() { # name sed-script
  sed "$2" $C/gemm_f32.hip > /tmp/abl/gemm_$1.hip
  cp $C/common.h $C/internal.h /tmp/abl/ 2>/dev/null
  sed -i 's#"../../include/speech_amd.h"#"'$R'/include/speech_amd.h"#' /tmp/abl/common.h
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c /tmp/abl/gemm_$1.hip -o /tmp/abl/gemm_$1.o || return 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/abl/lib_$1.so /tmp/abl/gemm_$1.o $objs
}
mk base 's/XXXX/XXXX/' &
mk noload '/global loads of the next tile: issued first/,/^        }/{/load_tile/d}' &
mk nobarrier 's|^        __syncthreads(); *// the other buffer.*$|        // (ablated)|' &
mk nostore '/next tile -> the other LDS buffer/,/^        }/{/store_tile/d}' &
wait
for v in base noload nobarrier nostore; do
  echo "== $v"; grep -c "load_tile\|store_tile\|__syncthreads" /tmp/abl/gemm_$v.hip
  SPEECH_AMD_LIB=/tmp/abl/lib_$v.so timeout 200 python tools/gemm_bench.py 2>&1 | grep -E "square|dW_ih/hh|dmid|dx layer0|i2h layer0" | tee $O/$v.txt
done
