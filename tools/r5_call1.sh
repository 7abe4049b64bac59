#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5c1; mkdir -p $O
timeout 300 python tools/gru_fwd_variants.py 4 3 > $O/fwd_variants_L4.log 2>&1; tail -16 $O/fwd_variants_L4.log
timeout 200 python tools/gru_fwd_variants.py 2 2 > $O/fwd_variants_L2.log 2>&1; tail -8 $O/fwd_variants_L2.log
timeout 200 python tools/gru_fwd_variants.py 1 2 > $O/fwd_variants_L1.log 2>&1; tail -8 $O/fwd_variants_L1.log
bash tools/ab_env.sh 2 "SPEECH_AMD_LIB=$PWD/tools/ab/libspeech_amd_r4.so" "SA_GRU_FWD_R4=1" "-" "SA_GRU_FWD_POLLAT=4" "SA_GRU_FWD_POLLAT=6" 2>&1 | tee $O/ab.log
( time timeout 900 python -m pytest tests -m gpu -x -q -k "fused or persist or xcd or wavefront or dropout" ) > $O/pytest_gru.log 2>&1; tail -5 $O/pytest_gru.log
