cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/gpu_run.sh r6ck "tests:fused or dropout or baseline_configs or transducer or rnnt or persistent or model" "configs:M-RNNT,M-TIMIT"
