mkdir -p gpurun_out/r4j; O=gpurun_out/r4j
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log | grep -v "^$"
bash tools/profile_configs.sh r04 cfg4_bf16 cfg3_bf16
