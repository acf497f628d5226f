mkdir -p gpurun_out/r4c
python -m pytest tests -m gpu -q --durations=12 -p no:cacheprovider > gpurun_out/r4c/gpu_suite.log 2>&1; echo "gpu suite rc=$?"
tail -40 gpurun_out/r4c/gpu_suite.log
