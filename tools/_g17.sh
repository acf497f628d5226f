mkdir -p gpurun_out/r4q
python -m pytest tests -m gpu -q -x --durations=5 -p no:cacheprovider > gpurun_out/r4q/gpu_suite.log 2>&1; echo "gpu suite rc=$?"
tail -6 gpurun_out/r4q/gpu_suite.log
bash tools/collect_profiles.sh r04 2>&1 | tail -2
