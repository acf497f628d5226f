mkdir -p gpurun_out/r4e
python -m pytest tests/test_gpu_goodgan.py -q -k "svhn_bf16_solver_runs" -p no:cacheprovider > gpurun_out/r4e/svhn_fixture.log 2>&1; echo "svhn fixture rc=$?"; tail -3 gpurun_out/r4e/svhn_fixture.log
bash tools/collect_profiles.sh r04 2>&1 | tail -3
bash tools/exposed_time.sh > gpurun_out/r4e/exposed.txt 2>&1; tail -5 gpurun_out/r4e/exposed.txt
