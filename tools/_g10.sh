mkdir -p gpurun_out/r4j
python -m pytest tests -m gpu -q -x --durations=5 -p no:cacheprovider > gpurun_out/r4j/gpu_suite.log 2>&1; echo "gpu suite rc=$?"
tail -8 gpurun_out/r4j/gpu_suite.log
for f in 1 0; do TG_POOL_FUSE=$f python bench.py --exec plan --steps 100 --no-cpu-baseline --soak-seconds 0 > gpurun_out/r4j/bench_pool$f.json 2> gpurun_out/r4j/bench_pool$f.err; python -c "
import json;d=json.load(open('gpurun_out/r4j/bench_pool$f.json'));r=d['roofline'];print('cifar pool_fuse=$f',d['ms_per_step'],r['class_ms_per_step'])"; done
bash tools/pmc_step_shapes.sh > gpurun_out/r4j/pmc_shapes.log 2>&1; tail -25 gpurun_out/r4j/pmc_shapes.log
