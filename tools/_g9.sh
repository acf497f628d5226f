mkdir -p gpurun_out/r4i
python -m pytest tests -m gpu -q -x --durations=5 -p no:cacheprovider > gpurun_out/r4i/gpu_suite.log 2>&1; echo "gpu suite rc=$?"
tail -12 gpurun_out/r4i/gpu_suite.log
for f in 1 0; do TG_CONCAT_FUSE=$f python bench.py --exec plan --steps 100 --no-cpu-baseline --soak-seconds 0 > gpurun_out/r4i/bench_fuse$f.json 2> gpurun_out/r4i/bench_fuse$f.err; python -c "
import json;d=json.load(open('gpurun_out/r4i/bench_fuse$f.json'));r=d['roofline'];print('cifar fuse=$f',d['ms_per_step'],r['class_ms_per_step'])"; done
for f in 1 0; do TG_CONCAT_FUSE=$f TG_EXEC_MODE=plan python tools/bench_config.py --config svhn-bf16 > gpurun_out/r4i/svhn_fuse$f.json 2> gpurun_out/r4i/svhn_fuse$f.err; python -c "
import json;d=json.load(open('gpurun_out/r4i/svhn_fuse$f.json'));print('svhn-bf16 fuse=$f',d['ms_per_step'],{k:v['ms'] for k,v in d['classes'].items()})"; done
