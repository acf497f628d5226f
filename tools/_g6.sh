mkdir -p gpurun_out/r4f
for t in hip we221 we331 pf111; do TG_LIB=libtg_$t.so python tools/bench_step_shapes.py f32 gpurun_out/r4f/shapes_$t.csv > gpurun_out/r4f/shapes_$t.txt 2>&1; echo $t; tail -1 gpurun_out/r4f/shapes_$t.txt; done
for t in hip we221 pf111; do TG_LIB=libtg_$t.so TG_EXEC_MODE=plan python tools/bench_config.py --config svhn-bf16 > gpurun_out/r4f/svhn_$t.json 2> gpurun_out/r4f/svhn_$t.err; python -c "
import json;d=json.load(open('gpurun_out/r4f/svhn_$t.json'));print('svhn-bf16 $t',d['ms_per_step'],{k:v['ms'] for k,v in d['classes'].items()})"; done
for t in hip we221 pf111; do TG_LIB=libtg_$t.so python bench.py --exec plan --steps 100 --no-cpu-baseline --soak-seconds 0 > gpurun_out/r4f/bench_$t.json 2> gpurun_out/r4f/bench_$t.err; python -c "
import json;d=json.load(open('gpurun_out/r4f/bench_$t.json'));r=d['roofline'];print('cifar $t',d['ms_per_step'],r['all_igemm_launches']['achieved'],r['class_ms_per_step']['igemm_f32'])"; done
TG_LIB=libtg_we221.so python -m pytest tests/test_gpu_igemm.py -q -x -p no:cacheprovider 2>&1 | tail -2
