mkdir -p gpurun_out/r4k
TG_LIB=libtg_role1.so timeout -k 10 300 python -m pytest tests/test_gpu_igemm.py tests/test_gpu_kernels.py -q -x -p no:cacheprovider 2>&1 | tail -3
for t in hip role1 role1p2 role3; do TG_LIB=libtg_$t.so timeout -k 10 200 python tools/bench_step_shapes.py f32 gpurun_out/r4k/shapes_$t.csv > gpurun_out/r4k/shapes_$t.txt 2>&1; echo $t; tail -1 gpurun_out/r4k/shapes_$t.txt; done
for t in hip role1 role3; do TG_LIB=libtg_$t.so timeout -k 10 300 python bench.py --exec plan --steps 100 --no-cpu-baseline --soak-seconds 0 > gpurun_out/r4k/bench_$t.json 2> gpurun_out/r4k/bench_$t.err; python -c "
import json;d=json.load(open('gpurun_out/r4k/bench_$t.json'));r=d['roofline'];print('cifar $t',d['ms_per_step'],r['all_igemm_launches']['achieved'],r['class_ms_per_step']['igemm_f32'])"; done
for t in hip role1; do TG_LIB=libtg_$t.so TG_EXEC_MODE=plan timeout -k 10 300 python tools/bench_config.py --config svhn-bf16 > gpurun_out/r4k/svhn_$t.json 2> gpurun_out/r4k/svhn_$t.err; python -c "
import json;d=json.load(open('gpurun_out/r4k/svhn_$t.json'));print('svhn-bf16 $t',d['ms_per_step'],{k:v['ms'] for k,v in d['classes'].items()})"; done
