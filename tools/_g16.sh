mkdir -p gpurun_out/r4p
timeout -k 10 300 python -m pytest tests/test_gpu_igemm.py tests/test_gpu_kernels.py -q -x -p no:cacheprovider 2>&1 | tail -3
for t in prev hip; do TG_LIB=libtg_$t.so timeout -k 10 200 python tools/bench_step_shapes.py f32 gpurun_out/r4p/shapes_$t.csv > gpurun_out/r4p/shapes_$t.txt 2>&1; echo $t; tail -1 gpurun_out/r4p/shapes_$t.txt; done
for t in prev hip; do TG_LIB=libtg_$t.so timeout -k 10 300 python bench.py --exec plan --steps 100 --no-cpu-baseline --soak-seconds 0 > gpurun_out/r4p/bench_$t.json 2> gpurun_out/r4p/bench_$t.err; python -c "
import json;d=json.load(open('gpurun_out/r4p/bench_$t.json'));r=d['roofline'];print('cifar $t',d['ms_per_step'],r['all_igemm_launches']['achieved'],r['class_ms_per_step']['igemm_f32'])"; done
