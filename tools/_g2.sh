mkdir -p gpurun_out/r4b
python -m pytest tests/test_gpu_igemm.py tests/test_gpu_kernels.py -x -q > gpurun_out/r4b/kernel_tests.log 2>&1; echo "kernel tests rc=$?"
tail -15 gpurun_out/r4b/kernel_tests.log
for t in hip pf111 pf221 pf432; do TG_LIB=libtg_$t.so python tools/bench_step_shapes.py f32 gpurun_out/r4b/shapes_$t.csv > gpurun_out/r4b/shapes_$t.txt 2>&1; tail -1 gpurun_out/r4b/shapes_$t.txt; done
for t in hip pf111 pf432; do TG_LIB=libtg_$t.so python bench.py --exec plan --steps 100 --no-cpu-baseline --soak-seconds 0 > gpurun_out/r4b/bench_$t.json 2> gpurun_out/r4b/bench_$t.err; python -c "
import json;d=json.load(open('gpurun_out/r4b/bench_$t.json'));r=d['roofline'];print('$t',d['ms_per_step'],d['value'],r['all_igemm_launches'],r['all_wgrad_launches'],r['class_ms_per_step'])"; done
