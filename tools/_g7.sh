mkdir -p gpurun_out/r4g
for t in hip b222 b332 b442; do TG_LIB=libtg_$t.so python tools/bench_step_shapes.py bf16 gpurun_out/r4g/shapes_bf16_$t.csv > gpurun_out/r4g/shapes_bf16_$t.txt 2>&1; echo $t; tail -1 gpurun_out/r4g/shapes_bf16_$t.txt; done
for t in hip b222 b332 b442; do TG_LIB=libtg_$t.so TG_EXEC_MODE=plan python tools/bench_config.py --config svhn-bf16 > gpurun_out/r4g/svhn_$t.json 2> gpurun_out/r4g/svhn_$t.err; python -c "
import json;d=json.load(open('gpurun_out/r4g/svhn_$t.json'));print('svhn-bf16 $t',d['ms_per_step'],{k:v['ms'] for k,v in d['classes'].items()})"; done
TG_LIB=libtg_b442.so python -m pytest tests/test_gpu_igemm.py -q -x -p no:cacheprovider 2>&1 | tail -2
