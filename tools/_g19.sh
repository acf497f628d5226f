mkdir -p gpurun_out/r4s
for m in 7 0 4 3; do TG_IGEMM_SPLIT_MASK=$m timeout -k 10 300 python bench.py --exec plan --steps 100 --no-cpu-baseline --soak-seconds 0 > gpurun_out/r4s/bench_split$m.json 2> gpurun_out/r4s/bench_split$m.err; python -c "
import json;d=json.load(open('gpurun_out/r4s/bench_split$m.json'));r=d['roofline'];print('split mask $m',d['ms_per_step'],r['all_igemm_launches']['achieved'],r['class_ms_per_step']['igemm_f32'],r['all_igemm_launches']['launches_per_step'])"; done
