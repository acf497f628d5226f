mkdir -p gpurun_out/r4d
python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider > gpurun_out/r4d/gpu_suite.log 2>&1; echo "gpu suite rc=$?"
tail -14 gpurun_out/r4d/gpu_suite.log
python bench.py > gpurun_out/r4d/bench.json 2> gpurun_out/r4d/bench.err; echo "bench rc=$?"
python -c "
import json;d=json.load(open('gpurun_out/r4d/bench.json'));print(d['value'],d['ms_per_step'],d['host_issue_ms_per_step'],d['config']['exec_mode_chosen'],d['config']['exec_mode_timings_ms'],d['roofline']['frac'],d['roofline']['all_igemm_launches'],d['cpu_baseline']['value'])"
for c in mnist svhn-bf16; do python tools/bench_config.py --config $c > gpurun_out/r4d/cfg_$c.json 2> gpurun_out/r4d/cfg_$c.err; python -c "
import json;d=json.load(open('gpurun_out/r4d/cfg_$c.json'));print('$c',d['ms_per_step'],d['host_issue_ms_per_step'],d['exec_mode_chosen'],d['largest_mfma_launches'][0])"; done
