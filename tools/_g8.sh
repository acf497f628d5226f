mkdir -p gpurun_out/r4h
bash tools/bench_configs.sh 2>&1 | tail -40 > gpurun_out/r4h/configs.log; tail -12 gpurun_out/r4h/configs.log
bash tools/pmc_traffic.sh > gpurun_out/r4h/traffic.log 2>&1; tail -3 gpurun_out/r4h/traffic.log
python3 tools/make_traffic_json.py gpurun_out/traffic r04 > gpurun_out/r4h/r04_traffic.json 2> gpurun_out/r4h/traffic_json.err; head -c 600 gpurun_out/r4h/r04_traffic.json
python3 tools/bench_pipeline.py > gpurun_out/r4h/pipeline.json 2> gpurun_out/r4h/pipeline.err; cat gpurun_out/r4h/pipeline.json | head -c 600
