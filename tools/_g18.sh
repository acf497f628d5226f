mkdir -p gpurun_out/r4r
bash tools/pmc_traffic.sh > gpurun_out/r4r/traffic.log 2>&1
python3 tools/make_traffic_json.py gpurun_out/traffic r04 > gpurun_out/r4r/r04_traffic.json 2> gpurun_out/r4r/traffic_json.err; head -c 200 gpurun_out/r4r/r04_traffic.json; echo
bash tools/bench_configs.sh 2>&1 | tail -45 > gpurun_out/r4r/configs.log; grep -v "^  " gpurun_out/r4r/configs.log | tail -12
bash tools/exposed_time.sh plan > gpurun_out/r4r/exposed_plan.txt 2>&1; head -12 gpurun_out/r4r/exposed_plan.txt
python3 tools/bench_pipeline.py > gpurun_out/r4r/pipeline.json 2> gpurun_out/r4r/pipeline.err; tail -c 300 gpurun_out/r4r/pipeline.json
