mkdir -p gpurun_out/r4m
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4m/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r4m/smoke.log
bash tools/collect_profiles.sh r04 2>&1 | tail -2
bash tools/pmc_traffic.sh > gpurun_out/r4m/traffic.log 2>&1
python3 tools/make_traffic_json.py gpurun_out/traffic r04 > gpurun_out/r4m/r04_traffic.json 2> gpurun_out/r4m/traffic_json.err; head -c 300 gpurun_out/r4m/r04_traffic.json; echo
bash tools/exposed_time.sh plan > gpurun_out/r4m/exposed_plan.txt 2>&1; head -8 gpurun_out/r4m/exposed_plan.txt
bash tools/rehearse_bench_n2.sh > gpurun_out/r4m/rehearse_n2.log 2>&1; echo "rehearse rc=$?"; tail -c 600 gpurun_out/r4m/rehearse_n2.log
