#!/bin/bash
# round 6: the new tests, the cost of a device filter function at -O3, the driver's bench command, the batch sweep with the lane census
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06k; mkdir -p $O; rm -f $O/*
python -m pytest tests/test_gpu_round6.py -q -s --capture=sys -m gpu > $O/tests6.log 2>&1; grep -a -v "^  File\|Extension modules" $O/tests6.log | tail -25
timeout 600 python tests/gpu_devfilter.py > $O/devfilter.log 2>&1; grep -a "DEVFILTER\|calls /" $O/devfilter.log
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc $?"; tail -3 $O/bench_driver_cmd.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06k/bench_driver_cmd.json"))
r=d["roofline"]
print("value",d["value"],"ms/step",d["ms_per_step"],"pipelined",d.get("pipelined",{}).get("value"),"build",d["build"]["gpu_build_ms"],"low",d["build"]["low_quality"]["gpu_build_ms"],"high",d["build"]["high_quality"]["gpu_build_ms"])
print("roof bound",r["bound"],"frac",r["frac"],"alg frac",r["frac_algorithmic_cache_served"],"compulsory",r["compulsory_bytes"],"traffic",r["traffic"],"first4",r["kernel_ms_first4"],"last4",r["kernel_ms_last4"])
print("small_batch",d.get("small_batch",{}).get("legs"),"e2e",d.get("end_to_end"),"latency",d.get("per_call_latency",{}).get("rtcIntersect1_us_median"))
print("parity",d.get("parity_vs_reference"),"cpu",d.get("cpu_baseline",{}).get("value"))
PY
timeout 900 python tests/gpu_batch_sweep.py --lo 12 --hi 20 --census --md > $O/sweep_census.log 2>&1; grep -a "CENSUS\|^|" $O/sweep_census.log
