for g in 32 16 12 8; do MI355_REFILL_MIN=$g python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r2_bench_g$g.json 2>gpurun_out/r2_bench_g$g.err; done
python - <<'P'
import json
for g in (32,16,12,8):
    try:
        d=json.loads(open("gpurun_out/r2_bench_g%d.json"%g).read().strip().splitlines()[-1]); print(g, d["value"], d["roofline"]["kernel_ms_avg"], d["pipelined"]["value"], d["roofline"]["per_ray"])
    except Exception as e: print(g, "ERR", e, open("gpurun_out/r2_bench_g%d.err"%g).read()[-400:])
P
