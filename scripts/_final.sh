cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_final; mkdir -p $OUT
python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench (driver's command) rc=$?"
python3 - <<PY
import json
for f in ("bench.json","bench_driver_cmd.json"):
    d=json.load(open("$OUT/"+f))
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["cpu_baseline"]["value"], d["cpu_baseline"]["pinned_to_cpus"], d["recursive_blur_mode"]["cached_reference"]["ms_per_pass"], d["recursive_blur_mode"].get("ms_per_search_pass_gpu_side"), d["two_context_ratio"]["value"], d["roofline"]["counters"]["stale"])
PY
