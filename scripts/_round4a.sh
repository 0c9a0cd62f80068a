cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4i; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout -k 10 600 python tests/tools/gpu_blur_mode_gap.py 6 --only 3840x2160 --json $OUT/4k_search_both_modes.json > $OUT/blur_mode_gap_4k.log 2>&1; tail -2 $OUT/blur_mode_gap_4k.log
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d.get("two_context_ratio"), d["cpu_baseline"]["value"], d["cpu_baseline"].get("placement",{}).get("candidates"), d["recursive_blur_mode"]["cached_reference"])
PY
