cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/win
timeout 1500 python -m pytest tests/test_bam.py -x -q -m gpu -k "chr20_size" 2>&1 | tail -3
python bench.py 2>gpurun_out/win/e2e.err | tail -1 > gpurun_out/win/e2e.json
python - <<'PY'
import json
b=json.load(open("gpurun_out/win/e2e.json"))
e=b.get("end_to_end") or {}
print(b["value"], b["ms_per_step"], b["roofline"]["frac"], b["cpu_baseline"], (b.get("window_50k") or {}).get("value"))
print(json.dumps({k:v for k,v in e.items() if k in ("value","seconds","records_verified","stage_busy_s","device_finished")}, indent=1))
PY
grep -v "^\[plo\]" gpurun_out/win/e2e.err | tail -5
