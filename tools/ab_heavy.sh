cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/win
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_parity.py -x -q -m gpu -k "heavy_items_lane or indel_dense or fuzz_hip or tiny or golden or full_size_stress" 2>&1 | tail -3
B="python bench.py --workload stress --reads 100000 --steps 3 --warmup 1 --no-cpu-baseline --e2e-reads 0 --overlap-workers 0 --window-calls 0"
$B 2>/dev/null | tail -1 > gpurun_out/win/stress.json
python bench.py --workload stress --reads 2000000 --steps 2 --warmup 1 --no-cpu-baseline --e2e-reads 0 --window-calls 0 2>/dev/null | tail -1 > gpurun_out/win/stress2m.json
python - <<'PY'
import json
for n in ("stress","stress2m"):
    try:
        b=json.load(open(f"gpurun_out/win/{n}.json")); r=b["roofline"]
        print(n, round(b["value"]), round(b["ms_per_step"],3), r["kernel"], round(r["kernel_ms"],3), "frac", round(r["frac"],4), "lanes", round(r.get("lift_lanes_ms"),3), "mid", round(r.get("lift_mid_ms"),3), "enum", round(r.get("enumerate_ms"),3), b["config"].get("retry_items_per_gpu"), b.get("overlap"))
    except Exception as e: print(n, "failed", e)
PY
python tools/tune.py --workload stress --reads 100000 --settings auto --steps 3 --timing 2>&1 | grep "lane phase\|trip counts"
