run() { echo -n "$*: "; env "$@" python tools/tune.py --workload wgs30x --reads 2000000 --settings auto --steps 8 2>&1 | grep -o "lanes [0-9.]* ms" | head -1; }
for rep in 1 2; do
run PLO_X=default32
run PLO_LANE_H16=1
run PLO_LANE_H16=1 PLO_LANE_CAPW=2048 PLO_LANE_KVS=512 PLO_LANE_SORT_WINDOW=256
run PLO_LANE_H16=1 PLO_LANE_CAPW=2048 PLO_LANE_KVS=512 PLO_LANE_SORT_WINDOW=512
run PLO_LANE_SORT_WINDOW=256
done
