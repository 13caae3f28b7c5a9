run() { lib=$1; shift; echo -n "$lib $*: "; env "$@" python tools/tune.py --workload wgs30x --reads 2000000 --settings auto --steps 8 --lib $lib 2>&1 | grep -o "lanes [0-9.]* ms ([0-9]* items).*mid [0-9.]* ms ([0-9]* items, [0-9]* retried)" | sed 's/heavy lanes.*mid 0.000 ms//'; }
for rep in 1 2; do
run libplo_prev.so PLO_X=1
run libplo_wpe2.so PLO_X=1
run libplo_wpe2.so PLO_LANE_CAPW=4096
run libplo_wpe4.so PLO_LANE_CAPW=2304
run libplo_wpe4.so PLO_LANE_H16=1 PLO_LANE_CAPW=2048
done
