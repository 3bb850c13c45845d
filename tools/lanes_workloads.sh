for L in 1 2; do
FG_LANES=$L python3 tools/run_workload.py dmel_ont30 0.5 20 2>&1 | grep -E "pass 1|^\{" | cut -c1-700
done
for L in 1 2; do
FG_LANES=$L python3 tools/run_workload.py hifi30 1.0 20 2>&1 | grep -E "pass 1" | cut -c1-300
done
