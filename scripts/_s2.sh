cd $GRAFT_REPO_ROOT; O=gpurun_out/r03b; mkdir -p $O
( timeout 300 tests/gpu_quick 256 > $O/gpu_quick.log 2>&1; echo "gpu_quick rc=$?" ) | tee $O/summary.txt
for v in base pfw3 pfw5 pfw7 pfw11 pfw12; do bash scripts/gpu_session.sh variants r03b "21 262144 16384 3" $v; done
for v in base pfw5 pfw7; do bash scripts/gpu_session.sh variants r03b "41 262144 16384 3" $v; done
for cfg in "22 262144 16384 2" "42 262144 16384 2" "21 4194304 1024 2" "10 262144 65536 3" "30 262144 16384 3" "11 262144 16384 2" "13 262144 16384 2"; do bash scripts/gpu_session.sh variants r03b "$cfg" base; done
( timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" ) | tee -a $O/summary.txt; tail -6 $O/pytest.log | tee -a $O/summary.txt
bash scripts/gpu_traffic.sh r03b "21 262144 16384" > $O/traffic.log 2>&1; grep -E "^L" $O/traffic.log | tee -a $O/summary.txt
bash scripts/gpu_sq_counters.sh r03b base 21 262144 16384 > $O/sq.log 2>&1; tail -5 $O/sq.log | tee -a $O/summary.txt
