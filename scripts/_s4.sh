cd $GRAFT_REPO_ROOT; O=gpurun_out/r03d; mkdir -p $O
( timeout 600 tests/gpu_quick 256 > $O/gpu_quick.log 2>&1; echo "gpu_quick rc=$?" ) | tee $O/summary.txt
grep -E "FAIL|differs|one block n=262144" $O/gpu_quick.log | head -30 | tee -a $O/summary.txt
for cfg in "21 262144 16384 3" "41 262144 16384 3" "22 262144 16384 2" "42 262144 16384 2" "11 262144 16384 2" "31 262144 16384 2" "21 4194304 1024 2" "11 4194304 1024 2"; do for v in base drain; do bash scripts/gpu_session.sh variants r03d "$cfg" $v; done; done
for cfg in "10 262144 65536 3" "30 262144 16384 3" "13 262144 16384 2" "13 4194304 1024 2" "13 17825792 64 1" "21 17825792 256 1" "11 17825792 256 1"; do bash scripts/gpu_session.sh variants r03d "$cfg" base; done
( timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" ) | tee -a $O/summary.txt; tail -5 $O/pytest.log | tee -a $O/summary.txt
