cd $GRAFT_REPO_ROOT; O=gpurun_out/r03c; mkdir -p $O
( timeout 300 tests/gpu_quick 256 > $O/gpu_quick.log 2>&1; echo "gpu_quick rc=$?" ) | tee $O/summary.txt
grep -E "FAIL|differs" $O/gpu_quick.log | head -5 | tee -a $O/summary.txt
for v in base nosplit s13_2 s13_1; do bash scripts/gpu_session.sh variants r03c "10 262144 65536 3" $v; done
for v in base nosplit s13_2 s13_1; do bash scripts/gpu_session.sh variants r03c "30 262144 16384 3" $v; done
for v in base nosplit s13_2; do bash scripts/gpu_session.sh variants r03c "10 4194304 6656 2" $v; done
for v in base nosplit; do bash scripts/gpu_session.sh variants r03c "10 4194304 256 2" $v; bash scripts/gpu_session.sh variants r03c "30 4194304 256 2" $v; done
for v in prof3 prof9; do ( export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/lizard_amd/variants/$v; timeout 300 tests/gpu_quick 2048 21 1 2>&1 | grep -A12 "L21 batch" | sed "s/^/[$v] /" ) | tee -a $O/summary.txt; done
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_random_parity.py tests/test_frame.py tests/test_host_pipeline.py -m gpu -q -k "not soak" > $O/pytest.log 2>&1; echo "pytest rc=$?" ) | tee -a $O/summary.txt; tail -4 $O/pytest.log | tee -a $O/summary.txt
