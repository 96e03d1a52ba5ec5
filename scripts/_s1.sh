cd $GRAFT_REPO_ROOT; O=gpurun_out/r03a; mkdir -p $O
( timeout 300 tests/gpu_quick 256 > $O/gpu_quick.log 2>&1; echo "gpu_quick rc=$?" ) | tee $O/summary.txt
( timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" ) | tee -a $O/summary.txt; tail -15 $O/pytest.log | tee -a $O/summary.txt
( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" ) | tee -a $O/summary.txt
tail -c 600 $O/bench.err
