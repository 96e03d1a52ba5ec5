# Round-2 first GPU session: correctness of the restructured library, tuning-variant sweeps, GPU tests, bench.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02a; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
( timeout 600 tests/gpu_quick 512 > $O/gpu_quick.log 2>&1; echo "gpu_quick rc=$?" ) | tee -a $O/summary.txt
tail -3 $O/gpu_quick.log
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 180 tests/gpu_bench "$@" 2>&1 | tail -2 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
for v in base pf_w12 pf_w8 pf_w6 pf_w4; do run $v 21 262144 16384 3; done
for v in base f_nt f_skip f_nt_skip f_lds13; do run $v 10 262144 65536 3; done
run base 30 262144 16384 3
run base 41 262144 16384 3
run base 10 4194304 4096 3
run base 11 262144 16384 3
run base 13 262144 8192 3
run base 22 262144 16384 2
( timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" ) | tee -a $O/summary.txt
tail -15 $O/pytest.log
( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" ) | tee -a $O/summary.txt
tail -3 $O/bench.err; cat $O/bench.json | cut -c1-1500
