cd $GRAFT_REPO_ROOT; O=gpurun_out/r03e; mkdir -p $O; : > $O/summary.txt
for v in base nolazy prio; do bash scripts/gpu_session.sh variants r03e "21 262144 16384 3" $v; done
for v in base nolazy; do bash scripts/gpu_session.sh variants r03e "41 262144 16384 3" $v; bash scripts/gpu_session.sh variants r03e "22 262144 16384 2" $v; done
for v in base p11c5b3 p10c6b3; do bash scripts/gpu_session.sh variants r03e "30 262144 16384 3" $v; done
( timeout 300 tests/gpu_quick 256 21 > $O/gpu_quick21.log 2>&1; echo "gpu_quick 21 rc=$?" ) | tee -a $O/summary.txt
( export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/lizard_amd/variants/p11c5b3; timeout 300 tests/gpu_quick 256 30 > $O/gpu_quick30.log 2>&1; echo "gpu_quick 30 p11c5b3 rc=$?" ) | tee -a $O/summary.txt
