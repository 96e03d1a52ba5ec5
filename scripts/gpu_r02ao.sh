# fast parser: the wave-wide extension compare of the lowest candidate lane requested together with the candidates' bytes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02ao; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
( timeout 300 tests/gpu_quick 512 10 2>&1 | tail -2 | head -1 ) | tee -a $O/summary.txt
for v in base nospec base; do run $v 10 262144 65536 3; done
for v in base nospec; do run $v 30 262144 16384 3; run $v 11 262144 16384 2; done
