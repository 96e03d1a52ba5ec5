# level 41: Huffman workspaces from a pool of 3, four LDS tables (4 + 6 waves) instead of three (3 + 9)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02ap; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
( timeout 300 tests/gpu_quick 512 41 2>&1 | tail -3 ) | tee -a $O/summary.txt
for v in base h41_old h41_p2; do run $v 41 262144 16384 2; done
run base 41 131072 32768 2
