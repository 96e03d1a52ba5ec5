# streaming loops of the encode passes and the Huffman stage with their loads ahead / side by side
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02aa; mkdir -p $O
run () { name=$1; shift; ( timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
( timeout 600 tests/gpu_quick 512 2>&1 | tail -3 ) | tee -a $O/summary.txt
run base 10 262144 65536 3
run base 30 262144 16384 3
run base 21 262144 16384 3
run base 41 262144 16384 3
run base 11 262144 16384 3
run base 13 262144 16384 3
run base 10 4194304 6656 3
