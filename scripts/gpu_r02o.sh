# chain build with loads in flight: throughput, pool size, phase split
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02o; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
( timeout 300 tests/gpu_quick 256 13 2>&1 | tail -2 ) | tee -a $O/summary.txt
( timeout 300 tests/gpu_quick 256 36 2>&1 | tail -2 ) | tee -a $O/summary.txt
for v in base hcp2 hcp4w12; do run $v 13 262144 8192 2; done
for l in 14 15 16 17 35; do run base $l 262144 8192 2; done
run base 13 4194304 512 2
run base 13 65536 32768 2
export LIZARDGPU_CHUNK_MB=8192
for l in 13; do
( LD_LIBRARY_PATH=$V/prof timeout 300 tests/gpu_quick 8192 $l 1 2>&1 | grep -E "batch|prof" | sed "s/^/[prof L$l] /" ) | tee -a $O/summary.txt
done
