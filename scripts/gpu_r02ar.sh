# level 30: twelve LDS tables with three pooled workspaces?
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02ar; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
for v in base l30_12p3 l30_12p3w15 l30_11p4; do run $v 30 262144 16384 3; done
