#!/bin/bash
# Runs ON THE GPU BOX: the bench.py lines committed under profiles/ (copied from gpurun_out/<tag>/).
#   gpurun -- 'bash tools/collect_bench.sh r03'
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG/bench
mkdir -p "$O"
cd "$R"
run() { # name, args...
    local name=$1; shift
    timeout -k 10 300 python3 bench.py "$@" > "$O/${TAG}_bench_$name.json" 2> "$O/$name.err" || { echo "FAILED $name"; tail -3 "$O/$name.err"; exit 1; }
    echo "done $name"
}
run L
run L_driver --steps 20 --warmup 5                 # the invocation the round-end driver uses
for v in T S M; do run $v-fwdbwd --workload $v-fwdbwd --no-cpu-baseline; done
run T-fwd --workload T-fwd                         # (with the CPU baseline: OPUS-T forward on the host cores)
run L-fwd --workload L-fwd --no-cpu-baseline
run T-fwd_batch8 --workload T-fwd --batch 8 --no-cpu-baseline
run L_batch8 --batch 8 --steps 100 --no-cpu-baseline
run L_atomic --bwd-algo atomic --steps 100 --no-cpu-baseline
run L_cameras --points cameras --no-cpu-baseline
run decoder-L --workload decoder-L --steps 100 --no-cpu-baseline
run decoder-L_accumulate --workload decoder-L --dfeat accumulate --steps 100 --no-cpu-baseline
run decoder-T --workload decoder-T --steps 100 --no-cpu-baseline
run decoder-T_accumulate --workload decoder-T --dfeat accumulate --steps 100 --no-cpu-baseline
run sampling4d-L --workload sampling4d-L --no-cpu-baseline
run sampling4d-L_input-order --workload sampling4d-L --fwd-order input --no-cpu-baseline
run train-M --workload train-M --steps 50 --no-cpu-baseline
run sampling4d-T --workload sampling4d-T --no-cpu-baseline
run L_graph --graph --no-cpu-baseline
run decoder-L_graph --workload decoder-L --graph --steps 100 --no-cpu-baseline
run decoder-L_cameras --workload decoder-L --points cameras --steps 100 --no-cpu-baseline
run L_queries --points queries --no-cpu-baseline                   # the decoder's query-clustered points
run decoder-L_queries --workload decoder-L --points queries --steps 100 --no-cpu-baseline
run chain-L --workload chain-L --steps 100 --no-cpu-baseline       # the whole sampling chain of a training step
run chain-L_queries --workload chain-L --points queries --steps 100 --no-cpu-baseline
run chain-T --workload chain-T --steps 100 --no-cpu-baseline
run chain-L_no-checkpoint --workload chain-L --no-checkpoint --steps 100 --no-cpu-baseline   # one forward per layer (288 GB of HBM: nothing to recompute)
run infer-L --workload infer-L --steps 200 --no-cpu-baseline       # ... and of an inference frame
run infer-L_queries --workload infer-L --points queries --steps 200 --no-cpu-baseline
run train-M_group --workload train-M --steps 50 --init-group --no-cpu-baseline   # RCCL in a group of one: schedule + exposed time
timeout -k 10 200 python3 tools/bench_sampling4d.py > "$O/${TAG}_sampling4d_fusion.json" 2> "$O/fusion.err" || exit 1
echo "bench lines under $O"
