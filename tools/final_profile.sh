#!/bin/bash
# Round artefacts on the GPU box: bench line, rocprofv3 kernel trace of the same command, HBM traffic (two --pmc passes).
# usage: tools/final_profile.sh <tag>     (writes gpurun_out/<tag>_*)
tag=$1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
rm -rf gpurun_out/prof_$tag; mkdir -p gpurun_out/prof_$tag
rocprofv3 --kernel-trace -d gpurun_out/prof_$tag/kt -o kt -- python3 bench.py --no-cpu-baseline --extra-events 0 > gpurun_out/prof_$tag/kt.log 2>&1 || exit 2
python3 tools/rocpd_stats.py $(find gpurun_out/prof_$tag/kt -name "*.db" | head -1) gpurun_out/${tag}_bench_kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_$tag/f -o f -- python3 bench.py --no-cpu-baseline --steps 5 --extra-events 0 > gpurun_out/prof_$tag/f.log 2>&1 || exit 3
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_$tag/w -o w -- python3 bench.py --no-cpu-baseline --steps 5 --extra-events 0 > gpurun_out/prof_$tag/w.log 2>&1 || exit 4
python3 tools/traffic_from_pmc.py $(find gpurun_out/prof_$tag/f -name "*.db" | head -1) $(find gpurun_out/prof_$tag/w -name "*.db" | head -1) gpurun_out/${tag}_traffic.json 4096
find gpurun_out/prof_$tag -name "*.db" -delete
