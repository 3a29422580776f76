#!/usr/bin/env bash
# round-4 final evidence in one GPU call: every bench workload (tools/r4_profiles.sh), the extras / cropped channels-last / NDHWC
# tool profiles; everything lands in gpurun_out/r04p/ (copy to profiles/)
bash tools/r4_profiles.sh c2 c2a c3_pad0 c3_pad1 c3_pad2 c3_pad3 c3_pad4 c4 c5 c2crop c2acrop t1 t1a c1d c1da c1dh
for t in extra clcrop cl3d; do
    if [ $t = extra ]; then bash profiles/collect_tool.sh $t tools/prof_workloads.py > /dev/null 2>&1
    else bash profiles/collect_tool.sh $t tools/prof_workloads.py --only $t > /dev/null 2>&1; fi
    cp gpurun_out/prof_$t/summary.txt gpurun_out/r04p/r04_${t}_rocprof_summary.txt
    cp gpurun_out/prof_$t/traffic.json gpurun_out/r04p/r04_${t}_traffic.json 2>/dev/null
done
python3 tools/cl3d_bench.py > gpurun_out/r04p/r04_cl3d_bench.txt 2>&1
python3 tools/clt_depth_bench.py > gpurun_out/r04p/r04_cl_tiled_times.txt 2>&1
tail -3 gpurun_out/r04p/r04_cl3d_bench.txt; cat gpurun_out/r04p/r04_cl_tiled_times.txt | tail -1
