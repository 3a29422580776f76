#!/bin/bash
# After the round-5 consolidation: (1) the route census with channels-last outputs, (2) the wrapping paddings of the row-span /
# flat-stream kernels before (variants/oldpads.so: one instantiation per mode -- the library of commit f52a8b4 linked with the other
# objects of the tree, built by hand with tools/build_variant.sh-style commands; not kept) and after (the mode as a kernel argument)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/consol
python3 tools/route_census.py --cases 6000 --seed 1 --out gpurun_out/consol/route_census.txt > gpurun_out/consol/census.log 2>&1
tail -2 gpurun_out/consol/census.log
#  the "before" leg needs the hand-built variant library.  Recipe: `git worktree add /tmp/oldpads f52a8b4`, compile its
# csrc/shiftnd_span.hip and csrc/shiftnd_flat.hip with build.py's HIPCC_FLAGS, link them with the other objects of THIS tree's
# build/ into variants/oldpads.so (tools/build_variant.sh shows the link line).  Without it only the "after" leg runs and the
# script says so with a non-zero exit code -- it never writes an error message into a result file.
OLD=$GRAFT_REPO_ROOT/variants/oldpads.so
for r in 1 2; do
  python3 tools/flat_bench.py --pads 0,1,2,3,4 --iters 30 > gpurun_out/consol/pads_new_$r.txt 2>&1
  [ -f "$OLD" ] && SHIFTND_HIP_LIB=$OLD python3 tools/flat_bench.py --pads 0,1,2,3,4 --iters 30 > gpurun_out/consol/pads_old_$r.txt 2>&1
done
tail -3 gpurun_out/consol/pads_new_2.txt
if [ ! -f "$OLD" ]; then echo "r5_consolidation.sh: variants/oldpads.so is missing -- the before/after comparison was NOT regenerated (see the recipe above)" >&2; exit 3; fi
