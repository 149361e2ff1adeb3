#!/bin/bash
# round 5, call C: per-launch timeline of rank 0 of 8 (one 64-layer pass), round-4 schedule against census-sized grids
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
for cfg in "old:RAYHIP_DYNAMIC=0 RAYHIP_CENSUS=0" "census:RAYHIP_DYNAMIC=0"; do
  name=${cfg%%:*}; env=${cfg#*:}
  rm -rf /tmp/prof_$name
  (cd /tmp && env $env rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$name -o shard -- python $GRAFT_REPO_ROOT/tools/shard_profile.py 8 64 > $O/c_prof_$name.log 2>&1)
  python tools/pass_timeline.py /tmp/prof_$name > $O/c_timeline_n8_$name.txt 2>&1
  tail -3 $O/c_timeline_n8_$name.txt
done
