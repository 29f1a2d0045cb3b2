#!/bin/bash
# The reference's own example drivers (examples/_ref, built unmodified by `make -C examples ref`) on the GPU: first and last KKT errors, time per update.
cd $GRAFT_REPO_ROOT
W=$(mktemp -d); mkdir -p $W/build $W/iiwa_description/urdf $W/anymal_b_simple_description/urdf
cp tests/golden/urdf/iiwa14.urdf $W/iiwa_description/urdf/; cp tests/golden/urdf/anymal.urdf $W/anymal_b_simple_description/urdf/
cd $W/build
for b in $GRAFT_REPO_ROOT/examples/_ref/*; do
  n=$(basename $b)
  $b > $n.out 2> $n.err; rc=$?
  echo "== $n (exit code $rc)"
  grep -E "Initial KKT|KKT error after iteration (1|2|5|10|20|30|50|100) =|CPU time per update" $n.out
done
