#!/bin/bash
# usage: tools/sweep_lib.sh "name1 name2 ..." [bench args]  -> ms per step with raytracer_amd/lib_<name>.so ("-" = the in-tree build)
names=$1; shift
for n in $names; do
  if [ "$n" = "-" ]; then unset PM_LIB; else export PM_LIB=$GRAFT_REPO_ROOT/raytracer_amd/lib_$n.so; fi
  bash tools/sweep_env.sh PM_VARIANT "$n" "$@" | grep "ms/step"
done
