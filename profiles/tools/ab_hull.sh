#!/bin/bash
# k_convex by phase / variant on the dense workload and the headline workload (kernel statistics, one context)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset PSEG_LIB; else export PSEG_LIB=$GRAFT_REPO_ROOT/variants/libpiscoseg_$v.so; fi
  for wl in uvp5_dense_saved pisco; do
    echo "== $v $wl"; timeout -k 10 250 bash profiles/tools/kstats.sh abh_${v}_$wl --workload $wl | grep -i "k_convex" || exit 1
  done
done
