#!/bin/bash
# one kernel's time across library variants: ab_kstat.sh <kernel name pattern> <variant>... (base = in-tree build)
cd $GRAFT_REPO_ROOT
pat=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset PSEG_LIB; else export PSEG_LIB=$GRAFT_REPO_ROOT/variants/libpiscoseg_$v.so; fi
  echo "== $v"; timeout -k 10 250 bash profiles/tools/kstats.sh abk_$v | grep -i "$pat" || exit 1
done
