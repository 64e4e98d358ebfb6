#!/bin/bash
# one kernel's average time (rocprofv3 --stats, one context) under environment settings, alternating: r3_ab_kstat.sh <pattern> <reps> CFG...
cd $GRAFT_REPO_ROOT
pat=$1; reps=$2; shift 2
for rep in $(seq 1 $reps); do
for cfg in "$@"; do
  envs=$(echo $cfg | tr , ' '); [ "$cfg" = "-" ] && envs="AB_NONE=1"
  echo "== $cfg"; env $envs timeout -k 10 250 bash profiles/tools/kstats.sh abk | grep -i "$pat" || exit 1
done
done
