#!/bin/bash
# needs a build with the PSEG_BG_NFULL / PSEG_BG_TAILK exploration overrides (removed from the product source) as libpiscoseg_x.so
for cfg in "3072 7" "3072 4" "3072 3" "2944 7" "2816 7" "2560 4" "2560 7" "2560 3" "3136 7" "2048 2" "3072 7"; do
  set -- $cfg
  echo "nfull=$1 tailk=$2"
  PSEG_BG_NFULL=$1 PSEG_BG_TAILK=$2 bash profiles/tools/ab.sh x || exit 1
done
