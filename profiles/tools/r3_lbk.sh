# k_runs_band look-back windows per round (LB_K = 1 / 2 / 4: 256 / 512 / 1024 predecessor states), scan in block-index order
cd $GRAFT_REPO_ROOT
for k in "" _lbk2 _lbk4; do
  export PSEG_LIB=$GRAFT_REPO_ROOT/imageprocessing_amd/libpiscoseg$k.so
  echo "== lib $k"
  bash profiles/tools/kstats.sh lbk$k 2>&1 | grep "k_runs_band\|k_convex\|k_label_fill\|k_rank_ext\|k_crops"
done
