# the GPU suite under the switches that select alternate code paths with the same results (README "Runtime switches")
set -u
mkdir -p gpurun_out/envsweep
run() { name=$1; shift; echo "== $name"; env "$@" timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/envsweep/$name.log; }
for cfg in "$@"; do
  case $cfg in
    eager_beta) run eager_beta GTNX_EAGER_BETA=1 ;;
    no_ranked) run no_ranked GTNX_NO_RANKED_TIES=1 ;;
    no_slice) run no_slice GTNX_REGION_NO_SLICE_PATH=1 ;;
    no_prefetch) run no_prefetch GTNX_NO_ITEM_PREFETCH=1 ;;
    threads4) run threads4 GTN_AMD_THREADS=4 ;;
    viterbi_wg) run viterbi_wg GTNX_VITERBI_WG=1 ;;
    eager_weights) run eager_weights GTNX_REGION_EAGER_WEIGHTS=1 ;;
    no_eager_beta) run no_eager_beta GTNX_NO_EAGER_BETA=1 ;;
    device_levelize) run device_levelize GTNX_DEVICE_LEVELIZE=1 ;;
    no_fused_copy) run no_fused_copy GTNX_NO_FUSED_COPY=1 ;;
    no_node_order) run no_node_order GTNX_NO_NODE_ORDER_TIES=1 ;;
    # round 6
    no_band_patch) run no_band_patch GTNX_NO_BAND_PATCH=1 ;;
    no_ranked_first) run no_ranked_first GTNX_NO_RANKED_FIRST=1 ;;
    no_closed_ranks) run no_closed_ranks GTNX_NO_CLOSED_RANKS=1 ;;
    check_closed_ranks) run check_closed_ranks GTNX_CHECK_CLOSED_RANKS=1 ;;
    fixed_grad_per_wave) run fixed_grad_per_wave GTNX_FIXED_GRAD_PER_WAVE=1 ;;
    fixed_grad_atomics) run fixed_grad_atomics GTNX_FIXED_GRAD_ATOMICS=1 ;;
    h2d_runtime) run h2d_runtime GTNX_H2D_KERNEL_BYTES=0 ;;
    # round 5
    no_graph_slab) run no_graph_slab GTNX_NO_GRAPH_SLAB=1 ;;
    defer_full_64) run defer_full_64 GTNX_DEFER_FULL=64 ;;
    defer_full_max) run defer_full_max GTNX_DEFER_FULL=100000000 ;;
    lazy0) run lazy0 GTNX_LAZY_COMPOSE=0 ;;
    lazy1) run lazy1 GTNX_LAZY_COMPOSE=1 ;;
    lazy2) run lazy2 GTNX_LAZY_COMPOSE=2 ;;
    no_band) run no_band GTNX_NO_BAND=1 ;;
    no_lazy_pairs) run no_lazy_pairs GTNX_NO_LAZY_PAIRS=1 ;;
    dense_valu) run dense_valu GTNX_DENSE_VALU=1 ;;
    no_dense) run no_dense GTNX_NO_DENSE=1 ;;
    no_fused_scatter) run no_fused_scatter GTNX_NO_FUSED_SCATTER=1 ;;
    chain) run chain GTNX_CHAIN=1 ;;
    classic_bitmaps) run classic_bitmaps GTNX_CLASSIC_BITMAPS=1 ;;
    full_compose) run full_compose GTNX_FULL_COMPOSE=1 ;;
    sync_compose) run sync_compose GTNX_SYNC_COMPOSE=1 ;;
    no_deep) run no_deep GTNX_NO_DEEP=1 ;;
    no_wide_compose) run no_wide_compose GTNX_NO_WIDE_COMPOSE=1 ;;
    force_wide_compose) run force_wide_compose GTNX_FORCE_WIDE_COMPOSE=1 ;;
    no_pairs_compose) run no_pairs_compose GTNX_NO_PAIRS_COMPOSE=1 ;;
    trim_fwd_first) run trim_fwd_first GTNX_TRIM_FWD_FIRST=1 ;;
    trim_fwd_never) run trim_fwd_never GTNX_TRIM_FWD_FIRST=0 ;;
    grid_replication) run grid_replication GTNX_GRID_REPLICATION=1 ;;
    inline_replication) run inline_replication GTNX_INLINE_REPLICATION=1 ;;
    narrow_compose) run narrow_compose GTNX_NARROW_COMPOSE=1 ;;
    fixed_grad_narrow) run fixed_grad_narrow GTNX_FIXED_GRAD_NARROW=1 ;;
    mallopt) run mallopt GTNX_MALLOPT=1 ;;
    args_in_place) run args_in_place GTNX_ARGS_IN_PLACE=1024 ;;
  esac
done
