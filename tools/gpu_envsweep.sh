# the GPU suite under the switches that select alternate code paths with the same results (README "Runtime switches")
set -u
mkdir -p gpurun_out/envsweep
run() { name=$1; shift; echo "== $name"; env "$@" timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/envsweep/$name.log; }
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
  esac
done
