#!/bin/bash
# HBM-side fetch traffic (FETCH_SIZE, KiB per launch) and duration (GRBM_GUI_ACTIVE) of the attention kernel by causal work-order group size
# usage: tools/order_traffic.sh <cfg> g1 g2 ...     (group -1 = automatic, 0 = head-major)
cfg="$1"; shift
for g in "$@"; do
  echo "== $cfg SAGE_ORDER_GROUP=$g"
  SAGE_ATTN64=0 SAGE_ORDER_GROUP=$g SAGE_PMC_CFG=$cfg bash tools/pmc_passes.sh gpurun_out/order_traffic_$g "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"
done
