#!/bin/bash
# HBM-side fetch traffic (FETCH_SIZE / WRITE_SIZE, KiB per launch) and duration (GRBM_GUI_ACTIVE) of the attention kernel by causal work order
# usage: tools/order_traffic.sh <cfg> spec1 spec2 ...   spec = SAGE_ORDER_GROUP[:SAGE_ORDER_TAILG]
#   group -1 = automatic size; 0 = head-major; n = groups of n heads  (SAGE_ORDER_TAILG: a removed experiment, see profiles/r3_run_k_order_traffic.txt)
cfg="$1"; shift
for spec in "$@"; do
  g="${spec%%:*}"; t=""; [ "$spec" != "$g" ] && t="${spec#*:}"
  echo "== $cfg SAGE_ORDER_GROUP=$g SAGE_ORDER_TAILG=$t"
  SAGE_ORDER_GROUP=$g SAGE_ORDER_TAILG=$t SAGE_PMC_CFG=$cfg bash tools/pmc_passes.sh gpurun_out/order_traffic_tmp "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"
done
