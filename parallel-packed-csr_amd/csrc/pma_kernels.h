// All kernels of the PMA engine (one translation unit: csrc/ppcsr_hip.hip).
//   pma_rounds.h       strict prefix rounds (k_plan / k_check / k_apply) and the exclusive executor (k_exclusive)
//   pma_spec_rounds.h  speculative rounds (o_plan / o_check / o_apply / o_big / o_settle): the default scheduler
//   pma_rebalance.h    whole-array / big-window rebalance, in-place window rebalance, snapshots, maintenance
//   pma_scan.h         queries, bulk neighbour scan, bulk build, BFS / PageRank
//   pma_exchange.h     owner bucketing for the multi-GPU exchange
#pragma once
#include "pma_rounds.h"
#include "pma_spec_rounds.h"
#include "pma_rebalance.h"
#include "pma_scan.h"
#include "pma_exchange.h"
